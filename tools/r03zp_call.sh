OUT=gpurun_out/r03zp; mkdir -p $OUT
timeout 1100 python tools/soak_pipeline.py 20000 > $OUT/soak.txt 2> $OUT/soak.err; echo "exit $?" >> $OUT/soak.txt
cat $OUT/soak.txt; tail -3 $OUT/soak.err
