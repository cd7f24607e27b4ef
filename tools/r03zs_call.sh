OUT=gpurun_out/r03zs; mkdir -p $OUT
timeout 420 python tools/config4_biobank.py --samples 250000 --variants 10000000 > $OUT/config5_full_size_one_gpu.json 2> $OUT/err.txt; echo "exit $?" >> $OUT/err.txt
cat $OUT/config5_full_size_one_gpu.json; tail -3 $OUT/err.txt
