OUT=gpurun_out/r03zl; mkdir -p $OUT
timeout 400 tools/exp_bits --coreside > $OUT/coreside.txt 2>&1; echo "exit $?" >> $OUT/coreside.txt
cat $OUT/coreside.txt
