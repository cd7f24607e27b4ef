OUT=gpurun_out/r03zr; mkdir -p $OUT
timeout 300 tools/exp_bits --coreside-alt > $OUT/coreside_alt.txt 2>&1; echo "exit $?" >> $OUT/coreside_alt.txt
cat $OUT/coreside_alt.txt
