OUT=gpurun_out/r03r; mkdir -p $OUT
timeout 300 tools/exp_bits --pipe-study > $OUT/pipe_study.txt 2>&1; echo "exit $?" >> $OUT/pipe_study.txt
cat $OUT/pipe_study.txt
