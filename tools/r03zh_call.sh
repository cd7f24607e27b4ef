OUT=gpurun_out/r03zh; mkdir -p $OUT
export TMPDIR=/tmp
date > $OUT/summary.txt
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 800 --durations=8 > $OUT/tests.log 2>&1; echo "tests exit $?" | tee -a $OUT/summary.txt )
tail -15 $OUT/tests.log | tee -a $OUT/summary.txt
( PCOA_GUARD_CASES=60 timeout 1200 python -m pytest tests/test_gpu_guard.py -m gpu -q -p no:cacheprovider --timeout 1100 --durations=5 > $OUT/guard_long.log 2>&1; echo "guard long exit $?" | tee -a $OUT/summary.txt )
tail -6 $OUT/guard_long.log | tee -a $OUT/summary.txt
date >> $OUT/summary.txt
