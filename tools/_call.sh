set +e
OUT=gpurun_out/r06h; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_multi_engine.py -m gpu -q -p no:cacheprovider --timeout 120 -x > $OUT/tests.log 2>&1
echo "tests exit $?" > $OUT/summary.txt; tail -3 $OUT/tests.log >> $OUT/summary.txt
echo "== 4M variants, default --stream-rows" >> $OUT/summary.txt
timeout 600 python tools/plink_stream_e2e.py 4000000 2504 >> $OUT/summary.txt 2>&1
echo "== 1M variants, default" >> $OUT/summary.txt
timeout 600 python tools/plink_stream_e2e.py 1000000 2504 2>&1 | grep -E "device decode\]|two engines" | head -3 >> $OUT/summary.txt
cat $OUT/summary.txt
