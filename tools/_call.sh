set +e
OUT=gpurun_out/r06f; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x -k "plink or bed or multi_engine" > $OUT/tests.log 2>&1
echo "tests exit $?" > $OUT/summary.txt; tail -3 $OUT/tests.log >> $OUT/summary.txt
echo "== 4M variants, default --stream-rows" >> $OUT/summary.txt
timeout 900 python tools/plink_stream_e2e.py 4000000 2504 >> $OUT/summary.txt 2>&1
echo "== 4M variants, --stream-rows 65536" >> $OUT/summary.txt
timeout 900 python tools/plink_stream_e2e.py 4000000 2504 /tmp/plink_e2e "--stream-rows 65536" 2>&1 | grep -E "device decode\]|two engines" | head -3 >> $OUT/summary.txt
echo "== 1M variants, default" >> $OUT/summary.txt
timeout 900 python tools/plink_stream_e2e.py 1000000 2504 2>&1 | grep -E "device decode\]|two engines" | head -3 >> $OUT/summary.txt
for rnd in 1 2; do
for set in "X=0" "PCOA_KBITS_W4=2 PCOA_KBITS_PIPE_WGS=256"; do
  echo "== [$set]" >> $OUT/summary.txt
  env $set timeout 300 python tools/alt_inputs_ab.py 20 2>&1 | grep -E "^pipeline|S equal" | head -3 >> $OUT/summary.txt
done
done
cat $OUT/summary.txt
