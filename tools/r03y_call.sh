OUT=gpurun_out/r03y; mkdir -p $OUT
export TMPDIR=/tmp
date > $OUT/summary.txt
# 1. co-resident pipeline under the guard
( timeout 900 python -m pytest tests/test_gpu_guard.py -m gpu -q -p no:cacheprovider --timeout 800 -x -k "co_resident" > $OUT/guard_ring.log 2>&1; echo "guard ring exit $?" | tee -a $OUT/summary.txt )
tail -5 $OUT/guard_ring.log | tee -a $OUT/summary.txt
# 2. other sample counts: co-resident vs disjoint vs serial
for N in 1536 2048 2304 2504; do
  for E in "" "PCOA_KBITS_CORESIDE=0" "PCOA_PIPELINE=0"; do
    env $E timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --samples $N --no-extras --no-cpu-baseline --pcoa-reps 1 > $OUT/b.json 2>> $OUT/b.err
    python -c "import json; d=json.load(open('$OUT/b.json')); print('N=$N [%s] value %.1f M/s  ms/step %.3f  pack %.3f ms  gram %.3f ms  co_resident=%s pipeline=%s' % ('$E', d['value']/1e6, d['ms_per_step'], d['pack_ms_per_step'], d['gram_ms_per_step'], d['pipeline'].get('co_resident'), d['pipeline']['pipeline']))" | tee -a $OUT/summary.txt
  done
done
# 3. small calls
for E in "" "PCOA_KBITS_CORESIDE=0"; do
  echo "== small calls [$E]" | tee -a $OUT/summary.txt
  env $E timeout 300 python tools/small_calls.py 2>&1 | tail -6 | tee -a $OUT/summary.txt
done
date >> $OUT/summary.txt
