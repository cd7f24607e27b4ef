OUT=gpurun_out/r03z; mkdir -p $OUT
export TMPDIR=/tmp
date > $OUT/summary.txt
for N in 2048 1792 1280 2504; do
  for E in "" "PCOA_KBITS_CORESIDE=0" "PCOA_KBITS_MODE=2"; do
    env $E timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --samples $N --no-extras --no-cpu-baseline --pcoa-reps 1 > $OUT/b.json 2>> $OUT/b.err
    python -c "import json; d=json.load(open('$OUT/b.json')); print('N=$N [%s] value %.1f M/s  ms/step %.3f  pack %.3f ms  gram %.3f ms  co_resident=%s lockstep=%d evensplit=%d' % ('$E', d['value']/1e6, d['ms_per_step'], d['pack_ms_per_step'], d['gram_ms_per_step'], d['pipeline'].get('co_resident'), d['pipeline']['lockstep_launches'], d['pipeline']['evensplit_launches']))" | tee -a $OUT/summary.txt
  done
done
echo "== small calls" | tee -a $OUT/summary.txt
timeout 300 python tools/small_calls.py 2>&1 | tail -5 | tee -a $OUT/summary.txt
date >> $OUT/summary.txt
