OUT=gpurun_out/r03zn; mkdir -p $OUT
export TMPDIR=/tmp
date > $OUT/summary.txt
for N in 10240 12288 16384; do
  for E in "PCOA_KBITS_CORESIDE_MAX_NPAD=32768" "PCOA_PIPELINE=0"; do
    env $E timeout 500 python bench.py --gpus 1 --steps 6 --warmup 2 --samples $N --distinct-batches 2 --variants 500000 --no-extras --no-cpu-baseline --pcoa-reps 1 > $OUT/b.json 2>> $OUT/b.err
    python -c "import json; d=json.load(open('$OUT/b.json')); print('N=$N [%s] value %.1f M/s  ms/step %.3f  pack %.3f ms  gram %.3f ms  co_resident=%s pipeline=%s lockstep=%d evensplit=%d launches=%d' % ('$E', d['value']/1e6, d['ms_per_step'], d['pack_ms_per_step'], d['gram_ms_per_step'], d['pipeline'].get('co_resident'), d['pipeline']['pipeline'], d['pipeline']['lockstep_launches'], d['pipeline']['evensplit_launches'], d['roofline_other' if d['roofline']['bound']=='hbm' else 'roofline']['launches']))" | tee -a $OUT/summary.txt
  done
done
date >> $OUT/summary.txt
