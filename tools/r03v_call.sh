OUT=gpurun_out/r03v; mkdir -p $OUT
export TMPDIR=/tmp
date > $OUT/summary.txt
( timeout 300 tools/exp_bits > $OUT/exp_bits.txt 2>&1; echo "exp_bits exit $?" | tee -a $OUT/summary.txt )
grep -E "part 1|MISMATCH|RESULT|^pipe|serial" $OUT/exp_bits.txt | tee -a $OUT/summary.txt
for round in 1 2; do
for cr in 0 1; do
  PCOA_KBITS_CORESIDE=$cr timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --pcoa-reps 1 > $OUT/bench_cr$cr.json 2>> $OUT/bench_ab.err
  python -c "import json; d=json.load(open('$OUT/bench_cr$cr.json')); print('CORESIDE=$cr: value %.1f M/s, ms/step %.3f, gram %.3f ms, pack %.3f ms, frac %.3f, kernel %s, pipe %s' % (d['value']/1e6, d['ms_per_step'], d['gram_ms_per_step'], d['pack_ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], {k: d['pipeline'][k] for k in ('co_resident','pipeline_pre_pass_cus','pipeline_contraction_cus','lockstep_launches','evensplit_launches')}))" | tee -a $OUT/summary.txt
done; done
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 800 --durations=8 -x > $OUT/tests.log 2>&1; echo "tests exit $?" | tee -a $OUT/summary.txt )
tail -15 $OUT/tests.log | tee -a $OUT/summary.txt
date >> $OUT/summary.txt
