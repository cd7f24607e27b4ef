OUT=gpurun_out/r03zq; mkdir -p $OUT
export TMPDIR=/tmp
date > $OUT/summary.txt
( timeout 300 tools/exp_bits > $OUT/exp_bits.txt 2>&1; echo "exp_bits exit $?" | tee -a $OUT/summary.txt ); grep -E "part 1|MISMATCH|RESULT" $OUT/exp_bits.txt | tee -a $OUT/summary.txt
echo "== small calls" | tee -a $OUT/summary.txt
timeout 300 python tools/small_calls.py 2>&1 | tail -5 | tee -a $OUT/summary.txt
( timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 800 > $OUT/tests.log 2>&1; echo "full GPU suite: exit $? -- $(tail -1 $OUT/tests.log)" | tee -a $OUT/summary.txt )
( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --pcoa-reps 1 > $OUT/bench.json 2> $OUT/bench.err; python -c "import json; d=json.load(open('$OUT/bench.json')); print('bench value %.1f M/s ms/step %.3f' % (d['value']/1e6, d['ms_per_step']))" | tee -a $OUT/summary.txt )
date >> $OUT/summary.txt
