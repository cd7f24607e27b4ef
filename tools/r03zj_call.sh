OUT=gpurun_out/r03zj; mkdir -p $OUT
export TMPDIR=/tmp
date > $OUT/summary.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/summary.txt ); tail -2 $OUT/smoke.log | tee -a $OUT/summary.txt
for i in 1 2 3; do
  ( timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 800 > $OUT/tests_$i.log 2>&1; echo "full GPU suite, run $i: exit $? -- $(tail -1 $OUT/tests_$i.log)" | tee -a $OUT/summary.txt )
done
( PCOA_GUARD_CASES=150 timeout 1200 python -m pytest tests/test_gpu_guard.py -m gpu -q -p no:cacheprovider --timeout 1100 > $OUT/guard_long.log 2>&1; echo "guard sweeps, 150 cases per mode (+ 75 pipeline cases): exit $? -- $(tail -1 $OUT/guard_long.log)" | tee -a $OUT/summary.txt )
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o trace -- python $OLDPWD/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --pcoa-reps 1 > $OLDPWD/$OUT/prof_bench.json 2> $OLDPWD/$OUT/prof.err; echo "prof exit $?" | tee -a $OLDPWD/$OUT/summary.txt )
find $OUT/prof -name "*kernel_stats*" | head -1 | while read f; do cp "$f" $OUT/kernel_stats.csv; head -8 "$f" | cut -c1-220; done | tee -a $OUT/summary.txt
python - <<'PY' | tee -a gpurun_out/r03zj/summary.txt
import json
for name in ("bench.json", "prof_bench.json"):
    d=json.load(open("gpurun_out/r03zj/" + name))
    print(name, "value %.1f M/s ms/step %.3f frac %.3f avg_launch_ms %.4f (pack) %.4f (gram) traffic %s" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline_other"]["avg_launch_ms"], d["roofline"]["traffic"]))
    for k in ("sustained","alt_input_u8","alt_input_bits"):
        if k in d: print("  ", k, d[k].get("value"))
PY
find $OUT/prof -name "*kernel_trace*" -size +6M -delete
date >> $OUT/summary.txt
