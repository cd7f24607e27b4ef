OUT=gpurun_out/r03p
mkdir -p $OUT
export TMPDIR=/tmp
date > $OUT/summary.txt
# 1. long guarded sweeps (both modes, kernels serialised)
( PCOA_GUARD_CASES=150 timeout 1200 python -m pytest tests/test_gpu_guard.py -m gpu -q -p no:cacheprovider --timeout 1100 --durations=5 > $OUT/guard_long.log 2>&1; echo "guard long exit $?" | tee -a $OUT/summary.txt )
tail -6 $OUT/guard_long.log | tee -a $OUT/summary.txt
# 2. full GPU suite
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 800 --durations=12 > $OUT/tests.log 2>&1; echo "tests exit $?" | tee -a $OUT/summary.txt )
tail -20 $OUT/tests.log | tee -a $OUT/summary.txt
# 3. the driver's bench command
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt )
# 4. kernel trace of exactly that command without the extras
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o trace -- python $OLDPWD/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --pcoa-reps 1 > $OLDPWD/$OUT/prof_bench.json 2> $OLDPWD/$OUT/prof.err; echo "prof exit $?" | tee -a $OLDPWD/$OUT/summary.txt )
find $OUT/prof -name "*kernel_stats*" | head -2 | while read f; do cp "$f" $OUT/kernel_stats.csv; head -12 "$f" | cut -c1-200; done | tee -a $OUT/summary.txt
python - <<'PY' | tee -a gpurun_out/r03p/summary.txt
import csv, glob, collections
f = glob.glob("gpurun_out/r03p/prof/**/*kernel_trace.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    per = collections.defaultdict(list)
    for r in rows:
        per[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:8]:
        v2 = sorted(v)
        print("%-72s n=%4d sum %8.3f ms  median %7.4f  max %7.4f" % (k, len(v), sum(v), v2[len(v2)//2], v2[-1]))
PY
find $OUT/prof -name "*kernel_trace*" -size +6M -delete
date >> $OUT/summary.txt
