OUT=gpurun_out/r03zm; mkdir -p $OUT
export TMPDIR=/tmp
bash tools/gpu_round.sh r03zm pmc > /dev/null 2>&1
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 800 -k "pipeline or guard or bench or config1 or fuzz" > $OUT/tests.log 2>&1; echo "tests (pipeline / guard / bench / config1 / fuzz) exit $? -- $(tail -1 $OUT/tests.log)" | tee -a $OUT/summary.txt )
for i in 1 2; do
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --pcoa-reps 1 > $OUT/bench_$i.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt )
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o trace -- python $OLDPWD/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --pcoa-reps 1 > $OLDPWD/$OUT/prof_bench.json 2> $OLDPWD/$OUT/prof.err; echo "prof exit $?" | tee -a $OLDPWD/$OUT/summary.txt )
find $OUT/prof -name "*kernel_stats*" | head -1 | while read f; do cp "$f" $OUT/kernel_stats.csv; head -5 "$f" | cut -c1-220; done | tee -a $OUT/summary.txt
python - <<'PY' | tee -a gpurun_out/r03zm/summary.txt
import json
for name in ("bench_1.json", "bench_2.json", "prof_bench.json"):
    d=json.load(open("gpurun_out/r03zm/" + name))
    print(name, "value %.1f M/s ms/step %.3f frac %.3f avg_launch_ms %.4f (pack) %.4f (gram) kernel %s" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline_other"]["avg_launch_ms"], d["roofline"]["kernel"]))
s=json.load(open("gpurun_out/r03zm/pmc_summary.json"))
for k in ("pack_kbits_ring_kernel","gram_kbits_kernel"):
    v=s[k]; print(k, "FETCH_SIZE %.0f KiB (x2 = %.3f GB) WRITE_SIZE %.0f KiB" % (v["FETCH_SIZE"], 2*v["FETCH_SIZE"]*1024/1e9, v["WRITE_SIZE"]))
PY
find $OUT/prof -name "*kernel_trace*" -size +6M -delete; find $OUT -name "*counter_collection*" -size +4M -delete
