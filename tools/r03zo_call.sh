OUT=gpurun_out/r03zo; mkdir -p $OUT
export TMPDIR=/tmp
date > $OUT/summary.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/summary.txt )
( timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 800 > $OUT/tests.log 2>&1; echo "full GPU suite: exit $? -- $(tail -1 $OUT/tests.log)" | tee -a $OUT/summary.txt )
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt )
python - <<'PY' | tee -a gpurun_out/r03zo/summary.txt
import json
d=json.load(open("gpurun_out/r03zo/bench.json"))
print("value %.1f M/s ms/step %.3f frac %.3f traffic %.3f GB kernel %s" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"]/1e9, d["roofline"]["kernel"]))
for k in ("sustained","alt_input_u8","alt_input_bits"):
    print("  ", k, d[k].get("value"))
print("   standalone", d["roofline_standalone"]["ms_per_step"], d["roofline_standalone"]["pre_pass"]["avg_launch_ms"], d["roofline_standalone"]["contraction"]["avg_launch_ms"])
PY
date >> $OUT/summary.txt
