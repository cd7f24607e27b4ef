OUT=gpurun_out/r03ze; mkdir -p $OUT
export TMPDIR=/tmp
date > $OUT/summary.txt
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 800 --durations=8 -x > $OUT/tests.log 2>&1; echo "tests exit $?" | tee -a $OUT/summary.txt )
tail -15 $OUT/tests.log | tee -a $OUT/summary.txt
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt )
python - <<'PY' | tee -a gpurun_out/r03ze/summary.txt
import json
d=json.load(open("gpurun_out/r03ze/bench.json"))
print("value %.1f M/s ms/step %.3f frac %.3f traffic %s kernel %s" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["kernel"]))
print("sustained", d.get("sustained", {}).get("value"))
for k in ("alt_input_u8","alt_input_bits","config2_one_gpu_bits","roofline_standalone"):
    v=d.get(k,{}); print(k, {kk:v[kk] for kk in v if kk in ("value","ms_per_step","pack_ms_per_step","gram_ms_per_step","expand_ms_per_step","gram_variants_per_s","variants_per_s")})
PY
date >> $OUT/summary.txt
