mkdir -p gpurun_out/r03l
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_guard.py tests/test_gpu_concurrency.py -m gpu -q -p no:cacheprovider --timeout 800 --durations=10 > gpurun_out/r03l/new_tests.log 2>&1; echo "exit $?" >> gpurun_out/r03l/new_tests.log )
tail -25 gpurun_out/r03l/new_tests.log
( timeout 900 python -m pytest tests/test_gpu_baseline_sizes.py -m gpu -q -p no:cacheprovider --timeout 800 -k "config2_full_size" --durations=5 > gpurun_out/r03l/config2.log 2>&1; echo "exit $?" >> gpurun_out/r03l/config2.log )
tail -15 gpurun_out/r03l/config2.log
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03l/bench.json 2> gpurun_out/r03l/bench.err; echo "bench exit $?" )
tail -5 gpurun_out/r03l/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03l/bench.json"))
print("value %.1f M/s ms/step %.3f frac %.3f step_hbm_frac %.3f pcoa %.3f ms" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["frac"], d["step_hbm_frac"], d["pcoa_wall_ms"]))
for k in ("sustained","alt_input_u8","alt_input_bits","config2_one_gpu_bits","roofline_standalone"):
    print(k, json.dumps(d.get(k))[:600])
PY
