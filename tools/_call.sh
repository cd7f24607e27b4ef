set +e
OUT=gpurun_out/r06i; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_guard.py tests/test_gpu_multi_engine.py tests/test_gpu_baseline_sizes.py -m gpu -q -p no:cacheprovider --timeout 400 -x -k "guard or unmapped or plink or bed or jni or engines" > $OUT/tests.log 2>&1
echo "tests exit $?" > $OUT/summary.txt; tail -4 $OUT/tests.log >> $OUT/summary.txt
timeout 300 python - >> $OUT/summary.txt 2>&1 <<'PY'
import importlib, sys, time, numpy as np, torch
sys.path.insert(0, '.')
P = importlib.import_module("spark-examples_amd")
n, v = 2504, 1000000
bpv = (n + 3) // 4
g = torch.Generator(device="cuda").manual_seed(1)
raw = torch.randint(0, 256, (v, bpv), dtype=torch.uint8, device="cuda", generator=g) | 0xAA
with P.PcoaEngine(n) as e:
    e.reserve(v, 0)
    for rep in range(3):
        e.reset(); e.reset_timings(); e.sync()
        t0 = time.perf_counter(); e.accumulate_plink_bed(raw); e.finalize(); e.sync(); dt = time.perf_counter() - t0
        t = e.timings()
        print("device rows: %.1f M variants/s, decode %.3f ms, transpose %.3f ms, contraction %.3f ms" % (v / dt / 1e6, 1e3 * t["densify_seconds"], 1e3 * t["pack_seconds"], 1e3 * t["gram_kernel_seconds"]))
PY
cat $OUT/summary.txt
