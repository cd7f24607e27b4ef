import sys, time, importlib, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("spark-examples_amd"); synth = importlib.import_module("spark-examples_amd.synth")
n, v = 2504, 1 << 20
offs = synth.pop_offsets(n)
x = torch.empty((v, n), dtype=torch.float32, device="cuda")
eng = P.PcoaEngine(n)
for v0 in range(0, v, 1 << 18):
    eng.synth_fill(1002, offs, synth.thresholds(1002, v0, 1 << 18), v0, x[v0:v0 + (1 << 18)].data_ptr(), n)
eng.sync()
ref = None
for calls in (1, 4, 16, 64, 256):
    step = v // calls
    for rep in range(2):
        eng.reset(); eng.reset_timings(); eng.sync()
        t0 = time.perf_counter()
        for k in range(calls):
            eng.accumulate_dense(x[k * step:(k + 1) * step])
        eng.finalize(); eng.sync()
        dt = time.perf_counter() - t0
    s = eng.gram()
    if ref is None: ref = s
    t = eng.timings()
    print("calls %4d x %7d variants: %.3f ms total, %.1f M variants/s, gram launches %d (%.3f ms), pack %.3f ms, same S: %s"
          % (calls, step, 1e3 * dt, v / dt / 1e6, t["gram_kernel_launches"], 1e3 * t["gram_kernel_seconds"], 1e3 * t["pack_seconds"], np.array_equal(s, ref)))
