"""Lanczos (default) PCoA path probe: wall-clock of pcoa_compute at N = 2504 (and optionally other N) over many calls.
usage: python tools/lanczos_probe.py [N ...]   (run under rocprofv3 --kernel-trace --stats for per-kernel durations)"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("spark-examples_amd"); synth = importlib.import_module("spark-examples_amd.synth")
for n in [int(a) for a in sys.argv[1:]] or [2504]:
    offs = synth.pop_offsets(n)
    with P.PcoaEngine(n) as eng:
        eng.accumulate_synthetic(1002, offs, synth.thresholds(1002, 0, 20000), 0)
        eng.finalize()
        walls = []
        for rep in range(30):
            eng.reset_timings()
            t0 = time.perf_counter()
            comps, lam, nz = eng.compute(2)
            walls.append(1e3 * (time.perf_counter() - t0))
            t = eng.timings()
        walls = np.array(walls[5:])
        print("N=%d lanczos wall ms: min %.3f median %.3f max %.3f; device total %.3f (center %.3f, lanczos %.3f, back %.3f), "
              "steps %d, method %d; lam %s" % (n, walls.min(), np.median(walls), walls.max(), 1e3 * t["compute_total_seconds"],
                                               1e3 * t["center_seconds"], 1e3 * t["lanczos_seconds"],
                                               1e3 * t["backtransform_seconds"], t["lanczos_steps"], t["eig_method"], lam))
