"""Dense (Householder) eigensolver probe: wall-clock and stage breakdown at N = 2504 (and optionally another N).
usage: python tools/hh_probe.py [N ...]   (run under rocprofv3 --kernel-trace --stats for per-kernel durations)"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("spark-examples_amd"); synth = importlib.import_module("spark-examples_amd.synth")
VARIANTS = int(os.environ.get("HH_PROBE_VARIANTS", "20000"))
for n in [int(a) for a in sys.argv[1:]] or [2504]:
    offs = synth.pop_offsets(n)
    with P.PcoaEngine(n, eig="householder") as eng:
        eng.accumulate_synthetic(1002, offs, synth.thresholds(1002, 0, VARIANTS), 0)
        eng.finalize()
        for rep in range(int(os.environ.get("HH_PROBE_REPS", "3"))):
            eng.reset_timings()
            t0 = time.perf_counter()
            comps, lam, nz = eng.compute(2)
            dt = time.perf_counter() - t0
            t = eng.timings()
            if os.environ.get("HH_PROBE_EVERY_CALL"):
                print("   call %d: wall %.2f ms, tridiag %.2f, eig %.2f, back %.2f" % (
                    rep, 1e3 * dt, 1e3 * t["tridiag_seconds"], 1e3 * t["eig_seconds"], 1e3 * t["backtransform_seconds"]))
        print("N=%d householder wall %.2f ms: tridiag %.2f, eig %.2f, back %.2f, center %.3f; lam %s" % (
            n, 1e3 * dt, 1e3 * t["tridiag_seconds"], 1e3 * t["eig_seconds"], 1e3 * t["backtransform_seconds"],
            1e3 * t["center_seconds"], lam))
    with P.PcoaEngine(n) as eng2:
        eng2.accumulate_synthetic(1002, offs, synth.thresholds(1002, 0, VARIANTS), 0)
        c2, l2, _ = eng2.compute(2)
        print("   vs Lanczos path: max |dlam|/lam %.2e, max vector diff %.2e" % (
            float(np.max(np.abs(l2 - lam) / np.abs(lam))),
            float(max(np.linalg.norm(comps[:, c] - c2[:, c] * np.sign(comps[:, c] @ c2[:, c])) for c in range(2)))))
