#!/usr/bin/env python3
"""BASELINE configs[3]: biobank-scale N on ONE MI355X -- synthetic 100,000 samples x V variants.

The reference cannot run this at all (Breeze DenseMatrix[Int] needs N^2 < 2^31, N <= 46,340; MLlib RowMatrix
N <= 65,535 -- BASELINE.md section 1).  Here S (int32, 40 GB) and the operand workspace live in one 288 GB HBM; the
centred matrix B (80 GB in fp64) is never written: the Lanczos matvec evaluates it on the fly from S; genotypes are generated on the device chunk by chunk (the fp32 input,
400 GB for 10^6 variants, never exists as a whole).

Checks (no CPU oracle can hold this):
  * the top-left 2504 x 2504 block of S equals, bit for bit, the S of an independent N = 2504 engine fed the same
    variants restricted to the first 2504 samples (same Philox counters, all of them in population 0);
  * two far-apart off-diagonal blocks are each other's transpose (the mirror of the computed triangle);
  * the eigenpairs come back only after the engine's own on-device residual test ||B u - theta u|| passed.
Usage: python tools/config4_biobank.py [--samples 100000] [--variants 1000000]
Not part of pytest: it needs ~50 GB of HBM and a few seconds.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=100000)
    ap.add_argument("--variants", type=int, default=1000000)
    ap.add_argument("--seed", type=int, default=1004)
    ap.add_argument("--chunk", type=int, default=65536)
    args = ap.parse_args()
    P = importlib.import_module("spark-examples_amd")
    synth = importlib.import_module("spark-examples_amd.synth")
    n, v, seed = args.samples, args.variants, args.seed
    offs = synth.pop_offsets(n)
    n_small = 2504
    assert offs[1] >= n_small, "the first population must cover the spot-check block"
    offs_small = np.array([0, n_small], dtype=np.int32)

    t0 = time.perf_counter()
    eng = P.PcoaEngine(n)
    small = P.PcoaEngine(n_small)
    name, cus = eng.device_info()
    t_host = 0.0
    for v0 in range(0, v, args.chunk):
        cnt = min(args.chunk, v - v0)
        th0 = time.perf_counter()
        thr = synth.thresholds(seed, v0, cnt)
        t_host += time.perf_counter() - th0
        eng.accumulate_synthetic(seed, offs, thr, v0)
        small.accumulate_synthetic(seed, offs_small, np.ascontiguousarray(thr[:, :1]), v0)
    eng.finalize()
    eng.sync()
    t_gram_wall = time.perf_counter() - t0
    tim = eng.timings()

    blk = eng.gram_block(0, 0, n_small, n_small)
    ok_block = bool(np.array_equal(blk, small.gram()))
    small.close()
    a = eng.gram_block(10, n - 300, 200, 256)
    b = eng.gram_block(n - 300, 10, 256, 200)
    ok_mirror = bool(np.array_equal(a, b.T)) and int(a.sum()) > 0
    diag = eng.gram_block(n - 64, n - 64, 64, 64)
    ok_diag = bool(np.array_equal(diag, diag.T)) and bool((np.diag(diag) >= diag.max(axis=1)).all())

    t1 = time.perf_counter()
    comps, lam, nz = eng.compute(2)
    t_pcoa = time.perf_counter() - t1
    tim2 = eng.timings()
    # the two forms of the eigensolver's mat-vec on this S (pcoa_debug_centred_matvec; the vector crosses PCIe, 0.8 MB at N = 10^5)
    xv = np.random.default_rng(1).standard_normal(n)
    mv = {}
    for form in (0, 1):
        eng.debug_centred_matvec(xv, form)
        tm = time.perf_counter()
        for _ in range(3):
            yv = eng.debug_centred_matvec(xv, form)
        mv[form] = ((time.perf_counter() - tm) / 3.0, yv)
    mv_diff = float(np.abs(mv[0][1] - mv[1][1]).max() / np.abs(mv[0][1]).max())
    out = {
        "workload": "biobank scale (configs[3] = 100,000 x 10^6; configs[4] sample count = 250,000): synthetic %d samples x %d variants, 1x MI355X (Gram + eig on one GPU)" % (n, v),
        "device": name, "cu_count": cus,
        "gram_wall_s": t_gram_wall, "host_threshold_generation_s": t_host,
        "gram_kernel_s": tim["gram_kernel_seconds"], "pack_s": tim["pack_seconds"], "synth_s": tim["synth_seconds"],
        "gram_launches": tim["gram_kernel_launches"],
        "variants_per_s_kernels": v / (tim["gram_kernel_seconds"] + tim["pack_seconds"]),
        "algorithmic_pops": 2.0 * v * n * n / tim["gram_kernel_seconds"] / 1e15,
        "pcoa_wall_s": t_pcoa, "pcoa_method": tim2["eig_method"], "lanczos_steps": tim2["lanczos_steps"],
        "matvec_row_form_s": mv[0][0], "matvec_upper_triangle_form_s": mv[1][0], "matvec_forms_max_rel_diff": mv_diff,
        "matvec_note": "row form reads 4 N^2 bytes, upper-triangle form 2 N^2 + tile sums; both include centring, H2D / D2H of the vector",
        "eigenvalues": [float(x) for x in lam], "nonzero_rows": int(nz),
        "unit_norm": [float(np.linalg.norm(comps[:, c])) for c in range(2)],
        "orthogonality": float(abs(comps[:, 0] @ comps[:, 1])),
        "check_block_vs_independent_engine": ok_block, "check_mirror": ok_mirror, "check_diagonal": ok_diag,
    }
    print(json.dumps(out))
    eng.close()
    return 0 if (ok_block and ok_mirror and ok_diag) else 1


if __name__ == "__main__":
    sys.exit(main())
