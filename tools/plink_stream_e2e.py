#!/usr/bin/env python3
"""End-to-end run of the compiled host on a synthetic whole-cohort PLINK fileset: V variants x N samples
written as .bed/.bim/.fam (genotype codes drawn on the GPU, ~14 % carriers), then
  variants_pca_driver --input-path <prefix>.bed [--gpus k --gpu-map ...]
with the device decode and with the host decode; prints each run's own report (variants/s of ingest -> S, peak RSS) and checks
that both give the same S and the same coordinates.  usage: tools/plink_stream_e2e.py [V] [N] [dir] ["extra host args"]"""
import os, subprocess, sys, time
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
v = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2504
d = sys.argv[3] if len(sys.argv) > 3 else "/tmp/plink_e2e"
host_extra = sys.argv[4].split() if len(sys.argv) > 4 else []
os.makedirs(d, exist_ok=True)
prefix = os.path.join(d, "cohort")
bpv = (n + 3) // 4
t0 = time.perf_counter()
with open(prefix + ".fam", "w") as f:
    f.write("".join("F%d S%05d 0 0 0 -9\n" % (i, i) for i in range(n)))
with open(prefix + ".bim", "w") as f:
    for c0 in range(0, v, 100000):
        f.write("".join("%d\trs%d\t0\t%d\tC\tA\n" % (1 + (k * 22) // v, k, 1000 + 10 * k) for k in range(c0, min(v, c0 + 100000))))
g = torch.Generator(device="cuda").manual_seed(7)
with open(prefix + ".bed", "wb") as f:
    f.write(bytes([0x6c, 0x1b, 0x01]))
    for c0 in range(0, v, 1 << 16):
        rows = min(1 << 16, v - c0)
        u = torch.rand((rows, bpv * 4), device="cuda", generator=g)
        p = torch.rand((rows, 1), device="cuda", generator=g) * 0.28          # per-variant carrier rate 0 .. 28 %
        codes = torch.full((rows, bpv * 4), 3, dtype=torch.uint8, device="cuda")   # hom A2 = reference
        codes[u < p] = 2                                                       # het
        codes[u < p * 0.15] = 0                                                # hom A1 (non-reference)
        codes[u > 0.995] = 1                                                   # missing
        q = codes.view(rows, bpv, 4)
        f.write((q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).cpu().numpy().tobytes())
print("wrote %s.bed: %d variants x %d samples, %.1f MB, in %.1f s" % (prefix, v, n, (3 + v * bpv) / 1e6, time.perf_counter() - t0))
exe = os.path.join(ROOT, "spark-examples_amd", "variants_pca_driver")
outs = {}
for tag, extra in (("device decode", []), ("device decode again (page cache warm)", []), ("host decode", ["--plink-decode", "host"]),
                   ("two engines on one GPU", ["--gpus", "2", "--gpu-map", "0,0"])):
    t1 = time.perf_counter()
    res = subprocess.run([exe, "--input-path", prefix + ".bed", "--all-references", "--dump-similarity", os.path.join(d, "s.bin")] + host_extra + extra,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    wall = time.perf_counter() - t1
    assert res.returncode == 0, res.stderr
    s = np.fromfile(os.path.join(d, "s.bin"), dtype="<i8")
    outs[tag] = (s, res.stdout)
    print("[%s] whole process %.2f s | %s" % (tag, wall, " | ".join(l for l in res.stderr.splitlines() if "Streamed" in l or "Variants accumulated" in l)))
ref = outs["device decode"]
for tag, (s, out) in outs.items():
    print("[%s] S == device-decode S: %s; printed coordinates identical: %s" % (tag, bool(np.array_equal(s, ref[0])), out == ref[1]))
