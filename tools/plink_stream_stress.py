#!/usr/bin/env python3
"""Stress of the streaming PLINK host (reader thread, four rotating page-locked blocks, queued feed): the same fileset through
many block sizes, engine counts and --references windows, every run's S compared with the first.  A race between the reader
and a copy still in flight, or a block released a call too early, shows up as a different S (or a hang: every run has a
timeout).  usage: tools/plink_stream_stress.py [variants] [samples] [rounds]"""
import os, subprocess, sys, time
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
v = int(sys.argv[1]) if len(sys.argv) > 1 else 600000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1001
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
d = "/tmp/plink_stress"
os.makedirs(d, exist_ok=True)
prefix = os.path.join(d, "cohort")
bpv = (n + 3) // 4
with open(prefix + ".fam", "w") as f:
    f.write("".join("F%d S%05d 0 0 0 -9\n" % (i, i) for i in range(n)))
with open(prefix + ".bim", "w") as f:
    f.write("".join("%d\trs%d\t0\t%d\tC\tA\n" % (1 + (k * 22) // v, k, 1000 + 10 * k) for k in range(v)))
g = torch.Generator(device="cuda").manual_seed(11)
with open(prefix + ".bed", "wb") as f:
    f.write(bytes([0x6c, 0x1b, 0x01]))
    for c0 in range(0, v, 1 << 16):
        rows = min(1 << 16, v - c0)
        f.write(torch.randint(0, 256, (rows, bpv), dtype=torch.uint8, device="cuda", generator=g).cpu().numpy().tobytes())
exe = os.path.join(ROOT, "spark-examples_amd", "variants_pca_driver")
dump = os.path.join(d, "s.bin")


def run(extra):
    res = subprocess.run([exe, "--input-path", prefix + ".bed", "--dump-similarity", dump] + extra, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, universal_newlines=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    return np.fromfile(dump, dtype="<i8")


cases = []
for refs in (["--all-references"], ["--references", "3:1:99999999,7:1:99999999,20:1:99999999"]):
    ref_s = run(refs + ["--no-stream"]) if v <= 200000 else run(refs + ["--stream-rows", "65536"])
    for r in range(rounds):
        for rows in (37, 1000, 4096, 65536, 131072, 300000):
            for eng in ([], ["--gpus", "2", "--gpu-map", "0,0"], ["--gpus", "3", "--gpu-map", "0,0,0"]):
                if rows < 1000 and (v > 100000):
                    continue            # (tiny blocks over a big file: minutes of launches, nothing new)
                t0 = time.perf_counter()
                s = run(refs + ["--stream-rows", str(rows)] + eng)
                ok = bool(np.array_equal(s, ref_s))
                cases.append(ok)
                print("%-18s rows %-7d engines %d  %.2f s  %s" % (refs[0][2:], rows, 1 + len(eng) // 2 if not eng else int(eng[1]),
                                                                  time.perf_counter() - t0, "same S" if ok else "DIFFERENT S"), flush=True)
print("%d runs, %d with the reference S" % (len(cases), sum(cases)))
sys.exit(0 if all(cases) else 1)
