#!/usr/bin/env python3
"""uint8 and bitset device tiles (configs[1] cohort): the co-resident pipeline against the serial order (PCOA_FLAG_NO_PIPELINE),
same process, interleaved.  usage: tools/alt_inputs_ab.py [steps]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = importlib.import_module("spark-examples_amd")
synth = importlib.import_module("spark-examples_amd.synth")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n, v, seed = 2504, 1000000, 1002
dev = torch.device("cuda:0")
with P.PcoaEngine(n, device=0) as e0:
    x = torch.empty((v, n), dtype=torch.float32, device=dev)
    po = synth.pop_offsets(n)
    for v0 in range(0, v, 1 << 18):
        v1 = min(v, v0 + (1 << 18))
        e0.synth_fill(seed, po, synth.thresholds(seed, v0, v1 - v0), v0, x[v0:v1].data_ptr(), n)
    e0.sync()
x8 = x.to(torch.uint8)
words = (n + 31) // 32
bits = torch.empty((v, words), dtype=torch.int32, device=dev)
wts = (1 << torch.arange(32, device=dev, dtype=torch.int64))
for r0 in range(0, v, 1 << 16):
    xb = torch.nn.functional.pad(x[r0:r0 + (1 << 16)] > 0, (0, words * 32 - n))
    val = (xb.view(-1, words, 32).to(torch.int64) * wts).sum(dim=2)
    bits[r0:r0 + val.shape[0]] = torch.where(val >= 2 ** 31, val - 2 ** 32, val).to(torch.int32)
del x, xb, val
torch.cuda.synchronize()


def run(eng, feed, tile):
    eng.reset()
    for _ in range(2):
        feed(tile)
    eng.finalize(); eng.sync(); eng.reset(); eng.reset_timings(); eng.sync()
    t = time.perf_counter()
    for _ in range(steps):
        feed(tile)
    eng.finalize(); eng.sync()
    dt = time.perf_counter() - t
    tm = eng.timings()
    return v * steps / dt / 1e6, 1e3 * dt / steps, 1e3 * tm["pack_seconds"] / steps, 1e3 * tm["gram_kernel_seconds"] / steps, int(tm["pipeline_launches"])


# one engine alive at a time: an engine owns three streams, and HIP maps a process's streams onto a handful of hardware queues --
# with two engines alive the pre-pass stream and the contraction stream of one of them can share a queue and serialise (seen:
# uint8 tiles 1.62 ms per step with the two kernels back to back, r05d)
grams = {}
for rnd in range(2):
    for name, kw in (("pipeline", {}), ("serial", {"pipeline": False})):
        with P.PcoaEngine(n, device=0, **kw) as eng:
            for kind, tile in (("bits", bits), ("u8", x8)):
                feed = eng.accumulate_bits if kind == "bits" else eng.accumulate_dense_u8
                r = run(eng, feed, tile)
                print("%-8s %-4s %7.1f M variants/s  %.3f ms per step (pre-pass %.3f, contraction %.3f, pipelined launches %d)" % ((name, kind) + r), flush=True)
            grams[name] = eng.gram()
print("S equal:", bool(np.array_equal(grams["pipeline"], grams["serial"])))
