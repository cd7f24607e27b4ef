#!/usr/bin/env python3
"""Where a resident fp32 batch lies and what the pipelined step costs on it (not a test: prints).  The step of bench.py is
period-5 in its five resident batches (profiles/r06y kernel trace: 1.87 / 1.87 / 1.87 / 2.02 / 2.12 ms of ring pre-pass): the
same kernels, another 10-GB region of HBM.  This probe times 30 pipelined steps on EACH batch of several allocation layouts.
usage: tools/placement_probe.py [steps per batch]"""
import importlib, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = importlib.import_module("spark-examples_amd")
synth = importlib.import_module("spark-examples_amd.synth")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n, v, seed, nb = 2504, 1000000, 1002, 5
dev = torch.device("cuda:0")
offs = synth.pop_offsets(n)


def fill(eng, t, first):
    for v0 in range(0, v, 1 << 18):
        v1 = min(v, v0 + (1 << 18))
        eng.synth_fill(seed, offs, synth.thresholds(seed, first + v0, v1 - v0), first + v0, t[v0:v1].data_ptr(), n)
    eng.sync()


def time_batch(eng, t):
    eng.reset()
    for _ in range(4):
        eng.accumulate_dense(t)
    eng.finalize(); eng.sync(); eng.reset(); eng.reset_timings(); eng.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.accumulate_dense(t)
    eng.finalize(); eng.sync()
    dt = (time.perf_counter() - t0) / steps
    tm = eng.timings()
    return 1e3 * dt, 1e3 * tm["pack_seconds"] / steps, 1e3 * tm["gram_kernel_seconds"] / steps


with P.PcoaEngine(n, device=0) as eng:
    eng.reserve(v, 2)
    layouts = []
    big = torch.empty((nb * v, n), dtype=torch.float32, device=dev)
    layouts.append(("one 50-GB tensor, batches = consecutive slices", [big[k * v:(k + 1) * v] for k in range(nb)], big))
    for name, parts, keep in layouts:
        for k, t in enumerate(parts):
            fill(eng, t, k * v)
        print("== " + name)
        for rnd in range(2):
            for k, t in enumerate(parts):
                print("  batch %d  ptr %#x (mod 2 MiB %7d)  step %.3f ms  pre-pass %.3f  contraction %.3f" % ((k, t.data_ptr(), t.data_ptr() % (2 << 20)) + time_batch(eng, t)), flush=True)
    del layouts, big
    torch.cuda.empty_cache()
    sep = [torch.empty((v, n), dtype=torch.float32, device=dev) for _ in range(nb)]
    for k, t in enumerate(sep):
        fill(eng, t, k * v)
    print("== five separate 10-GB tensors")
    for rnd in range(2):
        for k, t in enumerate(sep):
            print("  batch %d  ptr %#x (mod 2 MiB %7d)  step %.3f ms  pre-pass %.3f  contraction %.3f" % ((k, t.data_ptr(), t.data_ptr() % (2 << 20)) + time_batch(eng, t)), flush=True)
    del sep
    torch.cuda.empty_cache()
    # rows padded to a 2-MiB multiple per batch: every batch starts on a 2-MiB boundary of one allocation
    rows_pad = ((v * n * 4 + (2 << 20) - 1) // (2 << 20)) * (2 << 20) // 4
    flat = torch.empty(nb * rows_pad, dtype=torch.float32, device=dev)
    al = [flat[k * rows_pad:k * rows_pad + v * n].view(v, n) for k in range(nb)]
    for k, t in enumerate(al):
        fill(eng, t, k * v)
    print("== one tensor, every batch on a 2-MiB boundary")
    for k, t in enumerate(al):
        print("  batch %d  ptr %#x (mod 2 MiB %7d)  step %.3f ms  pre-pass %.3f  contraction %.3f" % ((k, t.data_ptr(), t.data_ptr() % (2 << 20)) + time_batch(eng, t)), flush=True)
