"""Probe of pcoa_accumulate_calls_ex: every memory kind against the dense path, several orders (not a test: prints)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
P = importlib.import_module("spark-examples_amd")
n, v = 2504, int(sys.argv[1]) if len(sys.argv) > 1 else 300000
torch.manual_seed(1)
x = (torch.rand((v, n), device="cuda") < 0.13).to(torch.float32)
with P.PcoaEngine(n) as e:
    e.accumulate_dense(x); want = e.gram()
cols, cnt = [], torch.zeros(v, dtype=torch.int64, device="cuda")
for r0 in range(0, v, 1 << 17):
    nzr = x[r0:r0 + (1 << 17)] != 0
    cnt[r0:r0 + nzr.shape[0]] = nzr.sum(1)
    cols.append(nzr.nonzero()[:, 1].to(torch.int32))
idx_dev = torch.cat(cols)
offs_dev = torch.zeros(v + 1, dtype=torch.int64, device="cuda"); offs_dev[1:] = torch.cumsum(cnt, 0)
idx_cpu, offs_cpu = idx_dev.cpu(), offs_dev.cpu()
idx_pin, offs_pin = idx_cpu.pin_memory(), offs_cpu.pin_memory()
with P.PcoaEngine(n) as e:
    for name, ti, to in [("pageable", idx_cpu, offs_cpu), ("pinned", idx_pin, offs_pin), ("device", idx_dev, offs_dev),
                         ("pageable", idx_cpu, offs_cpu), ("pageable", idx_cpu, offs_cpu), ("pinned", idx_pin, offs_pin)]:
        e.reset(); e.sync()
        t0 = time.perf_counter()
        e.accumulate_calls_tensors(ti, to)
        g = e.gram()
        dt = time.perf_counter() - t0
        d = int((g != want).sum())
        print("%-9s %.4f s  mismatching entries %d  max abs diff %d  chunks %d" % (name, dt, d, int(np.abs(g - want).max()), e.timings()["csr_fast_chunks"]))
