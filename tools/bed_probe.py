#!/usr/bin/env python3
"""Probe of pcoa_accumulate_plink_bed from page-locked memory (not a test: prints): per-call time of 65,536-row blocks, the raw
H2D rate of the same block, and both again while host threads copy into the other block (what the streaming reader's preads do).
usage: tools/bed_probe.py [blocks]"""
import ctypes, importlib, os, sys, threading, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = importlib.import_module("spark-examples_amd")
L = importlib.import_module("spark-examples_amd._lib")
lib = L.load()
n, rows = 2504, 65536
bpv = (n + 3) // 4
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nbytes = rows * bpv
rng = np.random.default_rng(3)
src = rng.integers(0, 256, size=nbytes, dtype=np.uint8)
src |= 0xAA  # mostly hom-A2 / het codes: a sparse-ish cohort (bit pairs 10 / 11)
pins = []
for _ in range(2):
    p = ctypes.c_void_p()
    assert lib.pcoa_host_alloc_pinned(nbytes, ctypes.byref(p)) == 0
    ctypes.memmove(p, src.ctypes.data, nbytes)
    pins.append(p)
pageable = [np.ascontiguousarray(src.copy()) for _ in range(2)]
hip = ctypes.CDLL("libamdhip64.so")
dst = torch.empty(nbytes, dtype=torch.uint8, device="cuda")


def raw_h2d(tag):
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(5):
        t0 = time.perf_counter()
        rc = hip.hipMemcpy(ctypes.c_void_p(dst.data_ptr()), pins[0], ctypes.c_size_t(nbytes), 1)
        dt = time.perf_counter() - t0
        assert rc == 0
        best = max(best, nbytes / dt / 1e9)
    print("%-28s raw hipMemcpy H2D of one block (%.1f MB): %.1f GB/s" % (tag, nbytes / 1e6, best), flush=True)


def feed(tag, bufs, pinned=True):
    with P.PcoaEngine(n) as e:
        for w in range(2):  # warm: allocations
            lib.pcoa_accumulate_plink_bed(e._ctx, bufs[w] if pinned else ctypes.c_void_p(bufs[w].ctypes.data), rows, bpv, 0, 0)
        e.finalize(); e.sync(); e.reset(); e.sync()
        per = []
        t0 = time.perf_counter()
        for b in range(blocks):
            t1 = time.perf_counter()
            rc = lib.pcoa_accumulate_plink_bed(e._ctx, bufs[b & 1] if pinned else ctypes.c_void_p(bufs[b & 1].ctypes.data), rows, bpv, 0, 0)
            assert rc == 0
            per.append(time.perf_counter() - t1)
        e.finalize(); e.sync()
        dt = time.perf_counter() - t0
        per = np.array(per) * 1e3
        print("%-28s %d blocks: %.1f M variants/s (%.1f GB/s of rows), per call median %.3f ms, max %.3f ms" %
              (tag, blocks, blocks * rows / dt / 1e6, blocks * nbytes / dt / 1e9, np.median(per), per.max()), flush=True)


raw_h2d("quiet host")
feed("pinned, quiet host", pins)
feed("pageable, quiet host", pageable, pinned=False)
stop = False
scratch = [np.empty(nbytes // 8, dtype=np.uint8) for _ in range(8)]


def churn(i):
    while not stop:
        np.copyto(scratch[i], src[i * (nbytes // 8):(i + 1) * (nbytes // 8)])


th = [threading.Thread(target=churn, args=(i,)) for i in range(8)]
for t in th:
    t.start()
raw_h2d("8 host threads copying")
feed("pinned, 8 threads copying", pins)
stop = True
for t in th:
    t.join()

# the streaming reader's own pattern: while block b is fed from one page-locked buffer, 8 threads pread the next block out of the
# page cache into the OTHER page-locked buffer
path = "/tmp/bed_probe.bin"
with open(path, "wb") as f:
    for _ in range(4):
        f.write(src.tobytes())
fd = os.open(path, os.O_RDONLY)
views = [memoryview((ctypes.c_ubyte * nbytes).from_address(p.value)).cast("B") for p in pins]


def pread_into(which, nthreads=8):
    per = (nbytes + nthreads - 1) // nthreads
    def one(i):
        lo, hi = i * per, min(nbytes, (i + 1) * per)
        while lo < hi:
            lo += os.preadv(fd, [views[which][lo:hi]], lo)
    ts = [threading.Thread(target=one, args=(i,)) for i in range(nthreads)]
    for t in ts:
        t.start()
    return ts


for nthreads in (8, 4):
    with P.PcoaEngine(n) as e:
        for w in range(2):
            lib.pcoa_accumulate_plink_bed(e._ctx, pins[w], rows, bpv, 0, 0)
        e.finalize(); e.sync(); e.reset(); e.sync()
        per, rd = [], []
        t0 = time.perf_counter()
        for b in range(blocks):
            ts = pread_into((b & 1) ^ 1, nthreads)
            t1 = time.perf_counter()
            assert lib.pcoa_accumulate_plink_bed(e._ctx, pins[b & 1], rows, bpv, 0, 0) == 0
            t2 = time.perf_counter()
            for t in ts:
                t.join()
            per.append(t2 - t1); rd.append(time.perf_counter() - t2)
        e.finalize(); e.sync()
        dt = time.perf_counter() - t0
        per, rd = np.array(per) * 1e3, np.array(rd) * 1e3
        print("fed beside %d pread threads    %d blocks: %.1f M variants/s, feed call median %.3f ms (max %.3f), read not hidden median %.3f ms" %
              (nthreads, blocks, blocks * rows / dt / 1e6, np.median(per), per.max(), np.median(rd)), flush=True)
os.close(fd)
os.unlink(path)
