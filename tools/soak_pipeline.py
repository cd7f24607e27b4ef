"""Soak of the co-resident pipeline (on an MI355X): K steps of 10^6 resident variants in fp32 / uint8 / bitset form, alternating
between two batches, S compared bit for bit with K/2 x (S1 + S2) from a PCOA_FLAG_NO_PIPELINE engine.  Counts pass 2^24 and the
int32 -> int64 fold threshold many times over; a rare wrong accumulator (a hazard that only shows beside another kernel) would
show as a mismatch.  usage: soak_pipeline.py [steps per format, default 2000]"""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("spark-examples_amd"); synth = importlib.import_module("spark-examples_amd.synth")
ingest = importlib.import_module("spark-examples_amd.ingest")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n, v = 2504, 1000000
offs = synth.pop_offsets(n)
dev = torch.device("cuda", 0)
xs = []
with P.PcoaEngine(n, pipeline=False) as ref:
    s_ref = []
    for b in range(2):
        x = torch.empty((v, n), dtype=torch.float32, device=dev)
        for v0 in range(0, v, 1 << 18):
            v1 = min(v, v0 + (1 << 18))
            ref.synth_fill(1002, offs, synth.thresholds(1002, b * v + v0, v1 - v0), b * v + v0, x[v0:v1].data_ptr(), n)
        ref.sync()
        xs.append(x)
        ref.reset(); ref.accumulate_dense(x); s_ref.append(ref.gram().astype(np.int64))
want_pair = s_ref[0] + s_ref[1]
x8 = [x.to(torch.uint8) for x in xs]
bits = []
for x in x8:
    words = (n + 31) // 32
    b = torch.empty((v, words), dtype=torch.int32, device=dev)
    wts = (1 << torch.arange(32, device=dev, dtype=torch.int64))
    for r0 in range(0, v, 1 << 16):
        xb = torch.nn.functional.pad(x[r0:r0 + (1 << 16)] > 0, (0, words * 32 - n))
        val = (xb.view(-1, words, 32).to(torch.int64) * wts).sum(dim=2)
        b[r0:r0 + val.shape[0]] = torch.where(val >= 2 ** 31, val - 2 ** 32, val).to(torch.int32)
    bits.append(b)
torch.cuda.synchronize()
ok = True
with P.PcoaEngine(n) as eng:
    for name, feed in (("fp32", lambda i: eng.accumulate_dense(xs[i & 1])), ("uint8", lambda i: eng.accumulate_dense_u8(x8[i & 1])),
                       ("bitsets", lambda i: eng.accumulate_bits(bits[i & 1])),
                       ("mixed", lambda i: (eng.accumulate_dense(xs[i & 1]) if i % 6 < 2 else eng.accumulate_dense_u8(x8[i & 1]) if i % 6 < 4 else eng.accumulate_bits(bits[i & 1])))):
        eng.reset(); eng.reset_timings(); eng.sync()
        t0 = time.perf_counter()
        for i in range(K):
            feed(i)
            if i % 64 == 63:
                eng.sync()          # releases the engine's references to the input tensors
        s = eng.gram()
        dt = time.perf_counter() - t0
        t = eng.timings()
        exact = bool(np.array_equal(s, (K // 2) * want_pair))
        ok = ok and exact
        print("%-8s %d steps x 10^6 variants in %.2f s (%.0f M variants/s): pipelined launches %d, S[0,0] = %d, largest entry %d, exact: %s"
              % (name, K, dt, K * v / dt / 1e6, t["pipeline_launches"], s[0, 0], s.max(), exact), flush=True)
print("SOAK", "ok" if ok else "FAILED")
sys.exit(0 if ok else 1)
