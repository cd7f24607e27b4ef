"""Several ctxs driven from several host threads at once (GPU only).

SURVEY 8(b): "one handle is used from one host thread at a time (handles are independent, so Spark tasks/threads may each
hold one)"; INTEGRATION.md deploys local[k] = k task threads in one JVM (the mapPartitions tasks of VariantsPca.scala:184
are threads).  ctypes releases the GIL around every C call, so the threads below really are inside libpcoa_hip.so together:
per-thread hipSetDevice, the library's statics (debug knobs, the lazily bound RCCL), event pools and side streams."""
import threading

import numpy as np
import pytest

from conftest import int_gram, load_pkg

pytestmark = pytest.mark.gpu

N_THREADS = 4


def _workload(seed):
    rng = np.random.default_rng(seed)
    n = [257, 1100, 640, 2504][seed % 4]
    parts = []
    for k in range(6):
        v = int(rng.integers(200, 5000))
        parts.append((rng.random((v, n)) < rng.choice([0.02, 0.2, 0.5])).astype(np.uint8))
    return n, parts


def _run(P, ingest, torch, seed, dev_tiles, out, barrier=None):
    """One task: its own ctx, every boundary interleaved, finalize, computePca, an error in the middle of it."""
    n, parts = _workload(seed)
    try:
        with P.PcoaEngine(n) as eng:
            if barrier is not None:
                barrier.wait()
            eng.accumulate_dense(dev_tiles[seed][0])                       # fp32 device tile
            eng.accumulate_dense_u8(parts[1])                              # uint8 host tile
            eng.accumulate_bits(ingest.pack_bits(parts[2]))                # host bitsets
            eng.accumulate_callsets([list(np.nonzero(r)[0]) for r in parts[3][:300]])
            with pytest.raises(P.IndexRangeError) as ei:                   # an error on THIS ctx only
                eng.accumulate_calls(np.array([0, n + seed], dtype=np.int32), np.array([0, 2], dtype=np.int64))
            assert str(n + seed) in str(ei.value)
            eng.accumulate_dense_u8(dev_tiles[seed][1])                    # uint8 device tile
            mid = eng.gram()                                               # a synchronising read in the middle
            eng.accumulate_dense(parts[5].astype(np.float32))              # fp32 host tile
            s = eng.gram()
            comps, lam, nz = eng.compute(2)
            out[seed] = (mid, s, comps, lam, nz)
    except BaseException as exc:  # noqa: BLE001  (reported by the main thread)
        out[seed] = exc


def test_four_threads_four_ctxs_match_the_sequential_run():
    import torch
    P = load_pkg()
    ingest = load_pkg("ingest")
    seeds = list(range(N_THREADS))
    dev_tiles = {}
    for s in seeds:
        n, parts = _workload(s)
        dev_tiles[s] = (torch.from_numpy(parts[0].astype(np.float32)).cuda(), torch.from_numpy(parts[4]).cuda())
    torch.cuda.synchronize()
    # sequential reference (same code, one task after the other)
    ref = {}
    for s in seeds:
        _run(P, ingest, torch, s, dev_tiles, ref)
        assert not isinstance(ref[s], BaseException), ref[s]
        n, parts = _workload(s)
        want = sum(int_gram(p) for p in (parts[0], parts[1], parts[2], parts[3][:300], parts[4], parts[5]))
        assert np.array_equal(ref[s][1], want)
    # the same four tasks at once, three rounds (different interleavings)
    for _ in range(3):
        got = {}
        barrier = threading.Barrier(N_THREADS)
        threads = [threading.Thread(target=_run, args=(P, ingest, torch, s, dev_tiles, got, barrier)) for s in seeds]
        for t in threads:
            t.start()
        for t in threads:
            t.join(600)
            assert not t.is_alive(), "a task hangs"
        for s in seeds:
            assert not isinstance(got[s], BaseException), "task %d: %r" % (s, got[s])
            mid, sg, comps, lam, nz = got[s]
            assert np.array_equal(mid, ref[s][0]) and np.array_equal(sg, ref[s][1]), "S of task %d differs from the sequential run" % s
            # the eigensolver is deterministic (fixed reduction orders): bit-identical, not just close
            assert np.array_equal(comps, ref[s][2]) and np.array_equal(lam, ref[s][3]) and nz == ref[s][4]
