"""Guard-page sweep (run as a child process by tests/test_gpu_guard.py with PCOA_DEBUG_GUARD=1 or 2 in the environment).

Every device buffer of libpcoa_hip.so -- and, through pcoa_debug_alloc, every INPUT tile handed to it here -- is then a
virtual range of its own whose end (mode 1) or start (mode 2) lies against a page that is never mapped.  A kernel that
reads or writes one element too far dies with "Memory access fault by GPU" on the spot; the last "case ..." line printed
says which boundary of which shape it was (AMD_SERIALIZE_KERNEL=3 / HIP_LAUNCH_BLOCKING=1 keep the kernels one at a time).
Results are still compared with an integer matmul, so an out-of-bounds read that happened to land in mapped memory and
changed S is caught as well.

usage: guard_sweep.py <n_cases> <first_seed> [loops] [index of the first case] [ring]

"ring": the co-resident fp32 pipeline instead -- shapes where it is on (N > 1024, stride a multiple of 4, several
accumulate calls per job), meant to run with PCOA_DEBUG_MAX_LAUNCH small so that operand buffers fill, alternate and the
persistent ring pre-pass (pack_kbits_ring_kernel) runs beside a contraction: its row clamp, its look-ahead into the wave's
next unit and its column offsets all sit against unmapped pages then.
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from conftest import int_gram, load_pkg  # noqa: E402

P = load_pkg()
L = load_pkg("_lib")
ingest = load_pkg("ingest")
lib = L.load()
hip = ctypes.CDLL("libamdhip64.so")  # the runtime torch / libpcoa_hip already mapped
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipMemcpy.restype = ctypes.c_int
hip.hipDeviceSynchronize.restype = ctypes.c_int


class DevBuf(object):
    """Host array -> exactly-sized device allocation from the library's (guarded) allocator."""

    def __init__(self, a):
        a = np.ascontiguousarray(a)
        self.ptr = ctypes.c_void_p()
        rc = lib.pcoa_debug_alloc(0, a.nbytes, ctypes.byref(self.ptr))
        assert rc == 0, lib.pcoa_last_error(None)
        assert hip.hipMemcpy(self.ptr, a.ctypes.data, a.nbytes, 1) == 0
        # a "synchronous" copy into a virtual-memory mapping was seen to return before the data is visible to kernels on
        # the engine's (non-blocking) stream: the first guarded sweeps read partly copied tiles.  (The library's own copies
        # are stream-ordered with the kernels that consume them.)
        assert hip.hipDeviceSynchronize() == 0

    def free(self):
        lib.pcoa_debug_free(self.ptr)


def case(seed):
    rng = np.random.default_rng(seed)
    edges_n = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513, 1025]
    edges_v = [1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 1023, 1025]
    n = int(rng.choice(edges_n)) if rng.random() < 0.5 else int(rng.integers(1, 1300))
    v = int(rng.choice(edges_v)) if rng.random() < 0.5 else int(rng.integers(1, 4000))
    dens = float(rng.choice([0.01, 0.1, 0.3, 0.6, 0.97]))
    x = (rng.random((v, n)) < dens).astype(np.uint8)
    return rng, n, v, x


def check(eng, want, what):
    got = eng.gram()
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        raise SystemExit("MISMATCH %s: %d entries, first %s got %d want %d" % (
            what, len(bad), bad[0], got[tuple(bad[0])], want[tuple(bad[0])]))


def one_case(seed, idx):
    rng, n, v, x = case(seed)
    want = int_gram(x)
    pad = int(rng.integers(0, 9))
    kernel = ["auto", "auto", "fp4", "i8", "f32"][idx % 5]
    operand = "fp4" if idx % 3 == 2 else "bits"
    mult = (idx % 7 == 3) and kernel in ("auto", "i8")   # carrier multiplicities: the int8 path / the auto fallback
    print("case seed=%d n=%d v=%d pad=%d kernel=%s operand=%s mult=%s" % (seed, n, v, pad, kernel, operand, mult), flush=True)
    xm = x.astype(np.int64)
    if mult:
        xm = xm * rng.integers(1, 6, size=x.shape)
        want = int_gram(xm)
    with P.PcoaEngine(n, gram_kernel=kernel, operand=operand) as eng:
        ctx = eng._ctx
        # fp32 device tile, padded stride, NaN in the padding -- the allocation ends with the last row
        a = np.full((v, n + pad), np.nan, dtype=np.float32)
        a[:, :n] = xm
        d = DevBuf(a)
        eng._check(lib.pcoa_accumulate_dense_f32(ctx, d.ptr, v, n + pad, 1))
        check(eng, want, "f32 device")
        d.free()
        eng.reset()
        # uint8 device tile
        a8 = np.full((v, n + pad), 255, dtype=np.uint8)
        a8[:, :n] = xm
        d = DevBuf(a8)
        eng._check(lib.pcoa_accumulate_dense_u8(ctx, d.ptr, v, n + pad, 1))
        check(eng, want, "u8 device")
        d.free()
        eng.reset()
        # host tiles (staging buffers of the library)
        eng.accumulate_dense(xm.astype(np.float32))
        eng.accumulate_dense_u8(xm.astype(np.uint8))
        check(eng, 2 * want, "host tiles")
        eng.reset()
        if kernel != "f32" and not mult:
            # carrier bitsets: device (padded stride with garbage) and host
            bits = ingest.pack_bits(x, pad_words=pad % 3)
            if pad % 3:
                bits[:, (n + 31) // 32:] = 0xa5a5a5a5
            d = DevBuf(bits)
            eng._check(lib.pcoa_accumulate_bits(ctx, d.ptr, v, bits.shape[1], 1))
            eng.accumulate_bits(bits)
            check(eng, 2 * want, "bits")
            d.free()
            eng.reset()
            # r05: the rows of a PLINK .bed on the device, exactly sized (rows ceil(N / 4) [+ pad] bytes apart start at any
            # byte: the decode reads aligned 8-byte units around them) and from the host; carrier = heterozygous, the rest
            # homozygous A2 (reference) or missing
            bpv = (n + 3) // 4 + (pad % 4)
            codes = np.where(x > 0, 2, np.where(rng.random(x.shape) < 0.2, 1, 3)).astype(np.uint8)
            quad = np.full((v, bpv * 4), 3, dtype=np.uint8)
            quad[:, :n] = codes
            quad = quad.reshape(v, bpv, 4)
            raw = (quad[:, :, 0] | (quad[:, :, 1] << 2) | (quad[:, :, 2] << 4) | (quad[:, :, 3] << 6)).astype(np.uint8)
            d = DevBuf(raw)
            eng._check(lib.pcoa_accumulate_plink_bed(ctx, d.ptr, v, bpv, 0, 1))
            eng.accumulate_plink_bed(raw)
            check(eng, 2 * want, "plink rows")
            d.free()
            eng.reset()
            # r05: carrier lists in exactly-sized DEVICE arrays (the LDS scatter streams a block's entries with 16-byte loads
            # from whatever alignment they start at)
            rows_, cols_ = np.nonzero(x)
            offs = np.zeros(v + 1, dtype=np.int64)
            np.cumsum(np.bincount(rows_, minlength=v), out=offs[1:])
            if cols_.size:
                di, do = DevBuf(cols_.astype(np.int32)), DevBuf(offs)
                eng._check(lib.pcoa_accumulate_calls_ex(ctx, di.ptr, do.ptr, v, L.PCOA_CALLS_DEVICE_PTR))
                check(eng, want, "csr device arrays")
                di.free(); do.free()
                eng.reset()
        # CSR carrier lists (repeats = multiplicities when mult)
        eng.accumulate_callsets([list(np.repeat(np.arange(n), r)) for r in xm])
        check(eng, want, "csr")
        if n >= 2 and idx % 4 == 0:
            comps, lam, nz = eng.compute(min(2, n))   # centring + Lanczos (dense solver below N = 32) under the guard
            assert np.all(np.isfinite(comps)) and np.all(np.isfinite(lam))
    if idx % 6 == 1 and n >= 64 and kernel != "f32" and not mult:
        # a strip owner (rectangular tile list, strip reductions) and the dense eigensolver
        c0, cols = n // 3, n - n // 3 - 1
        with P.PcoaEngine(n, gram_kernel=kernel, operand=operand, strip=(c0, cols)) as st:
            st.accumulate_dense_u8(x)
            check(st, want[:, c0:c0 + cols], "strip")
            st.strip_col_sums()
            st.strip_matvec(np.ones(n), np.zeros(n), 0.0)
        with P.PcoaEngine(n, eig="householder") as hh:
            hh.load_gram(want)
            hh.compute(2)


def ring_case(seed, idx):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1025, 1279, 1280, 1281, 1536, 2504, 2561, 3000, 4100])) if rng.random() < 0.7 else int(rng.integers(1025, 2600))
    ld = (n + 3) // 4 * 4 + 4 * int(rng.integers(0, 3))       # ring_ok: stride a multiple of 4 floats
    calls = [int(rng.choice([127, 129, 1000, 2049, 4097, 5000])) for _ in range(int(rng.integers(3, 7)))]
    dens = float(rng.choice([0.02, 0.2, 0.6]))
    print("ring case seed=%d n=%d ld=%d calls=%s" % (seed, n, ld, calls), flush=True)
    want = np.zeros((n, n), dtype=np.int64)
    with P.PcoaEngine(n, gram_kernel=["auto", "fp4"][idx % 2]) as eng:
        ctx = eng._ctx
        bufs = []
        auto = idx % 2 == 0                                   # gram_kernel "auto": a multiplicity is legal (int8 redo)
        for j, v in enumerate(calls):
            x = (rng.random((v, n)) < dens).astype(np.uint8)
            fmt = (idx + j) % 3 if idx % 4 else 0              # every fourth case fp32 only, else fp32 / uint8 / bitsets mixed
            if auto and fmt != 2 and idx % 4 == 2 and j % 2 == 1:
                # carrier multiplicities inside a tile that fills on the pre-pass stream: the generation's contraction is
                # skipped by the device-side predicate and its chunks are redone on the int8 kernel
                x = x.astype(np.int64)
                x[int(rng.integers(0, v)), int(rng.integers(0, n))] = int(rng.integers(2, 6))
                x[v - 1, n - 1] = 3
            want += int_gram(x)
            if fmt == 0:
                a = np.full((v, ld), np.nan, dtype=np.float32)    # garbage in the padding columns
                a[:, :n] = x
                d = DevBuf(a)                                     # the allocation ends with the last row
                eng._check(lib.pcoa_accumulate_dense_f32(ctx, d.ptr, v, ld, 1))
            elif fmt == 1:
                ld8 = (n + 7) // 8 * 8 + 8 * int(rng.integers(0, 3))   # u8 ring: stride a multiple of 8 bytes (% 16 == 8 too)
                a8 = np.full((v, ld8), 255, dtype=np.uint8)
                a8[:, :n] = x
                d = DevBuf(a8)
                eng._check(lib.pcoa_accumulate_dense_u8(ctx, d.ptr, v, ld8, 1))
            else:
                pw = int(rng.integers(0, 3))
                bits = ingest.pack_bits(x, pad_words=pw)
                if pw:
                    bits[:, (n + 31) // 32:] = 0xa5a5a5a5
                d = DevBuf(bits)
                eng._check(lib.pcoa_accumulate_bits(ctx, d.ptr, v, bits.shape[1], 1))
            bufs.append(d)
        check(eng, want, "device tiles (fp32 / uint8 / bitsets) through the pipeline")
        tim = eng.timings()
        for d in bufs:
            d.free()
        return int(tim["pipeline_launches"])


def aln_case(seed, idx):
    """fp32 SUB-tiles through the co-resident pipeline's ring pre-pass: every residue of the pitch mod 32 floats (rows on and off
    the 128-byte lines), sub-tiles that start at row 0..3 of an allocation (every offset of the first row inside its line),
    sample counts that fill the padded range to within a few samples, NaN in the pitch padding and in the rows before the
    sub-tile."""
    rng = np.random.default_rng(seed)
    n = [2504, 2504, 2532, 2536, 1284, 3000][idx % 6]
    ld = (n + 7) // 8 * 8 + 8 * int(rng.integers(0, 5))
    r0 = int(rng.integers(0, 4))
    calls = [int(rng.choice([2049, 4097, 4500, 6000])) for _ in range(int(rng.integers(3, 6)))]
    dens = float(rng.choice([0.05, 0.3]))
    print("aln case seed=%d n=%d ld=%d (mod 32: %d) first row %d calls=%s" % (seed, n, ld, ld % 32, r0, calls), flush=True)
    want = np.zeros((n, n), dtype=np.int64)
    with P.PcoaEngine(n, gram_kernel="fp4") as eng:
        ctx = eng._ctx
        bufs = []
        for v in calls:
            x = (rng.random((v, n)) < dens).astype(np.uint8)
            x[0, 0] = x[0, n - 1] = x[v - 1, 0] = x[v - 1, n - 1] = 1      # the corners of the sub-tile
            want += int_gram(x)
            a = np.full((v + r0, ld), np.nan, dtype=np.float32)
            a[r0:, :n] = x
            d = DevBuf(a)
            eng._check(lib.pcoa_accumulate_dense_f32(ctx, ctypes.c_void_p(d.ptr.value + 4 * r0 * ld), v, ld, 1))
            bufs.append(d)
        check(eng, want, "fp32 sub-tiles with rows off the 128-byte lines")
        tim = eng.timings()
        for d in bufs:
            d.free()
        return int(tim["pipeline_launches"])


def main():
    if len(sys.argv) > 5 and sys.argv[5] == "aln":
        n_cases, first = int(sys.argv[1]), int(sys.argv[2])
        mode = lib.pcoa_debug_guard_mode()
        print("guard mode %d (sub-tiles, pitches)" % mode, flush=True)
        piped = sum(aln_case(first + i, i) for i in range(n_cases))
        assert hip.hipDeviceSynchronize() == 0
        assert piped > 0, "no contraction was launched beside a pre-pass: is PCOA_DEBUG_MAX_LAUNCH set?"
        print("guard sweep ok: %d sub-tile / pitch cases, %d pipelined contraction launches, mode %d" % (n_cases, piped, mode), flush=True)
        return
    if len(sys.argv) > 5 and sys.argv[5] == "ring":
        n_cases, first = int(sys.argv[1]), int(sys.argv[2])
        mode = lib.pcoa_debug_guard_mode()
        print("guard mode %d (ring)" % mode, flush=True)
        piped = sum(ring_case(first + i, i) for i in range(n_cases))
        assert hip.hipDeviceSynchronize() == 0
        # the sweep must actually have reached the pipeline (else it proves nothing about the ring kernel)
        assert piped > 0, "no contraction was launched beside a pre-pass: is PCOA_DEBUG_MAX_LAUNCH set?"
        print("guard sweep ok: %d ring cases, %d pipelined contraction launches, mode %d" % (n_cases, piped, mode), flush=True)
        return
    n_cases, first = int(sys.argv[1]), int(sys.argv[2])
    loops = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    idx0 = int(sys.argv[4]) if len(sys.argv) > 4 else 0   # reproduce case number idx0 of an earlier sweep on its own
    mode = lib.pcoa_debug_guard_mode()
    print("guard mode %d" % mode, flush=True)
    assert mode == int(os.environ.get("PCOA_DEBUG_GUARD", "0"))
    for lp in range(loops):
        for i in range(n_cases):
            one_case(first + i, idx0 + i)
    assert hip.hipDeviceSynchronize() == 0
    print("guard sweep ok: %d cases x %d loops, mode %d" % (n_cases, loops, mode), flush=True)


if __name__ == "__main__":
    main()
