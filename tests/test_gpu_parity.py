"""GPU parity tests: the HIP path (through the C ABI) against the oracle on the same inputs.

Bars (BASELINE.json north_star): Gram bit-exact (integers); centred matrix bit-exact (fp64, same
operation order); sign-normalised eigenpairs within 1e-6 relative.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, align_sign, golden_cases, int_gram, load_golden, load_oracle, load_pkg, planted_callsets

pytestmark = pytest.mark.gpu

EIG_TOL = 1e-6  # north_star: 1e-6 relative on sign-normalised eigenpairs


@pytest.fixture(scope="module")
def P():
    return load_pkg()


@pytest.fixture(scope="module")
def O():
    return load_oracle()


def callsets_of(x):
    return [list(np.nonzero(r)[0]) for r in x]


# every Gram kernel must give the reference's integers: auto = MX-FP4 MFMA for binary tiles with a
# per-chunk int8 fallback, fp4 / i8 / f32 force one kernel
KERNELS = ["auto", "fp4", "i8", "f32"]


# ------------------------------------------------------------------------------------------ Gram
@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("name", golden_cases())
def test_gram_and_centering_match_reference_python_goldens(P, name, kernel):
    g = load_golden(name)
    n = int(g["n_samples"])
    with P.PcoaEngine(n, gram_kernel=kernel) as eng:
        eng.accumulate_calls(g["sample_idx"], g["row_offsets"])
        eng.finalize()
        assert np.array_equal(eng.gram(), g["similarity"])
        b, rs, nz, mm = eng.center()
        assert np.array_equal(b, g["centered"])  # bit for bit
        assert np.array_equal(rs, g["similarity"].sum(axis=1).astype(np.float64))
        assert nz == int((g["similarity"].sum(axis=1) > 0).sum())


def test_known_answer_survey_8c(P):
    callsets = [[0, 1], [0, 1, 2], [3, 4], [2, 3, 4], [0], [1, 4]]
    with P.PcoaEngine(5) as eng:
        eng.accumulate_callsets(callsets)
        s = eng.gram()
        assert s.tolist() == [[3, 2, 1, 0, 0], [2, 3, 1, 0, 1], [1, 1, 2, 1, 1], [0, 0, 1, 2, 2], [0, 1, 1, 2, 3]]
        comps, lam, nz = eng.compute(2)
    assert nz == 5
    assert np.allclose(lam, [4.291279249547, 1.505156222449], rtol=0, atol=1e-11)
    pc1 = [0.578932130782, 0.427318800977, -0.024951663211, -0.482297462206, -0.499001806342]
    pc2 = [-0.290911908486, 0.610085542084, -0.521742006736, -0.252576095428, 0.455144468566]
    assert np.allclose(comps[:, 0], pc1, atol=1e-11) and np.allclose(comps[:, 1], pc2, atol=1e-11)


@pytest.mark.parametrize("n,v", [(1, 1), (2, 3), (3, 17), (5, 16), (63, 100), (64, 15), (65, 33), (127, 1),
                                 (128, 64), (129, 257), (200, 1000), (257, 129), (384, 2048), (777, 300),
                                 (255, 63), (256, 64), (513, 65), (1030, 200)])
@pytest.mark.parametrize("kernel", KERNELS)
def test_gram_dense_csr_oracle_agree_on_ragged_shapes(P, O, n, v, kernel):
    rng = np.random.default_rng(1000 * n + v)
    x = (rng.random((v, n)) < rng.uniform(0.05, 0.5)).astype(np.float32)
    want = O.similarity_from_dense(x, n)
    with P.PcoaEngine(n, gram_kernel=kernel) as eng:
        eng.accumulate_dense(x)               # host tile; ld = n (vec4 path iff n % 4 == 0 after staging)
        got_dense = eng.gram()
        eng.reset()
        eng.accumulate_callsets(callsets_of(x))
        got_csr = eng.gram()
    assert np.array_equal(got_dense, want)
    assert np.array_equal(got_csr, want)


@pytest.mark.parametrize("kernel", KERNELS)
def test_gram_device_pointer_paths_and_padding_is_ignored(P, O, kernel):
    import torch
    rng = np.random.default_rng(3)
    n, v = 300, 500
    x = (rng.random((v, n)) < 0.3).astype(np.float32)
    want = O.similarity_from_dense(x, n)
    for ld in (300, 301, 303, 304, 320):  # ld % 4 != 0 -> 4-byte DMA kernel; padding holds NaN
        buf = torch.full((v, ld), float("nan"), dtype=torch.float32, device="cuda")
        buf[:, :n] = torch.from_numpy(x).cuda()
        with P.PcoaEngine(n, gram_kernel=kernel) as eng:
            eng.accumulate_dense(buf)
            assert np.array_equal(eng.gram(), want), "ld=%d" % ld
    # misaligned base pointer (4-byte aligned only)
    flat = torch.zeros(v * 304 + 1, dtype=torch.float32, device="cuda")
    view = flat[1:].view(v, 304)
    view[:, :n] = torch.from_numpy(x).cuda()
    with P.PcoaEngine(n, gram_kernel=kernel) as eng:
        eng.accumulate_dense(view)
        assert np.array_equal(eng.gram(), want)


@pytest.mark.parametrize("kernel", KERNELS)
def test_repeated_indices_count_with_multiplicity_like_the_reference_double_loop(P, O, kernel):
    callsets = [[0, 0, 1], [2], [1, 2, 2, 2]]
    want = O.similarity_matrix_python_loops(callsets, 4)
    assert want[0, 0] == 4 and want[2, 2] == 10
    with P.PcoaEngine(4, gram_kernel=kernel) as eng:
        if kernel == "fp4":   # a repeated callset is a multiplicity: the forced FP4 engine refuses it
            with pytest.raises(P.PcoaError):
                eng.accumulate_callsets(callsets)
            return
        eng.accumulate_callsets(callsets)
        assert np.array_equal(eng.gram(), want)
        assert eng.timings()["gram_kernel_kind"] == (1 if kernel == "f32" else 2)   # never the FP4 kernel
        eng.reset()
        eng.accumulate_callsets([sorted(set(c)) for c in callsets])                 # the same lists as SETS
        assert np.array_equal(eng.gram(), O.similarity_matrix_python_loops([sorted(set(c)) for c in callsets], 4))
        assert eng.timings()["gram_kernel_kind"] == {"f32": 1, "i8": 2}.get(kernel, 3)


@pytest.mark.parametrize("n,v", [(5, 3), (64, 100), (300, 777), (1030, 200)])
def test_u8_boundary_matches_oracle_host_and_device(P, O, n, v):
    import torch
    rng = np.random.default_rng(n + 7 * v)
    x = (rng.random((v, n)) < 0.25).astype(np.uint8)
    x[0, 0] = 3  # multiplicities up to 127 are legal
    want = (x.T.astype(np.int64) @ x.astype(np.int64))
    with P.PcoaEngine(n, gram_kernel="fp4") as eng:   # binary uint8 tile on the FP4 kernel
        xb = np.minimum(x, 1)
        eng.accumulate_dense_u8(xb)
        assert np.array_equal(eng.gram(), xb.T.astype(np.int64) @ xb.astype(np.int64))
        assert eng.timings()["gram_kernel_kind"] == 3
    with P.PcoaEngine(n) as eng:
        eng.accumulate_dense_u8(x)
        assert np.array_equal(eng.gram(), want)
        t = eng.timings()
        assert t["fp4_fallbacks"] == 1 and t["gram_kernel_kind"] == 2   # the 3 sent this chunk to the int8 kernel
        eng.reset()
        for ld in (n, n + 1, ((n + 15) // 16) * 16 + 16):  # unaligned and padded strides, padding = 0xff
            buf = torch.full((v, ld), 255, dtype=torch.uint8, device="cuda")
            buf[:, :n] = torch.from_numpy(x).cuda()
            eng.accumulate_dense_u8(buf)
        assert np.array_equal(eng.gram(), 3 * want)
    with P.PcoaEngine(n) as eng:
        bad = x.copy()
        bad[v // 2, n // 2] = 128
        eng.accumulate_dense_u8(bad)
        with pytest.raises(P.PcoaError):
            eng.gram()


@pytest.mark.parametrize("n,v", [(1, 1), (5, 3), (31, 40), (32, 33), (33, 100), (64, 64), (65, 31), (300, 777),
                                 (1030, 200), (2504, 3000)])
def test_bit_packed_boundary_matches_oracle_host_and_device(P, O, n, v):
    """pcoa_accumulate_bits: carrier bitsets, host and device pointers, padded / odd row strides whose padding
    bits are garbage, accumulation on top of the other boundaries."""
    import torch
    ingest = load_pkg("ingest")
    rng = np.random.default_rng(5 * n + v)
    x = (rng.random((v, n)) < 0.3).astype(np.uint8)
    want = int_gram(x)
    bits = ingest.pack_bits(x)
    assert bits.shape == (v, (n + 31) // 32)
    with P.PcoaEngine(n) as eng:
        eng.accumulate_bits(bits)                                    # host, tight stride
        assert np.array_equal(eng.gram(), want)
        assert eng.timings()["gram_kernel_kind"] == 3
        for pad in (1, 2, 5):                                        # device, padded stride, garbage in the padding
            wide = ingest.pack_bits(x, pad_words=pad)
            wide[:, bits.shape[1]:] = 0xdeadbeef
            if n % 32:
                wide[:, bits.shape[1] - 1] |= np.uint32((0xffffffff << (n % 32)) & 0xffffffff)  # bits of samples >= N
            eng.accumulate_bits(torch.from_numpy(wide.view(np.int32)).cuda())
        eng.accumulate_bits(wide)                                    # host, padded stride
        eng.accumulate_dense(x.astype(np.float32))                   # mixes with the dense boundary
        assert np.array_equal(eng.gram(), 6 * want)
    with P.PcoaEngine(n, gram_kernel="f32") as eng:
        with pytest.raises(P.PcoaError):
            eng.accumulate_bits(bits)
    with P.PcoaEngine(n) as eng:
        with pytest.raises(P.PcoaError):
            eng.accumulate_bits(bits[:, :-1] if bits.shape[1] > 1 else np.zeros((v, 0), dtype=np.uint32))


def test_i8_path_rejects_non_integer_or_large_values_and_f32_path_accepts_integers(P, O):
    x = np.zeros((40, 12), dtype=np.float32)
    x[3, 4] = 1.0
    for bad in (0.5, -1.0, 128.0, float("nan")):
        xb = x.copy()
        xb[7, 2] = bad
        with P.PcoaEngine(12) as eng:
            eng.accumulate_dense(xb)
            with pytest.raises(P.PcoaError) as ei:
                eng.gram()
            assert "integer in [0, 127]" in str(ei.value)
    xb = x.copy()
    xb[7, 2] = 127.0
    xb[9, 2] = 3.0
    want = (xb.T.astype(np.int64) @ xb.astype(np.int64))
    for kernel in ("auto", "i8", "f32"):
        with P.PcoaEngine(12, gram_kernel=kernel) as eng:
            eng.accumulate_dense(xb)
            assert np.array_equal(eng.gram(), want)
    with P.PcoaEngine(12, gram_kernel="fp4") as eng:   # forced FP4 refuses anything but 0 / 1
        eng.accumulate_dense(xb)
        with pytest.raises(P.PcoaError) as ei:
            eng.gram()
        assert "other than 0 or 1" in str(ei.value)
    xb[7, 2] = 300.0  # beyond int8: only the fp32-MFMA kernel takes it
    want = (xb.T.astype(np.int64) @ xb.astype(np.int64))
    with P.PcoaEngine(12, gram_kernel="f32") as eng:
        eng.accumulate_dense(xb)
        assert np.array_equal(eng.gram(), want)


def test_large_multiplicities_never_wrap_or_round_silently(P):
    """the exactness limits used to count variants, not summed counts.  With multiplicity m an entry grows
    by m^2 per variant: the int8 path now sizes its launches and int64 folds by the largest m its pre-pass met, and the
    fp32-MFMA kernel reports an accumulator that left the exact range instead of rounding."""
    import torch
    n, v = 40, 300000
    x = torch.zeros((v, n), dtype=torch.float32, device="cuda")
    x[:, 0] = 100.0                                   # S[0, 0] = 3e9 > 2^31: must come back exact (int64)
    x[:, 1] = 1.0
    x[::3, 2] = 7.0
    for kernel in ("i8", "auto"):
        with P.PcoaEngine(n, gram_kernel=kernel) as eng:
            eng.accumulate_dense(x)
            s = eng.gram()
            assert s[0, 0] == 100 * 100 * v and s[0, 1] == 100 * v and s[1, 1] == v
            assert s[2, 2] == 49 * ((v + 2) // 3) and s[0, 2] == 700 * ((v + 2) // 3)
            assert np.array_equal(s, s.T)
    with P.PcoaEngine(n, gram_kernel="f32") as eng:   # 3e9 does not fit an exact fp32 accumulator: an error, not a wrong S
        eng.accumulate_dense(x)
        with pytest.raises(P.PcoaError) as ei:
            eng.gram()
        assert "exact range" in str(ei.value)
    with P.PcoaEngine(n, gram_kernel="f32") as eng:   # ... while small products stay exact on it
        eng.accumulate_dense(x[:1000])
        assert eng.gram()[0, 0] == 100 * 100 * 1000
    # repeated callsets in carrier lists (CSR boundary): multiplicity 3 over many rows
    rows = 200000
    idx = np.tile(np.array([5, 5, 5, 9], dtype=np.int32), rows)
    offs = (np.arange(rows + 1, dtype=np.int64) * 4)
    with P.PcoaEngine(n) as eng:
        eng.accumulate_calls(idx, offs)
        s = eng.gram()
        assert s[5, 5] == 9 * rows and s[5, 9] == 3 * rows and s[9, 9] == rows


def test_auto_mode_picks_fp4_per_chunk_and_falls_back_to_int8_on_multiplicities(P, O):
    """auto: binary chunks run on the MX-FP4 MFMA, a chunk holding a multiplicity is re-packed as int8;
    the sum over chunks is exact either way (chunk size forced small through the debug hook)."""
    rng = np.random.default_rng(77)
    n, v = 333, 1000
    x = (rng.random((v, n)) < 0.3).astype(np.float32)
    x[700, 5] = 2.0          # only the chunk holding variant 700 must fall back
    x[701, 9] = 100.0
    want = (x.T.astype(np.int64) @ x.astype(np.int64))
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r);"
            "from conftest import load_pkg; P = load_pkg(); x = np.load(sys.argv[1]);"
            "e = P.PcoaEngine(x.shape[1]); e.accumulate_dense(x); s = e.gram(); t = e.timings();"
            "np.save(sys.argv[2], s); print(json.dumps([t['fp4_fallbacks'], t['pack_launches']]))"
            ) % (ROOT, os.path.join(ROOT, "tests"))
    import json
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "x.npy"), x)
        env = dict(os.environ, PCOA_DEBUG_PACK_CHUNK="256")
        out = subprocess.check_output([sys.executable, "-c", code, os.path.join(td, "x.npy"),
                                       os.path.join(td, "s.npy")], env=env)
        fallbacks, packs = json.loads(out.decode().strip().splitlines()[-1])
        assert np.array_equal(np.load(os.path.join(td, "s.npy")), want)
        assert fallbacks == 1 and packs == 4 + 1   # 4 chunks, one of them packed twice
    with P.PcoaEngine(n) as eng:                   # binary input: no fallback, FP4 kernel
        xb = np.minimum(x, 1.0)
        eng.accumulate_dense(xb)
        assert np.array_equal(eng.gram(), O.similarity_from_dense(xb, n))
        t = eng.timings()
        assert t["fp4_fallbacks"] == 0 and t["gram_kernel_kind"] == 3


def test_deferred_fp4_contraction_flushes_where_s_is_needed(P, O):
    """Binary chunks are only packed when they arrive; the contraction runs when the operand buffer is full or S is
    needed.  Many small calls -> one launch; reset / load_gram drop what was buffered for the old S; int8 chunks in
    between run at once; every reader of S sees the buffered calls."""
    rng = np.random.default_rng(91)
    n = 130
    xs = [(rng.random((v, n)) < 0.3).astype(np.float32) for v in (1, 31, 32, 33, 100, 7, 64)]
    want = sum(O.similarity_from_dense(x, n) for x in xs)
    with P.PcoaEngine(n) as eng:
        eng.reset_timings()
        for x in xs:
            eng.accumulate_dense(x)
        t = eng.timings()
        assert t["gram_kernel_launches"] == 0 and t["pack_launches"] == len(xs)     # nothing contracted yet
        assert np.array_equal(eng.gram(), want)                                      # the read flushes
        assert eng.timings()["gram_kernel_launches"] == 1
        assert np.array_equal(eng.gram_block(3, 5, 20, 30), want[3:23, 5:35])
        # buffered operands belong to the S they were accumulated for
        eng.accumulate_dense(xs[4])
        eng.reset()
        assert not eng.gram().any()
        eng.accumulate_dense(xs[4])
        eng.load_gram(want)
        eng.accumulate_dense(xs[0])
        assert np.array_equal(eng.gram(), want + O.similarity_from_dense(xs[0], n))
        # an int8 chunk between FP4 chunks, bitsets and uint8 on top, then the PCA consumes all of it
        eng.reset()
        xm = xs[3].copy()
        xm[2, 7] = 5.0
        eng.accumulate_dense(xs[4])
        eng.accumulate_dense(xm)
        eng.accumulate_dense_u8(xs[5].astype(np.uint8))
        eng.accumulate_bits(load_pkg("ingest").pack_bits(xs[6]))
        s_all = (O.similarity_from_dense(xs[4], n) + (xm.T.astype(np.int64) @ xm.astype(np.int64))
                 + O.similarity_from_dense(xs[5], n) + O.similarity_from_dense(xs[6], n))
        comps, lam, nz = eng.compute(2)                                              # compute flushes too
        ref = O.compute_pca(s_all, 2)
        assert np.abs(align_sign(comps, ref["components"]) - ref["components"]).max() < EIG_TOL
        assert np.array_equal(eng.gram(), s_all)


def test_empty_and_ragged_inputs(P):
    with P.PcoaEngine(7) as eng:
        eng.accumulate_calls(np.zeros(0, dtype=np.int32), np.zeros(1, dtype=np.int64))       # no variants
        eng.accumulate_calls(np.zeros(0, dtype=np.int32), np.zeros(4, dtype=np.int64))       # 3 empty rows
        eng.accumulate_dense(np.zeros((0, 7), dtype=np.float32))
        assert not eng.gram().any()
        eng.accumulate_callsets([[], [6], [], [0, 6]])
        s = eng.gram()
        assert s[6, 6] == 2 and s[0, 6] == 1 and s[6, 0] == 1 and s[0, 0] == 1 and s.sum() == 5


def test_gram_block_readback(P, O):
    rng = np.random.default_rng(8)
    x = (rng.random((300, 70)) < 0.3).astype(np.float32)
    want = O.similarity_from_dense(x, 70)
    with P.PcoaEngine(70) as eng:
        eng.accumulate_dense(x)
        assert np.array_equal(eng.gram_block(0, 0, 70, 70), want)
        assert np.array_equal(eng.gram_block(5, 33, 17, 20), want[5:22, 33:53])
        assert np.array_equal(eng.gram_block(69, 0, 1, 70), want[69:70])
        with pytest.raises(P.PcoaError):
            eng.gram_block(60, 60, 11, 5)


def test_index_out_of_range_is_rejected_and_leaves_s_unchanged(P):
    with P.PcoaEngine(5) as eng:
        eng.accumulate_callsets([[0, 1]])
        before = eng.gram()
        for bad in ([[0, 5]], [[-1]], [[1], [2, 7]]):
            with pytest.raises(P.IndexRangeError):
                eng.accumulate_callsets(bad)
        assert np.array_equal(eng.gram(), before)
        with pytest.raises(P.PcoaError):
            eng.compute(0)
        with pytest.raises(P.PcoaError):
            eng.compute(6)  # MLlib: require(k > 0 && k <= n)


def _csr_of(x):
    """(sample_idx int32, row_offsets int64) of a dense 0/1 matrix [V][N]."""
    rows, cols = np.nonzero(x)
    offs = np.zeros(x.shape[0] + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=x.shape[0]), out=offs[1:])
    return cols.astype(np.int32), offs


@pytest.mark.parametrize("where", ["pageable", "pinned", "pinned_async", "device"])
def test_carrier_lists_from_pageable_pinned_and_device_memory_give_the_oracle_matrix(P, O, where):
    """pcoa_accumulate_calls_ex (r04): the device validates, the lists travel in chunks through two staging slots; the result is
    the oracle's S whatever memory the arrays live in, across several calls, with an operand-buffer switch in between."""
    import torch
    rng = np.random.default_rng(77)
    n, v = 300, 6000
    x = (rng.random((v, n)) < 0.11).astype(np.float32)
    x[17] = 0                      # an empty row (legal: the reference filters it, VariantsPca.scala:166)
    want = O.similarity_from_dense(x, n)
    with P.PcoaEngine(n) as eng:
        for lo, hi in ((0, 1), (1, 2500), (2500, 2501), (2501, 6000)):
            idx, offs = _csr_of(x[lo:hi])
            ti, to = torch.from_numpy(idx), torch.from_numpy(offs)
            if where == "device":
                ti, to = ti.cuda(), to.cuda()
            elif where != "pageable":
                ti, to = ti.pin_memory(), to.pin_memory()
            eng.accumulate_calls_tensors(ti, to, asynchronous=(where == "pinned_async"))
        assert np.array_equal(eng.gram(), want)
        tim = eng.timings()
        assert tim["gram_kernel_kind"] == 3 and tim["csr_fast_chunks"] >= 4 and tim["csr_redo_chunks"] == 0


def test_pageable_carrier_lists_whose_size_is_not_a_multiple_of_the_copy_split(P):
    """Regression (r04): the threaded copy into pinned staging cut a chunk into floor(bytes / threads) pieces and lost the chunk's
    last entry whenever that floor was already a multiple of 64.  Chunks of 8 * 2^20 * 4 + 4 bytes and neighbours, pageable
    against device arrays (which are not staged)."""
    import torch
    n = 64
    for nnz in (8 * (1 << 20) + 1, 8 * (1 << 20) + 17, 3 * (1 << 20) + 5):
        v = nnz // 32 + 1
        g = torch.Generator().manual_seed(nnz)
        # rows of 32 distinct callsets each (the last one shorter): a random 32-subset of 64 per row
        perm = torch.argsort(torch.rand((v, n), generator=g), dim=1)[:, :32].to(torch.int32)
        perm, _ = torch.sort(perm, dim=1)
        idx = perm.reshape(-1)[:nnz].contiguous()
        offs = torch.clamp(torch.arange(v + 1, dtype=torch.int64) * 32, max=nnz)
        with P.PcoaEngine(n) as a, P.PcoaEngine(n) as b:
            a.accumulate_calls_tensors(idx, offs)
            b.accumulate_calls_tensors(idx.cuda(), offs.cuda())
            assert np.array_equal(a.gram(), b.gram())


@pytest.mark.parametrize("where", ["pageable", "pinned_async", "device"])
def test_the_device_side_check_of_carrier_lists_rejects_rolls_back_and_redoes(P, O, where):
    """(a) an index outside [0, N): PCOA_ERR_INDEX_RANGE -- from the call itself when it is synchronous, from the next
    synchronising call when the arrays were handed over asynchronously -- and S unchanged by the rejected lists;
    (b) a list that names a callset twice: found by the scatter's atomic OR, the chunk redone on the int8 kernel with the
    reference's multiplicities (VariantsPca.scala:187); the engine keeps working afterwards."""
    import torch

    def feed(eng, callsets):
        offs = np.zeros(len(callsets) + 1, dtype=np.int64)
        for i, c in enumerate(callsets):
            offs[i + 1] = offs[i] + len(c)
        idx = np.fromiter((j for c in callsets for j in c), dtype=np.int32, count=int(offs[-1]))
        ti, to = torch.from_numpy(idx), torch.from_numpy(offs)
        if where == "device":
            ti, to = ti.cuda(), to.cuda()
        elif where == "pinned_async":
            ti, to = ti.pin_memory(), to.pin_memory()
        eng.accumulate_calls_tensors(ti, to, asynchronous=(where == "pinned_async"))

    n = 40
    good = [[0, 1, 39], [2], [5, 6, 7, 8], [], [39]]
    with P.PcoaEngine(n) as eng:
        feed(eng, good)
        before = eng.gram()
        assert np.array_equal(before, O.similarity_matrix_python_loops(good, n))
        for bad in ([[0, 40]], [[-1]], [[1], [2, 77], [3]]):
            with pytest.raises(P.IndexRangeError):
                feed(eng, bad)
                eng.sync()          # asynchronous / device arrays: the check is reported here at the latest
            assert np.array_equal(eng.gram(), before)
        # repeats: multiplicity semantics of the reference's double loop
        rep = [[0, 0, 1], [2], [1, 2, 2, 2], [3, 4]]
        feed(eng, rep)
        want = before + O.similarity_matrix_python_loops(rep, n)
        assert np.array_equal(eng.gram(), want)
        assert eng.timings()["csr_redo_chunks"] >= 1
        feed(eng, good)
        assert np.array_equal(eng.gram(), want + before)
    with P.PcoaEngine(n, gram_kernel="fp4") as eng:   # the forced FP4 engine refuses a multiplicity
        with pytest.raises(P.PcoaError):
            feed(eng, [[0, 0, 1]])
            eng.sync()


@pytest.mark.parametrize("n", [5, 33, 300, 2504, 5000, 9000])
def test_carrier_lists_through_the_lds_scatter_on_ragged_blocks_and_misaligned_arrays(P, O, n):
    """r05: densify_csr_kbits_lds_kernel -- one workgroup per block of 128 variants streams the block's contiguous run of
    entries with 16-byte loads and scatters bits in LDS.  What it has to get right: runs that start at any of the four
    alignments (device arrays handed over as views, blocks that start anywhere in idx[]), empty rows at the ends and in the
    middle of a block, full rows (all N callsets), row counts around the block size, N with more than 64 KiB of LDS
    (5,000) and N where the r04 global-atomic form still runs (9,000)."""
    import torch
    rng = np.random.default_rng(n)
    for v in (1, 127, 128, 129, 257, 700):
        dens = rng.choice([0.002, 0.05, 0.4], size=v)
        x = (rng.random((v, n)) < dens[:, None]).astype(np.float32)
        x[0] = 0
        x[v - 1] = 0
        if v > 130:
            x[127] = 0
            x[128] = 1          # every callset
            x[64:70] = 0
        want = O.similarity_from_dense_blas(x)
        idx, offs = _csr_of(x)
        for shift in (0, 1, 2, 3):
            big = torch.zeros(idx.size + 8, dtype=torch.int32, device="cuda")
            big[shift:shift + idx.size] = torch.from_numpy(idx).cuda()
            ti = big[shift:shift + idx.size]
            to = torch.from_numpy(offs).cuda()
            with P.PcoaEngine(n) as eng:
                eng.accumulate_calls_tensors(ti, to)
                got = eng.gram()
            assert np.array_equal(got, want), (n, v, shift)
        with P.PcoaEngine(n) as eng:   # host arrays, two calls: the second starts a new block of the operand
            h = v // 2
            for lo, hi in ((0, h), (h, v)):
                if hi > lo:
                    i2, o2 = _csr_of(x[lo:hi])
                    eng.accumulate_calls_tensors(torch.from_numpy(i2), torch.from_numpy(o2))
            assert np.array_equal(eng.gram(), want), (n, v, "host")


@pytest.mark.parametrize("n", [1024, 2504, 3076, 4100])
def test_upper_triangle_form_of_the_centred_matvec_agrees_with_the_row_form_and_the_oracle(P, O, n):
    """r04: from N = 16,384 the Lanczos mat-vec reads only the upper-triangular 1024 x 1024 tiles of S (each entry serves y_i and
    y_j).  Same entries of B (reference operation order), another order of the additions: the two forms agree to 1e-13 of
    ||B|| ||x||, and with the oracle's materialised B.  Shapes: one tile, a ragged last tile in both directions, N % 1024 small."""
    rng = np.random.default_rng(n)
    v = 700
    x8 = (rng.random((v, n)) < rng.uniform(0.02, 0.4, size=(v, 1))).astype(np.uint8)
    with P.PcoaEngine(n) as eng:
        eng.accumulate_dense_u8(x8)
        s = eng.gram()
        b = O.center_matrix(s)[0]
        scale = np.abs(b).sum(axis=1).max()
        for seed in range(3):
            xv = np.random.default_rng(seed).standard_normal(n)
            y_row = eng.debug_centred_matvec(xv, 0)
            y_tri = eng.debug_centred_matvec(xv, 1)
            want = b @ xv
            assert np.abs(y_row - want).max() <= 1e-12 * scale * np.abs(xv).max()
            assert np.abs(y_tri - y_row).max() <= 1e-13 * scale * np.abs(xv).max()
        # a unit vector picks out one row / column of B exactly (the entries themselves are bit-identical in both forms)
        for k in (0, n // 2 + 1, n - 1):
            e = np.zeros(n)
            e[k] = 1.0
            assert np.array_equal(eng.debug_centred_matvec(e, 1), b[:, k]) and np.array_equal(eng.debug_centred_matvec(e, 0), b[:, k])


@pytest.mark.parametrize("n", [1024, 2504, 4100])
def test_compute_pca_through_the_large_n_forms_matches_the_oracle(P, O, n):
    """computePca as it runs from N = 16,384 -- exact row sums and the Lanczos mat-vec both from the upper-triangular tiles of S
    (rowsums_sym_tiles_kernel, symv_sym_tiles_kernel; r05: all tile kinds in one launch) -- forced at small N in a fresh process
    (PCOA_SYMV_SYM_MIN_N): eigenpairs within 1e-6 of the oracle's computePca, nonZeroRows equal (two callsets carry nothing),
    and the same eigenpairs as the default forms to 1e-12."""
    import tempfile
    rng = np.random.default_rng(n + 5)
    v = 900
    pops = rng.integers(0, 3, size=n)
    x8 = np.zeros((v, n), dtype=np.uint8)
    for r in range(v):
        k = rng.integers(0, 3)
        x8[r] = rng.random(n) < np.where(pops == k, 0.45, 0.06)
    x8[:, 7] = 0
    x8[:, n - 1] = 0
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r);"
            "from conftest import load_pkg; P = load_pkg(); x = np.load(sys.argv[1]);"
            "e = P.PcoaEngine(x.shape[1]); e.accumulate_dense_u8(x); c, lam, nz = e.compute(2);"
            "np.savez(sys.argv[2], c=c, lam=lam, nz=nz, s=e.gram(), method=e.timings()['eig_method'])") % (ROOT, os.path.join(ROOT, "tests"))
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "x.npy"), x8)
        res = {}
        for tag, extra in (("large_n_forms", {"PCOA_SYMV_SYM_MIN_N": "1024"}), ("default", {})):
            subprocess.check_call([sys.executable, "-c", code, os.path.join(td, "x.npy"), os.path.join(td, tag + ".npz")],
                                  env=dict(os.environ, **extra))
            res[tag] = np.load(os.path.join(td, tag + ".npz"))
    got = res["large_n_forms"]
    ref = O.compute_pca(got["s"], 2)
    assert int(got["method"]) == 1, "the Lanczos path (the one with the large-N forms) must have run"
    assert int(got["nz"]) == ref["nonzero_rows"] == n - 2
    assert np.allclose(got["lam"], ref["eigenvalues"], rtol=1e-6)
    for tag, other in (("oracle", ref["components"]), ("default forms", res["default"]["c"])):
        for k in range(2):
            a, b = got["c"][:, k], other[:, k]
            if np.dot(a, b) < 0:
                a = -a
            assert np.linalg.norm(a - b) < (1e-6 if tag == "oracle" else 1e-12), (tag, k)


def test_accumulation_is_additive_shard_invariant_and_resumable(P, O):
    rng = np.random.default_rng(11)
    n, v = 150, 900
    x = (rng.random((v, n)) < 0.2).astype(np.float32)
    want = O.similarity_from_dense(x, n)
    with P.PcoaEngine(n) as eng:
        for lo, hi in ((0, 1), (1, 400), (400, 401), (401, 900)):
            eng.accumulate_dense(x[lo:hi])
        assert np.array_equal(eng.gram(), want)       # finalize, then keep accumulating
        eng.accumulate_dense(x[:100])
        want2 = want + O.similarity_from_dense(x[:100], n)
        assert np.array_equal(eng.gram(), want2)
        # checkpoint / resume: load S into a fresh engine and continue
        with P.PcoaEngine(n) as eng2:
            eng2.load_gram(want)
            eng2.accumulate_dense(x[:100])
            assert np.array_equal(eng2.gram(), want2)


def test_multi_launch_and_int64_fold_paths(P, O):
    rng = np.random.default_rng(12)
    n, v = 90, 700
    x = (rng.random((v, n)) < 0.4).astype(np.float32)
    want = O.similarity_from_dense(x, n)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r);"
            "from conftest import load_pkg; P = load_pkg(); x = np.load(sys.argv[1]);"
            "e = P.PcoaEngine(x.shape[1]); e.accumulate_dense(x); e.accumulate_dense(x[:50]);"
            "np.save(sys.argv[2], e.gram())") % (ROOT, os.path.join(ROOT, "tests"))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "x.npy"), x)
        for extra in ({"PCOA_GRAM_KERNEL": "f32"}, {"PCOA_GRAM_KERNEL": "i8", "PCOA_DEBUG_PACK_CHUNK": "48"},
                      {"PCOA_GRAM_KERNEL": "fp4", "PCOA_DEBUG_PACK_CHUNK": "48"},
                      {"PCOA_GRAM_KERNEL": "auto", "PCOA_DEBUG_PACK_CHUNK": "40"}):
            env = dict(os.environ, PCOA_DEBUG_MAX_LAUNCH="64", PCOA_DEBUG_FOLD_THRESHOLD="200", **extra)
            subprocess.check_call([sys.executable, "-c", code, os.path.join(td, "x.npy"),
                                   os.path.join(td, "s.npy")], env=env)
            got = np.load(os.path.join(td, "s.npy"))
            assert np.array_equal(got, want + O.similarity_from_dense(x[:50], n)), extra


def test_synthetic_device_generator_is_bit_identical_to_host_twin(P, O):
    import torch
    synth = load_pkg("synth")
    n, v, seed = 2504, 4096, 1002
    offs = synth.pop_offsets(n)
    thr = synth.thresholds(seed, 1000, v)
    x_host = synth.genotypes(seed, 1000, thr, offs)
    buf = torch.empty((v, n), dtype=torch.float32, device="cuda")
    with P.PcoaEngine(n) as eng:
        eng.synth_fill(seed, offs, thr, 1000, buf.data_ptr(), n)
        assert np.array_equal(buf.cpu().numpy(), x_host)
        eng.accumulate_synthetic(seed, offs, thr, 1000)
        s_synth = eng.gram()
        eng.reset()
        eng.accumulate_dense(buf)
        s_dense = eng.gram()
    want = O.similarity_from_dense_blas(x_host)
    assert np.array_equal(s_synth, want) and np.array_equal(s_dense, want)


def test_config2_shape_against_faithful_pair_loop(P, O):
    """N = 2504 (the BASELINE configs[1] sample count), 20k variants: HIP vs the faithful pair loop."""
    synth = load_pkg("synth")
    n, v, seed = 2504, 20000, 1002
    offs = synth.pop_offsets(n)
    thr = synth.thresholds(seed, 0, v)
    x = synth.genotypes(seed, 0, thr, offs)
    want = O.similarity_from_dense(x, n)
    for kernel in KERNELS:
        with P.PcoaEngine(n, gram_kernel=kernel) as eng:
            eng.accumulate_synthetic(seed, offs, thr, 0)
            got = eng.gram()
        assert np.array_equal(got, want), kernel


def test_full_config2_size_properties(P):
    """BASELINE configs[1] at full size (2,504 samples x 1M variants fp32, 10 GB resident):
    size-independent properties -- diagonal = column popcounts, symmetry, checksum
    sum(S) = sum_v k_v^2, and split invariance (two half calls == one call)."""
    import torch
    synth = load_pkg("synth")
    n, v, seed = 2504, 1000000, 1002
    offs = synth.pop_offsets(n)
    x = torch.empty((v, n), dtype=torch.float32, device="cuda")
    with P.PcoaEngine(n) as eng:
        step = 1 << 18
        for v0 in range(0, v, step):
            v1 = min(v, v0 + step)
            eng.synth_fill(seed, offs, synth.thresholds(seed, v0, v1 - v0), v0, x[v0:v1].data_ptr(), n)
        eng.accumulate_dense(x)
        s = eng.gram()
        col = x.sum(dim=0, dtype=torch.float64).cpu().numpy().astype(np.int64)
        kv = x.sum(dim=1, dtype=torch.float64)
        checksum = int((kv * kv).sum().item())
        assert np.array_equal(np.diag(s), col)
        assert np.array_equal(s, s.T)
        assert int(s.sum()) == checksum
        eng.reset()
        eng.accumulate_dense(x[:400001])
        eng.accumulate_dense(x[400001:])
        assert np.array_equal(eng.gram(), s)
    for kernel in ("i8", "f32"):   # the int8- and fp32-MFMA kernels give the same integers as the FP4 one
        with P.PcoaEngine(n, gram_kernel=kernel) as eng:
            eng.accumulate_dense(x)
            assert np.array_equal(eng.gram(), s), kernel


# ------------------------------------------------------------------------------------------ PCA
def check_eigenpairs(comps, lam, ref_comps, ref_lam, b):
    n = b.shape[0]
    assert np.allclose(lam, ref_lam, rtol=EIG_TOL, atol=0)
    c = align_sign(comps, ref_comps)
    for k in range(c.shape[1]):
        assert abs(np.linalg.norm(c[:, k]) - 1.0) < 1e-12
        assert np.linalg.norm(c[:, k] - ref_comps[:, k]) <= EIG_TOL, "PC%d" % (k + 1)
        resid = np.linalg.norm(b @ c[:, k] - lam[k] * c[:, k]) / abs(lam[k])
        assert resid < 1e-10
    assert np.abs(c.T @ c - np.eye(c.shape[1])).max() < 1e-10


EIGS = ["auto", "householder"]  # Lanczos fast path (verified residual, fallback) and the dense solver


@pytest.mark.parametrize("eig", EIGS)
@pytest.mark.parametrize("n,v,k", [(6, 40, 2), (40, 300, 2), (64, 500, 3), (129, 800, 2), (500, 3000, 4)])
def test_compute_pca_matches_oracle(P, O, n, v, k, eig):
    rng = np.random.default_rng(n * 31 + v)
    x = planted_callsets(rng, n, v, k=max(3, k + 1))
    s = O.similarity_from_dense(x, n)
    ref = O.compute_pca(s, k)
    with P.PcoaEngine(n, eig=eig) as eng:
        eng.accumulate_dense(x)
        comps, lam, nz = eng.compute(k)
        t = eng.timings()
    assert t["eig_method"] == (2 if (eig == "householder" or n < 32) else 1)
    assert nz == ref["nonzero_rows"]
    check_eigenpairs(comps, lam, ref["components"], ref["eigenvalues"], ref["B"])
    # library output is sign-normalised exactly like the oracle's convention
    for c in range(k):
        i = int(np.argmax(np.abs(comps[:, c])))
        assert comps[i, c] > 0


@pytest.fixture(scope="module")
def config2_reference(O):
    synth = load_pkg("synth")
    n, v, seed = 2504, 30000, 1002
    offs = synth.pop_offsets(n)
    thr = synth.thresholds(seed, 0, v)
    x = synth.genotypes(seed, 0, thr, offs)
    s = O.similarity_from_dense_blas(x)
    return n, seed, offs, thr, s, O.compute_pca(s, 2)


@pytest.mark.parametrize("eig", EIGS + ["lanczos"])
def test_compute_pca_config2_sample_count(P, O, config2_reference, eig):
    """N = 2504 with planted 5-population structure: eigenpairs vs the MLlib-path oracle (dgesdd)."""
    n, seed, offs, thr, s, ref = config2_reference
    with P.PcoaEngine(n, eig=eig) as eng:
        eng.accumulate_synthetic(seed, offs, thr, 0)
        assert np.array_equal(eng.gram(), s)
        b, _, _, _ = eng.center()
        assert np.array_equal(b, ref["B"])
        comps, lam, nz = eng.compute(2)
        comps, lam, nz = eng.compute(2)  # second call: workspaces warm
        t = eng.timings()
    check_eigenpairs(comps, lam, ref["components"], ref["eigenvalues"], ref["B"])
    assert nz == ref["nonzero_rows"]
    print("PCoA wall [%s] %.2f ms (method %d, lanczos steps %d; lanczos %.2f, tridiag %.2f, eig %.2f, back %.2f per call)" %
          (eig, 1e3 * t["compute_total_seconds"], t["eig_method"], t["lanczos_steps"], 0.5e3 * t["lanczos_seconds"],
           0.5e3 * t["tridiag_seconds"], 0.5e3 * t["eig_seconds"], 0.5e3 * t["backtransform_seconds"]))


def test_tiny_spectral_gap_falls_back_or_converges_correctly(P, O):
    """Unstructured random genotypes: the top eigenvalues of B sit within a fraction of a percent of each
    other.  Whatever path the engine takes (Lanczos if it verifies, else Householder), the result must be
    a correct eigenpair: residual and eigenvalue are checked (eigenvectors are ill-conditioned here)."""
    rng = np.random.default_rng(77)
    n, v = 200, 5000
    x = (rng.random((v, n)) < 0.3).astype(np.float32)
    s = O.similarity_from_dense(x, n)
    b = O.center_matrix(s)[0]
    lam_ref = np.sort(np.linalg.eigvalsh(b))[::-1]
    for eig in EIGS:
        with P.PcoaEngine(n, eig=eig) as eng:
            eng.accumulate_dense(x)
            comps, lam, _ = eng.compute(2)
        assert np.allclose(lam, lam_ref[:2], rtol=1e-9)
        for c in range(2):
            assert np.linalg.norm(b @ comps[:, c] - lam[c] * comps[:, c]) <= 1e-8 * abs(lam[c])


@pytest.mark.parametrize("n,k", [(300, 15), (300, 11)])
def test_more_components_than_the_first_check_has_rows(P, O, n, k):
    """num_pc >= 11: the first Ritz check (m = 12 by default) has fewer rows than k + 2, so it must wait for k + 2 Krylov
    vectors (it used to index past its candidate list).  Eigenvalues vs a dense LAPACK solve of the oracle's B."""
    rng = np.random.default_rng(5 + k)
    x = planted_callsets(rng, n, 3000, k=6)
    s = O.similarity_from_dense(x, n)
    lam_ref = np.sort(np.linalg.eigvalsh(O.center_matrix(s)[0]))[::-1][:k]
    with P.PcoaEngine(n) as eng:
        eng.accumulate_dense(x)
        comps, lam, _ = eng.compute(k)
        t = eng.timings()
    assert t["eig_method"] in (1, 2) and (t["eig_method"] == 2 or t["lanczos_steps"] >= k + 2)
    assert np.max(np.abs(lam - lam_ref) / np.abs(lam_ref)) < 1e-9
    assert np.abs(comps.T @ comps - np.eye(k)).max() < 1e-9


def test_rank_deficient_inputs_through_the_lanczos_path(P, O):
    """Few variants => B has rank <= V: the Krylov space is exhausted after a handful of steps (breakdown).
    The fast path must either verify its pairs or fall back; the answer must be right either way."""
    rng = np.random.default_rng(31)
    n = 100
    for v in (1, 2, 3, 7):
        x = (rng.random((v, n)) < 0.4).astype(np.float32)
        s = O.similarity_from_dense(x, n)
        b = O.center_matrix(s)[0]
        lam_ref = np.sort(np.linalg.eigvalsh(b))[::-1]
        k = min(2, v)
        with P.PcoaEngine(n) as eng:
            eng.accumulate_dense(x)
            comps, lam, _ = eng.compute(k)
        assert np.allclose(lam, lam_ref[:k], rtol=1e-9, atol=1e-9 * abs(lam_ref[0]))
        for c in range(k):
            assert abs(np.linalg.norm(comps[:, c]) - 1) < 1e-12
            if lam_ref[c] > 1e-6 * lam_ref[0] and (c + 1 >= n or lam_ref[c] - lam_ref[c + 1] > 1e-6 * lam_ref[0]):
                assert np.linalg.norm(b @ comps[:, c] - lam[c] * comps[:, c]) <= 1e-9 * lam_ref[0]
    with P.PcoaEngine(64) as eng:   # no variants at all: B = 0
        comps, lam, nz = eng.compute(2)
    assert nz == 0 and np.allclose(lam, 0) and np.isfinite(comps).all()
    assert np.allclose(np.linalg.norm(comps, axis=0), 1.0)


def test_compute_from_loaded_matrix_entries_and_degenerate_inputs(P, O):
    # computePca(matrixEntries) boundary: S produced elsewhere
    rng = np.random.default_rng(21)
    x = planted_callsets(rng, 33, 200)
    s = O.similarity_from_dense(x, 33)
    ref = O.compute_pca(s, 2)
    with P.PcoaEngine(33) as eng:
        eng.load_gram(s)
        comps, lam, _ = eng.compute(2)
    check_eigenpairs(comps, lam, ref["components"], ref["eigenvalues"], ref["B"])
    # all-zero S (no variants): B = 0, every eigenvalue 0, vectors still unit-norm and finite
    with P.PcoaEngine(10) as eng:
        comps, lam, nz = eng.compute(2)
    assert nz == 0 and np.allclose(lam, 0) and np.isfinite(comps).all()
    assert np.allclose(np.linalg.norm(comps, axis=0), 1.0)
    # N = 1 and N = 2
    with P.PcoaEngine(1) as eng:
        eng.accumulate_callsets([[0]])
        comps, lam, nz = eng.compute(1)
    assert comps.shape == (1, 1) and abs(abs(comps[0, 0]) - 1) < 1e-15 and abs(lam[0]) < 1e-290 and nz == 1
    with P.PcoaEngine(2) as eng:
        eng.accumulate_callsets([[0], [0, 1], [0]])
        comps, lam, _ = eng.compute(2)
    ref = O.compute_pca(np.array([[3, 1], [1, 1]]), 2)
    # the oracle recovers |lambda| as sqrt(s * (N-1)) from the SVD of Cov, which amplifies the
    # rounding of the zero eigenvalue to ~1e-9; compare it loosely and the non-zero one tightly
    assert abs(lam[0] - ref["eigenvalues"][0]) < 1e-12 and abs(lam[1]) < 1e-7
    assert np.abs(align_sign(comps, ref["components"])[:, 0] - ref["components"][:, 0]).max() < 1e-12


def test_larger_sample_count_many_tiles(P, O):
    """N = 6000 (24 x 24 tile grid, 300 upper tiles): Gram vs the BLAS oracle, centring vs the oracle, and the
    eigenpairs by their residual against B (a 6000^2 LAPACK reference would take minutes)."""
    synth = load_pkg("synth")
    n, v, seed = 6000, 5000, 4242
    offs = synth.pop_offsets(n)
    thr = synth.thresholds(seed, 0, v)
    x = synth.genotypes(seed, 0, thr, offs)
    want = O.similarity_from_dense_blas(x)
    with P.PcoaEngine(n) as eng:
        eng.accumulate_synthetic(seed, offs, thr, 0)
        assert np.array_equal(eng.gram(), want)
        b, rs, nz, mm = eng.center()
        bo, rso, nzo, mmo = O.center_matrix(want)
        assert np.array_equal(b, bo) and nz == nzo and mm == mmo
        comps, lam, _ = eng.compute(2)
        t = eng.timings()
    for c in range(2):
        assert abs(np.linalg.norm(comps[:, c]) - 1) < 1e-12
        assert np.linalg.norm(b @ comps[:, c] - lam[c] * comps[:, c]) <= 1e-9 * abs(lam[c])
    assert abs(comps[:, 0] @ comps[:, 1]) < 1e-10 and lam[0] > lam[1] > 0
    print("N=6000 PCoA wall %.2f ms (method %d, %d Lanczos steps)" %
          (1e3 * t["compute_total_seconds"], t["eig_method"], t["lanczos_steps"]))


def test_implicit_centring_matvec_equals_the_materialised_b(P, O):
    """The Lanczos path evaluates B(i,j) on the fly from the integer S (no N x N fp64 matrix); with
    PCOA_EXPLICIT_CENTER=1 it runs on the materialised B.  Same per-entry expression and loop order => the same
    eigenpairs to the last bit; both within the parity bar of the oracle.  Also with a folded int64 part of S."""
    rng = np.random.default_rng(31)
    n, v = 700, 4000
    x = (rng.random((v, n)) < rng.uniform(0.02, 0.4, size=(v, 1))).astype(np.float32)
    x[:, :250] *= (rng.random((v, 1)) < 0.7)      # population structure
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r);"
            "from conftest import load_pkg; P = load_pkg(); x = np.load(sys.argv[1]);"
            "e = P.PcoaEngine(x.shape[1]); e.accumulate_dense(x); e.accumulate_dense(x[:100]);"
            "c, l, nz = e.compute(3); t = e.timings(); assert t['eig_method'] == 1;"
            "np.save(sys.argv[2], np.concatenate([c.ravel(), l]))") % (ROOT, os.path.join(ROOT, "tests"))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "x.npy"), x)
        res = {}
        for name, extra in (("implicit", {}), ("explicit", {"PCOA_EXPLICIT_CENTER": "1"}),
                            ("implicit64", {"PCOA_DEBUG_FOLD_THRESHOLD": "1000"}),
                            ("explicit64", {"PCOA_EXPLICIT_CENTER": "1", "PCOA_DEBUG_FOLD_THRESHOLD": "1000"})):
            env = dict(os.environ, **extra)
            subprocess.check_call([sys.executable, "-c", code, os.path.join(td, "x.npy"), os.path.join(td, name + ".npy")],
                                  env=env)
            res[name] = np.load(os.path.join(td, name + ".npy"))
    assert np.array_equal(res["implicit"], res["explicit"])
    assert np.array_equal(res["implicit64"], res["explicit64"])
    assert np.abs(res["implicit"] - res["implicit64"]).max() < 1e-9
    s = O.similarity_from_dense(x, n) + O.similarity_from_dense(x[:100], n)
    ref = O.compute_pca(s, 3)
    got = res["implicit"][:3 * n].reshape(n, 3)
    assert np.abs(align_sign(got, ref["components"]) - ref["components"]).max() < EIG_TOL


# ------------------------------------------------------------------------------------------ multi-GPU plumbing
def test_native_rccl_allreduce_single_rank(P, O):
    rng = np.random.default_rng(4)
    x = (rng.random((100, 50)) < 0.3).astype(np.float32)
    want = O.similarity_from_dense(x, 50)
    with P.PcoaEngine(50) as eng:
        eng.accumulate_dense(x)
        comm = eng.comm_init(eng.comm_unique_id(), 0, 1)
        eng.allreduce_rccl(comm)
        assert np.array_equal(eng.gram(), want)
        eng.accumulate_dense(x)                     # keeps accumulating after the reduce
        assert np.array_equal(eng.gram(), 2 * want)
        eng.comm_destroy(comm)


def test_torch_distributed_rccl_allreduce_single_rank(P, O):
    import torch
    import torch.distributed as td
    dist = load_pkg("dist")
    rng = np.random.default_rng(6)
    x = (rng.random((80, 40)) < 0.3).astype(np.float32)
    want = O.similarity_from_dense(x, 40)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    td.init_process_group("nccl", rank=0, world_size=1)
    try:
        with P.PcoaEngine(40) as eng:
            eng.accumulate_dense(x)
            scratch = torch.empty((40, 40), dtype=torch.int64, device="cuda")
            # world size 1 short-circuits; exercise the export -> all_reduce -> import path explicitly
            eng.export_device(scratch.data_ptr())
            eng.sync()
            td.all_reduce(scratch)
            torch.cuda.synchronize()
            assert np.array_equal(scratch.cpu().numpy(), want)
            eng.import_device(scratch.data_ptr())
            assert np.array_equal(eng.gram(), want)
            assert dist.allreduce_engine(eng) is None
            # the native communicator bootstrapped over torch.distributed (bench.py's N>1 path)
            native = dist.NativeComm(eng)
            eng.accumulate_dense(x)
            native.allreduce()
            assert np.array_equal(eng.gram(), 2 * want)
            native.close()
    finally:
        td.destroy_process_group()


# ------------------------------------------------------------------------------------------ driver
def test_driver_main_end_to_end(P, O, tmp_path, capsys):
    vp = load_pkg("variants_pca")
    g = load_golden("pops40")
    ids = [str(s) for s in g["callset_ids"]]
    names = ["NA%05d" % (40 - i) for i in range(40)]
    path = str(tmp_path / "pops40_input.npz")
    np.savez(path, callset_ids=np.array(ids), callset_names=np.array(names), sample_idx=g["sample_idx"],
             row_offsets=g["row_offsets"])
    rc = vp.main(["--input-path", path, "--output-path", str(tmp_path / "res"), "--spark-master", "local[4]"])
    assert rc == 0
    out = capsys.readouterr().out.splitlines()
    assert out[0] == "Matrix size: 40."
    assert out[1] == "Non zero rows in matrix: 40 / 40."
    rows = [l.split("\t") for l in out[2:]]
    assert len(rows) == 40 and [r[0] for r in rows] == sorted(names)
    ref = O.compute_pca(g["similarity"], 2)
    by_name = dict((r[0], r) for r in rows)
    got = np.array([[float(by_name[names[i]][2]), float(by_name[names[i]][3])] for i in range(40)])
    assert np.abs(align_sign(got, ref["components"]) - ref["components"]).max() < EIG_TOL
    assert by_name[names[0]][1] == ids[0].split("-")[0]
    saved = open(str(tmp_path / "res") + "-pca.tsv").read().splitlines()
    assert len(saved) == 40 and saved[0].split("\t")[0] == sorted(names)[0]


def test_driver_similarity_matrix_stream_has_the_nonzero_keys_of_the_full_matrix(P, O):
    vp = load_pkg("variants_pca")
    g = load_golden("zerorow9")          # sample 4 never varies: its row and column are absent from the stream form
    n = int(g["n_samples"])
    ids = [str(s) for s in g["callset_ids"]]
    conf = vp.PcaConf([])
    drv = vp.VariantsPcaDriver(conf, dict((c, i) for i, c in enumerate(ids)), dict(zip(ids, ids)), [])
    offs = g["row_offsets"]
    callsets = [list(g["sample_idx"][offs[k]:offs[k + 1]]) for k in range(len(offs) - 1)]
    entries = dict(drv.getSimilarityMatrixStream(callsets))
    s = g["similarity"]
    assert entries == dict(((i, j), int(s[i, j])) for i in range(n) for j in range(n) if s[i, j] != 0)
    assert all(4 not in k for k in entries) and len(entries) < n * n


# ------------------------------------------------------------------------------------------ compiled host
def _write_vcf(path, sample_names, records, gz=False):
    import gzip
    opener = gzip.open if gz else open
    with opener(path, "wt") as f:
        f.write("##fileformat=VCFv4.2\n")
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(sample_names) + "\n")
        for (chrom, pos, ref, alt, af, gts) in records:
            info = "AF=%s" % af if af is not None else "."
            f.write("%s\t%d\t.\t%s\t%s\t.\tPASS\t%s\tGT\t%s\n" % (chrom, pos, ref, alt, info, "\t".join(gts)))


def _make_cohorts(tmp_path, seed=5):
    """Three small variant sets over partly shared sites, with planted structure."""
    rng = np.random.default_rng(seed)
    bases = "ACGT"
    sites = []
    for k in range(160):
        ref = bases[rng.integers(0, 4)]
        alt = bases[(bases.index(ref) + 1 + rng.integers(0, 3)) % 4]
        sites.append(("chr17" if k % 2 else "17", 41196312 + 7 * k, ref, alt))
    sets = []
    for d, nsamp in enumerate((11, 9, 7)):
        names = ["D%dS%02d" % (d, i) for i in range(nsamp)]
        pops = np.arange(nsamp) % 2
        recs = []
        for k, (chrom, pos, ref, alt) in enumerate(sites):
            if d == 1 and k % 5 == 0:
                continue                      # site missing from set 1 -> not joined
            a = alt if not (d == 2 and k % 7 == 0) else alt + "T"   # different ALT in set 2 -> different key
            p = np.where(pops == (k % 2), 0.6, 0.1)
            gts = []
            for i in range(nsamp):
                g = (int(rng.random() < p[i]), int(rng.random() < p[i]))
                gts.append("%d|%d" % g if rng.random() > 0.05 else "./.")
            recs.append((chrom, pos, ref, a, "%.3f" % rng.uniform(0.0, 0.3) if k % 11 else None, gts))
        recs.append(("X", 5, "A", "C", "0.2", ["1|1"] * nsamp))                 # dropped contig
        recs.append(("17", 50000000, "A", "C", "0.2", ["1|1"] * nsamp))          # outside --references
        path = str(tmp_path / ("set%d.vcf%s" % (d, ".gz" if d == 1 else "")))
        _write_vcf(path, names, recs, gz=(d == 1))
        sets.append(path)
    return sets


@pytest.mark.parametrize("nsets,extra", [(1, []), (2, []), (3, []), (2, ["--min-allele-frequency", "0.1"])])
def test_cpp_driver_matches_python_driver(P, tmp_path, capsys, nsets, extra):
    exe = os.path.join(ROOT, "spark-examples_amd", "variants_pca_driver")
    if not os.path.exists(exe):   # normally built by __graft_entry__.build(); g++ exists on the GPU image too
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "spark-examples_amd", "host")])
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    vp = load_pkg("variants_pca")
    sets = _make_cohorts(tmp_path)[:nsets]
    args = ["--input-path"] + sets + ["--references", "chr17:41196311:41277499", "--output-path", str(tmp_path / "py")] + extra
    assert vp.main(args) == 0
    py_out = capsys.readouterr().out.splitlines()
    cpp_args = [a if a != str(tmp_path / "py") else str(tmp_path / "cpp") for a in args]
    res = subprocess.run([exe] + cpp_args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    assert res.returncode == 0, res.stderr
    cpp_out = res.stdout.splitlines()
    head_py = [l for l in py_out if "\t" not in l]
    head_cpp = [l for l in cpp_out if "\t" not in l]
    assert head_py == head_cpp
    rows_py = [l.split("\t") for l in py_out if "\t" in l]
    rows_cpp = [l.split("\t") for l in cpp_out if "\t" in l]
    assert [r[:2] for r in rows_py] == [r[:2] for r in rows_cpp] and len(rows_py) == sum((11, 9, 7)[:nsets])
    a = np.array([[float(r[2]), float(r[3])] for r in rows_py])
    b = np.array([[float(r[2]), float(r[3])] for r in rows_cpp])
    assert np.abs(align_sign(b, a) - a).max() < 1e-9
    # identical formatting of identical doubles (Double.toString rules)
    fmt = load_pkg("variants_pca").java_double_to_string
    for r in rows_cpp:
        assert fmt(float(r[2])) == r[2] and fmt(float(r[3])) == r[3]
    assert open(str(tmp_path / "cpp") + "-pca.tsv").read().count("\n") == len(rows_cpp)
    if nsets == 1:
        # --spark-output-layout (r06): <output-path>-pca.tsv as the directory saveAsTextFile leaves (VariantsPca.scala:241-245)
        flat = open(str(tmp_path / "cpp") + "-pca.tsv").read()
        sp_args = [a if a != str(tmp_path / "py") else str(tmp_path / "spark_cpp") for a in args] + ["--spark-output-layout"]
        r3 = subprocess.run([exe] + sp_args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
        assert r3.returncode == 0, r3.stderr
        d = str(tmp_path / "spark_cpp") + "-pca.tsv"
        assert sorted(os.listdir(d)) == ["_SUCCESS", "part-00000"] and open(os.path.join(d, "part-00000")).read() == flat
        assert subprocess.run([exe] + sp_args, stdout=subprocess.PIPE, stderr=subprocess.PIPE).returncode != 0   # exists: error, as Spark
        py_args = [a if a != str(tmp_path / "py") else str(tmp_path / "spark_py") for a in args] + ["--spark-output-layout"]
        assert vp.main(py_args) == 0
        capsys.readouterr()
        dpy = str(tmp_path / "spark_py") + "-pca.tsv"
        assert sorted(os.listdir(dpy)) == ["_SUCCESS", "part-00000"]
        assert open(os.path.join(dpy, "part-00000")).read().count("\n") == len(rows_cpp)
    # r06: joins / merges are streamed -- every set once into hash-partitioned spill files, one key partition in memory at a
    # time (VariantsPca.scala:115-148 is a shuffle) -- and so is a single VCF for several engines; all must give the S of the
    # in-memory path, whatever the partition count
    if nsets >= 2:
        assert "key partitions" in res.stderr, res.stderr
    mats = {}
    for tag, more in (("stream64", []), ("stream3", ["--join-partitions", "3"]), ("stream1", ["--join-partitions", "1"]),
                      ("memory", ["--no-stream"]), ("stream_two_engines", ["--gpus", "2", "--gpu-map", "0,0", "--join-partitions", "5"])):
        dump = str(tmp_path / (tag + ".bin"))
        r2 = subprocess.run([exe] + cpp_args + more + ["--dump-similarity", dump], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                            universal_newlines=True)
        assert r2.returncode == 0, (tag, r2.stderr)
        partitioned = tag != "memory" and (nsets >= 2 or "two_engines" in tag)
        assert ("key partitions" in r2.stderr) == partitioned, (tag, r2.stderr)
        mats[tag] = np.fromfile(dump, dtype="<i8")
    for tag in mats:
        assert np.array_equal(mats[tag], mats["memory"]), tag
    if True:
        assert not [f for f in os.listdir(os.environ.get("TMPDIR", "/tmp")) if f.startswith("pcoa_join_")]   # spill files removed


# ------------------------------------------------------------------------------------------ measurement contract
def test_bench_emits_one_json_line_with_the_contract_fields(P):
    """bench.py at a reduced size: ONE JSON line on stdout with the driver's fields, the roofline object of the
    dominant kernel (live HIP-event timing) and the bounded CPU baseline; the CPU sample doubles as a parity check."""
    import json
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2",
                                   "--warmup", "1", "--variants", "50000", "--no-extras", "--pcoa-reps", "1"],
                                  universal_newlines=True)
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "variants/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 50000 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6
    for r in (d["roofline"], d["roofline_other"]):
        assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["avg_launch_ms"] > 0 and "traffic" in r
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert d["parity_vs_cpu_sample"] is True
    assert d["dtype"].startswith("fp4") and d["pcoa_wall_ms"] > 0


# ---- r06: band Lanczos, the fallback for clustered leading eigenvalues (VERDICT r05 Missing 5 / Next 4) -----------------------
def _block_constant_similarity(n, pops, within, across, bump):
    """S[i, j] = within (+ bump inside population 0) if i and j belong to the same population, else across: the centred matrix
    has the between-population contrasts as its only non-zero eigen-directions, (len(pops) - 1) of them in one cluster that
    `bump` splits by a relative ~ bump / (within - across)."""
    offs = np.concatenate([[0], np.cumsum(pops)])
    s = np.full((n, n), across, dtype=np.int64)
    for p in range(len(pops)):
        s[offs[p]:offs[p + 1], offs[p]:offs[p + 1]] = within + (bump if p == 0 else 0)
    return s, offs


def _reduced_eigenvalues(pops, within, across, bump):
    """Eigenvalues of B = J S J for a block-constant S from the len(pops)-dimensional problem on the normalised population
    indicators (exact structure, long double arithmetic)."""
    sz = np.asarray(pops, dtype=np.longdouble)
    nn = sz.sum()
    k = len(pops)
    sr = np.empty((k, k), dtype=np.longdouble)
    for p in range(k):
        for q in range(k):
            val = (within + (bump if p == 0 else 0)) if p == q else across
            sr[p, q] = np.longdouble(val) * np.sqrt(sz[p] * sz[q])
    root = np.sqrt(sz)
    jr = np.eye(k, dtype=np.longdouble) - np.outer(root, root) / nn
    br = (jr @ sr @ jr).astype(np.float64)
    return np.sort(np.linalg.eigvalsh(br))[::-1]


def _centred_matvec_host(sf, u):
    """B u = (S - m 1^T - 1 m^T + mm 1 1^T) u for a float64 S, without forming B."""
    n = sf.shape[0]
    rs = sf.sum(axis=1)
    mean = rs / n
    mm = rs.sum() / n / n
    return sf @ u - mean * u.sum() - (mean @ u) + mm * u.sum()


def test_clustered_leading_eigenvalues_at_n_20000_return_residual_verified_pairs(P):
    """N = 20,000 > 16,384, where the dense fallback is disabled: S loaded through pcoa_gram_load_i64 whose centred matrix has
    THREE leading eigenvalues within a relative 2e-9 of each other on top of a full-rank bulk (a random symmetric +-2
    perturbation of a block-constant matrix; bulk eigenvalues of a few hundred against 5e12).  r05 answered
    PCOA_ERR_NOT_CONVERGED in such a case where the reference's dgesdd returns an answer (VariantsPca.scala:224-227).  The
    pairs that come back passed their true residual on the device; here they are checked again on the host: residual,
    eigenvalues against the exact 4 x 4 reduced problem (the perturbation moves them by O(1)), orthonormality.  Both the
    default path (single vector first) and the band iteration alone."""
    n = 20000
    pops = [5000, 5000, 5000, 5000]
    within, across, bump = 2000000000, 1000000000, 3
    s, offs = _block_constant_similarity(n, pops, within, across, bump)
    rng = np.random.default_rng(8)
    for r0 in range(0, n, 2000):                       # s += E + E^T, E in {-1, 0, 1}: full rank, symmetric, int32-sized
        e = rng.integers(-1, 2, size=(2000, n), dtype=np.int8)
        s[r0:r0 + 2000, :] += e
        s[:, r0:r0 + 2000] += e.T
    lam_ref = _reduced_eigenvalues(pops, within, across, bump)
    assert abs(lam_ref[0] - lam_ref[2]) < 1e-8 * lam_ref[0] and lam_ref[0] > lam_ref[1]   # the cluster
    sf = s.astype(np.float64)
    for eig in (None, "band"):
        with P.PcoaEngine(n, eig=eig) as eng:
            eng.load_gram(s)
            comps, lam, nz = eng.compute(2)
            t = eng.timings()
        assert t["eig_method"] == 1, t
        if eig == "band":
            assert t["lanczos_block_steps"] > 0
        assert t["gram_i64_live"] == 0 and t["matvec_form"] == 1      # the loaded counts fit int32: upper-triangle mat-vec
        assert nz == n
        assert np.max(np.abs(lam - lam_ref[:2]) / lam_ref[:2]) < 1e-11
        for c in range(2):
            u = comps[:, c]
            assert np.linalg.norm(_centred_matvec_host(sf, u) - lam[c] * u) <= 1e-9 * abs(lam[c])
            assert abs(np.linalg.norm(u) - 1.0) < 1e-12
        assert abs(comps[:, 0] @ comps[:, 1]) < 1e-9


def test_exactly_degenerate_leading_eigenvalue_needs_and_gets_the_band_iteration(P):
    """Three populations of equal size and equal structure: the leading eigenvalue of B has multiplicity 2 EXACTLY.  A single
    start vector sees one direction of that eigenspace (the documented limit of the default path, pcoa.h PCOA_FLAG_EIG_BAND);
    the band iteration (eig='band') returns two orthonormal vectors of the eigenspace, both with the leading eigenvalue."""
    n = 1536
    pops = [512, 512, 512]
    s, _ = _block_constant_similarity(n, pops, 900, 100, 0)
    lam_ref = _reduced_eigenvalues(pops, 900, 100, 0)
    assert abs(lam_ref[0] - lam_ref[1]) < 1e-9 * lam_ref[0] and lam_ref[2] < 1e-6 * lam_ref[0]
    sf = s.astype(np.float64)
    with P.PcoaEngine(n, eig="band") as eng:
        eng.load_gram(s)
        comps, lam, _ = eng.compute(2)
        t = eng.timings()
    assert t["eig_method"] == 1 and t["lanczos_block_steps"] > 0
    assert np.max(np.abs(lam - lam_ref[:2]) / lam_ref[:2]) < 1e-12
    for c in range(2):
        assert np.linalg.norm(_centred_matvec_host(sf, comps[:, c]) - lam[c] * comps[:, c]) <= 1e-10 * abs(lam[c])
    assert abs(comps[:, 0] @ comps[:, 1]) < 1e-10


def test_device_side_check_of_row_offsets_names_them(P):
    """Carrier lists as device arrays are validated on the device: row offsets that decrease are reported as what they are
    (PCOA_ERR_INVALID_ARG, r06: r05 reported 'callset index -1'), S unchanged."""
    import torch
    n = 300
    rng = np.random.default_rng(2)
    x = planted_callsets(rng, n, 400)
    idx, offs = _csr_of(x)
    with P.PcoaEngine(n) as eng:
        eng.accumulate_calls(idx, offs)
        before = eng.gram()
        bad = offs.copy()
        bad[200] = bad[199] - 3
        ti = torch.from_numpy(idx).cuda()
        to = torch.from_numpy(bad).cuda()
        with pytest.raises(P.PcoaError) as err:
            eng.accumulate_calls_tensors(ti, to)
            eng.sync()
        assert err.value.code == -1 and "row_offsets" in str(err.value)
        assert np.array_equal(eng.gram(), before)


def test_band_iteration_alone_matches_the_oracle_on_ordinary_spectra(tmp_path):
    """PCOA_LANCZOS_BAND=2: only the band iteration runs (planted populations, unstructured genotypes with gaps of a fraction of
    a percent, a rank-7 matrix).  Eigenvalues and -- where the gaps allow -- eigenvectors against the oracle."""
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import importlib
from conftest import load_oracle, planted_callsets, align_sign
P = importlib.import_module("spark-examples_amd"); O = load_oracle()
rng = np.random.default_rng(19)
out = {}
for tag, n, x in (("planted", 700, planted_callsets(rng, 700, 3000, k=4)),
                  ("flat", 200, (rng.random((5000, 200)) < 0.3).astype(np.float32)),
                  ("rank7", 100, (rng.random((7, 100)) < 0.4).astype(np.float32))):
    s = O.similarity_from_dense(x, n)
    b = O.center_matrix(s)[0]
    ref = O.compute_pca(s, 2)
    with P.PcoaEngine(n, eig="lanczos") as eng:
        eng.accumulate_dense(x)
        comps, lam, _ = eng.compute(2)
        t = eng.timings()
    res = max(np.linalg.norm(b @ comps[:, c] - lam[c] * comps[:, c]) / abs(lam[c]) for c in range(2))
    got = align_sign(comps, ref["components"])
    out[tag] = [t["lanczos_block_steps"], float(np.max(np.abs(lam - ref["eigenvalues"]) / np.abs(ref["eigenvalues"]))), float(res),
                float(max(np.linalg.norm(got[:, c] - ref["components"][:, c]) for c in range(2)))]
np.savez(sys.argv[1], **out)
""" % (ROOT, os.path.join(ROOT, "tests"))
    # the second run holds the basis to 48 vectors: the flat spectrum (132 columns in one go) then needs THICK RESTARTS -- the
    # 8 Ritz vectors of largest |theta| replace the processed part of the basis and the iteration goes on
    for tag_env, env in (("whole", {}), ("restarts", {"PCOA_LANCZOS_BAND_MMAX": "48"})):
        out = str(tmp_path / ("band_%s.npz" % tag_env))
        subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, PCOA_LANCZOS_BAND="2", **env))
        r = np.load(out)
        for tag in ("planted", "flat", "rank7"):
            steps, dlam, res, dvec = r[tag]
            assert steps > 0 and dlam < 1e-9 and res < 1e-8, (tag_env, tag, r[tag])
        assert r["planted"][3] < 1e-6      # clear gaps: the vectors themselves at the north_star tolerance
        if tag_env == "restarts":
            assert r["flat"][0] > 48       # more columns than the basis holds: at least one restart


def test_caller_supplied_operator_with_a_degenerate_leading_pair(P):
    """pcoa_lanczos_with_matvec (the strip-owner path: no dense fallback at all): a diagonal operator whose two leading
    eigenvalues differ by a relative 1e-9 and a third copy 1e-7 below -- r05 returned PCOA_ERR_NOT_CONVERGED (its pairs sat at
    a true residual of 1e-16 from m = 144 on and were refused for their gap until the steps ran out)."""
    import torch
    n = 4096
    d = torch.linspace(0.0, 0.9, n, dtype=torch.float64, device="cuda:0")
    d[17], d[1900], d[4000] = 1.0, 1.0 - 1e-9, 1.0 - 1e-7
    dd = d.cpu().numpy()
    for eig in (None, "band"):   # default: the single vector resolves this cluster itself (accepted at the fp64 residual floor)
        with P.PcoaEngine(n, eig=eig) as eng:
            comps, lam = eng.lanczos(lambda v: d * v, 2)
            t = eng.timings()
        assert (t["lanczos_block_steps"] > 0) == (eig == "band")
        assert abs(lam[0] - 1.0) < 1e-12 and abs(lam[1] - (1.0 - 1e-9)) < 1e-12
        for c in range(2):
            assert np.linalg.norm(dd * comps[:, c] - lam[c] * comps[:, c]) < 1e-10
        assert abs(comps[:, 0] @ comps[:, 1]) < 1e-9


@pytest.mark.parametrize("n,n_pops,calls", [(37, 1, (1, 127, 300)), (1001, 7, (130, 5, 4000, 129)), (2504, 5, (30000, 77)),
                                            (4100, 64, (3000, 1029))])
def test_synthetic_model_written_straight_into_the_operand_equals_its_host_twin(P, O, n, n_pops, calls):
    """pcoa_accumulate_synthetic on the k-bits operand (r06: synth_kbits_kernel, no fp32 tile, no pre-pass): S equals the oracle's
    on the host twin of the generator -- sample counts that are no multiple of 4, call lengths that end in part-filled blocks of
    128 variants, 1 to 64 populations, several calls -- and the r05 path through the staging tile (PCOA_SYNTH_TILE=1 in a
    child process) gives the same matrix."""
    synth = load_pkg("synth")
    seed = 4242 + n
    sizes = np.linspace(1.0, 2.0, n_pops)
    offs = synth.pop_offsets(n, sizes=sizes)
    v_total = sum(calls)
    thr = synth.thresholds(seed, 0, v_total, n_pops=n_pops)
    x = synth.genotypes(seed, 0, thr, offs)
    want = O.similarity_from_dense(x, n)
    with P.PcoaEngine(n) as eng:
        v0 = 0
        for c in calls:
            eng.accumulate_synthetic(seed, offs, thr[v0:v0 + c], v0)
            v0 += c
        got = eng.gram()
        t = eng.timings()
    assert np.array_equal(got, want)
    assert t["pack_launches"] == 0 and t["synth_seconds"] > 0      # no staging tile, no pre-pass
    if n == 1001:
        code = r"""
import sys, numpy as np, importlib
sys.path.insert(0, %r)
P = importlib.import_module("spark-examples_amd"); synth = importlib.import_module("spark-examples_amd.synth")
n, n_pops, seed, calls = 1001, 7, 4242 + 1001, (130, 5, 4000, 129)
offs = synth.pop_offsets(n, sizes=np.linspace(1.0, 2.0, n_pops)); thr = synth.thresholds(seed, 0, sum(calls), n_pops=n_pops)
with P.PcoaEngine(n) as eng:
    v0 = 0
    for c in calls:
        eng.accumulate_synthetic(seed, offs, thr[v0:v0 + c], v0); v0 += c
    np.save(sys.argv[1], eng.gram()); assert eng.timings()["pack_launches"] > 0
""" % ROOT
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "tile.npy")
            subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, PCOA_SYNTH_TILE="1"))
            assert np.array_equal(np.load(out), want)


def test_xcd_k_segment_launch_form_of_the_contraction_stays_exact(tmp_path):
    """PCOA_KBITS_MODE=5 (r06, measured and not the default: profiles/r07f): every XCD takes one eighth of the k-range for all
    tiles.  The launch form stays in the kernel behind the knob, so it stays bit-exact: S against the oracle on a cohort long
    enough for the segments to exist (>= 512 stages of 128 variants), with a ragged tail."""
    code = r"""
import sys, numpy as np, importlib
sys.path.insert(0, %r); sys.path.insert(0, %r)
from conftest import load_oracle, int_gram
P = importlib.import_module("spark-examples_amd"); ingest = importlib.import_module("spark-examples_amd.ingest")
rng = np.random.default_rng(6)
n, v = 1300, 70003
x = (rng.random((v, n)) < 0.12).astype(np.uint8)
with P.PcoaEngine(n) as eng:
    eng.accumulate_bits(ingest.pack_bits(x))
    got = eng.gram()
    t = eng.timings()
np.savez(sys.argv[1], same=np.array_equal(got, int_gram(x)), even=t["evensplit_launches"])
""" % (ROOT, os.path.join(ROOT, "tests"))
    out = str(tmp_path / "m5.npz")
    subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, PCOA_KBITS_MODE="5"))
    r = np.load(out)
    assert bool(r["same"]) and int(r["even"]) >= 1


def test_bench_with_two_ranks_runs_its_multi_rank_path_end_to_end(tmp_path):
    """bench.py --gpus 2 has never met a node: its multi-rank branch -- the self-spawn under torch.distributed.run, the sharded
    cohort, finalize -> reduction -> max-over-ranks timing, the telemetry the line must carry (VERDICT r05 Next 5) -- is run here
    with two REAL ranks on the one GPU of the box over a gloo wire (--rank-devices 0,0 --dist-backend gloo; RCCL refuses two ranks
    on one device).  A plumbing test, not a measurement: one JSON line, n_gpus = 2, both ranks' elapsed times, the reduction
    step's wall, value = the variants of BOTH ranks over the max-over-ranks time."""
    import json
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_PORT=str(port))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rank-devices", "0,0", "--dist-backend", "gloo",
                          "--steps", "3", "--warmup", "1", "--variants", "60000", "--no-extras", "--no-cpu-baseline", "--pcoa-reps", "1"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, env=env, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert d["config"]["allreduce"] == "torch" and d["config"]["dist_backend"] == "gloo" and d["rccl_ranks"] is None
    assert len(d["rank_elapsed_s"]) == 2 and 0 < d["rank_elapsed_min_s"] <= d["rank_elapsed_max_s"]
    assert d["allreduce_ms"] > 0 and d["allreduce_event_ms"] is None
    assert abs(d["value"] - 2 * 60000 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
    assert abs(d["n1_equivalent_value"] - d["value"] / 2) < 1e-6 * d["value"]
    assert d["nonzero_rows"] == 2504 and d["eigenvalues"][0] > d["eigenvalues"][1] > 0 and d["pcoa_wall_ms"] > 0
