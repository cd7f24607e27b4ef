"""GPU tests of the strip-owner mode (SURVEY 8e / BASELINE configs[4]: S tiled across HBMs).  One GPU plays every
owner in turn: the strips must tile the full engine's S bit for bit, and computePca over the strips must agree with the
single-engine path and the oracle."""
import numpy as np
import pytest

from conftest import align_sign, int_gram, load_oracle, load_pkg, planted_callsets

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    return load_pkg()


@pytest.fixture(scope="module")
def O():
    return load_oracle()


def test_strips_tile_the_full_similarity_matrix_and_give_the_same_pca(P, O):
    """N = 6000 (24 tile columns), three owners with ragged, non-tile-aligned strips, each fed the whole cohort through a
    different boundary (fp32 tile, bitsets, carrier lists)."""
    strips = load_pkg("strips")
    ingest = load_pkg("ingest")
    rng = np.random.default_rng(60)
    n, v = 6000, 4000
    x = planted_callsets(rng, n, v, k=4)
    with P.PcoaEngine(n) as full:
        full.accumulate_dense(x)
        s_full = full.gram()
        comps_full, lam_full, nz_full = full.compute(2)
    ranges = [(0, 1000), (1000, 2777), (3777, 2223)]
    offs = np.concatenate([[0], np.cumsum(x.sum(axis=1, dtype=np.int64))]).astype(np.int64)
    idx = np.nonzero(x)[1].astype(np.int32)
    feeders = [lambda e: e.accumulate_dense(x), lambda e: e.accumulate_bits(ingest.pack_bits(x)),
               lambda e: e.accumulate_calls(idx, offs)]
    owners = [P.PcoaEngine(n, strip=r) for r in ranges]
    try:
        for e, feed in zip(owners, feeders):
            feed(e)
        tiled = np.concatenate([e.gram() for e in owners], axis=1)
        assert tiled.shape == (n, n) and np.array_equal(tiled, s_full)             # bit for bit, both triangles
        for e, (c0, w) in zip(owners, ranges):
            assert np.array_equal(e.strip_col_sums(), s_full[:, c0:c0 + w].sum(axis=0).astype(np.float64))
            assert np.array_equal(e.gram_block(17, 5, 40, min(30, w - 5)), s_full[17:57, c0 + 5:c0 + 5 + min(30, w - 5)])
        comps, lam, nz = strips.compute_pca_over_strips(owners, 2)
        assert nz == nz_full
        assert np.max(np.abs(lam - lam_full) / np.abs(lam_full)) < 1e-10
        assert np.abs(align_sign(comps, comps_full) - comps_full).max() < 1e-9
        # against the oracle: eigenpairs of the ORACLE's centred matrix by their residual (a 6000^2 LAPACK reference takes
        # most of a minute; the full engine is held to it at this N in test_larger_sample_count_many_tiles the same way)
        bo = O.center_matrix(s_full)[0]
        for c in range(2):
            assert abs(np.linalg.norm(comps[:, c]) - 1) < 1e-12
            assert np.linalg.norm(bo @ comps[:, c] - lam[c] * comps[:, c]) <= 1e-9 * abs(lam[c])
        assert abs(comps[:, 0] @ comps[:, 1]) < 1e-10 and lam[0] > lam[1] > 0
        # a strip is additive and resumable like the full matrix; multiplicities take its int8 path
        e = owners[1]
        before = e.gram()
        xm = x[:50].copy()
        xm[3, 1500] = 4.0
        e.accumulate_dense(xm)
        want = before + int_gram(xm)[:, 1000:3777]
        assert np.array_equal(e.gram(), want)
        e.load_gram(before)
        assert np.array_equal(e.gram(), before)
        # what a strip owner cannot do says so
        with pytest.raises(P.PcoaError) as ei:
            e.compute(2)
        assert ei.value.code == -8
        with pytest.raises(P.PcoaError):
            e.center()
    finally:
        for e in owners:
            e.close()


def test_bitset_shards_reach_every_strip_owner_through_the_feeding_step(P):
    """strips.feed_owners_from_variant_shards on the device (single process: a chunked feed of device-resident bitsets);
    the two-rank exchange is covered on gloo in tests/test_strips_cpu.py"""
    import torch
    strips = load_pkg("strips")
    ingest = load_pkg("ingest")
    rng = np.random.default_rng(61)
    n, v = 1100, 3000
    x = planted_callsets(rng, n, v, k=3)
    want = int_gram(x)
    bits = torch.from_numpy(ingest.pack_bits(x).view(np.int32)).cuda()
    owners = [P.PcoaEngine(n, strip=r) for r in strips.strip_ranges(n, 2)]
    try:
        assert strips.feed_owners_from_variant_shards(owners, bits, chunk_variants=1024) == v
        assert np.array_equal(np.concatenate([e.gram() for e in owners], axis=1), want)
    finally:
        for e in owners:
            e.close()


def test_strip_creation_is_validated(P):
    for bad in ((-1, 10), (0, 0), (90, 20), (100, 1)):
        with pytest.raises(P.PcoaError):
            P.PcoaEngine(100, strip=bad)
    with pytest.raises(P.PcoaError):
        P.PcoaEngine(100, strip=(0, 50), gram_kernel="f32")
    with P.PcoaEngine(100, strip=(99, 1)) as e:
        e.accumulate_callsets([[0, 99], [99], [5, 6]])
        assert e.gram().ravel().tolist() == [1 if i == 0 else (2 if i == 99 else 0) for i in range(100)]
    with P.PcoaEngine(100) as e:
        with pytest.raises(P.PcoaError):
            e.strip_col_sums()


def test_resident_centring_does_not_outlive_the_matrix_it_was_computed_for(P):
    """pcoa.h: the row means of pcoa_strip_set_centering stay resident only until S changes.  pcoa_reset and
    pcoa_gram_load_i64 (checkpoint resume) replace S as well: a later pcoa_strip_matvec_device must refuse (PCOA_ERR_STATE)
    instead of multiplying the new S by the old row means."""
    import torch
    n = 300
    with P.PcoaEngine(n, strip=(40, 120)) as e:
        e.accumulate_callsets([[1, 50, 60], [50, 299], [41, 42, 43]])
        v = torch.ones(n, dtype=torch.float64, device="cuda:0")
        e.strip_set_centering(np.full(n, 0.25), 0.125)
        assert e.strip_matvec_device(v).shape[0] == 120
        e.load_gram(e.gram())
        with pytest.raises(P.PcoaError):
            e.strip_matvec_device(v)
        e.strip_set_centering(np.full(n, 0.25), 0.125)
        e.strip_matvec_device(v)
        e.reset()
        with pytest.raises(P.PcoaError):
            e.strip_matvec_device(v)


def test_strips_at_biobank_sample_count_match_the_single_engine(P):
    """N = 100,000 (BASELINE configs[3] sample count; 391 tile rows, many bands): two strip owners with a cut that is
    not tile-aligned against ONE engine holding all of S (40 GB) -- blocks on both sides of the diagonal and of the cut,
    the row sums, and computePca over the strips against the single-engine Lanczos path."""
    strips = load_pkg("strips")
    synth = load_pkg("synth")
    n, v, seed, chunk = 100000, 32768, 1005, 16384
    offs = synth.pop_offsets(n)
    cut = 49999
    full = P.PcoaEngine(n)
    owners = [P.PcoaEngine(n, strip=(0, cut)), P.PcoaEngine(n, strip=(cut, n - cut))]
    try:
        for v0 in range(0, v, chunk):
            thr = synth.thresholds(seed, v0, chunk)
            for e in [full] + owners:
                e.accumulate_synthetic(seed, offs, thr, v0)
        for (r0, c0) in ((0, 0), (70000, 123), (123, 70000), (49900, 49900), (99700, 99700), (60000, 49990), (20000, 30000)):
            want = full.gram_block(r0, c0, 290, 290)
            got = np.zeros_like(want)
            for e, (s0, w) in zip(owners, ((0, cut), (cut, n - cut))):
                a, b = max(c0, s0), min(c0 + 290, s0 + w)
                if a < b:
                    got[:, a - c0:b - c0] = e.gram_block(r0, a - s0, 290, b - a)
            assert np.array_equal(got, want), (r0, c0)
            assert int(want.sum()) > 0
        _, rs, nz_full, _ = full.center(want_matrix=False)
        assert np.array_equal(np.concatenate([e.strip_col_sums() for e in owners]), rs)
        comps_full, lam_full, nz = full.compute(2)
        comps, lam, nz2 = strips.compute_pca_over_strips(owners, 2)
        assert nz2 == nz == nz_full
        assert np.max(np.abs(lam - lam_full) / np.abs(lam_full)) < 1e-10
        assert np.abs(align_sign(comps, comps_full) - comps_full).max() < 1e-8
    finally:
        full.close()
        for e in owners:
            e.close()
