"""GPU tests of r04's host-side work: several engines in one process (--gpus k of the compiled
host: k threads, contiguous variant ranges, RCCL or peer reduction -- tested here with every engine on the one GPU of the
box, which the peer reduction allows), the streaming PLINK reader, and the device-side .bed decode.  Everything is held to the
matrices the reference's own Python produced (tests/golden)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden_cases, load_golden, load_pkg, write_golden_plink, write_golden_vcf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    return load_pkg()


def _exe():
    exe = os.path.join(ROOT, "spark-examples_amd", "variants_pca_driver")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "spark-examples_amd", "host")])
    return exe


def _similarity(args, n, tmp_path, tag):
    dump = str(tmp_path / ("s_%s.bin" % tag))
    res = subprocess.run([_exe()] + args + ["--dump-similarity", dump], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         universal_newlines=True)
    assert res.returncode == 0, res.stderr
    return np.fromfile(dump, dtype="<i8").reshape(n, n), res


@pytest.mark.parametrize("name", golden_cases())
def test_goldens_as_plink_streamed_through_one_two_and_three_engines_give_the_reference_matrix(name, tmp_path):
    g = load_golden(name)
    n = int(g["n_samples"])
    for flip in (False, True):
        prefix = str(tmp_path / ("flip" if flip else "plain"))
        write_golden_plink(g, prefix, flip=flip)
        base = ["--input-path", prefix + ".bed"] + (["--plink-ref-allele", "a1"] if flip else [])
        runs = {
            "stream_device_decode": [],
            "two_engines_small_blocks": ["--gpus", "2", "--gpu-map", "0,0", "--stream-rows", "7"],
            "three_engines_host_decode": ["--gpus", "3", "--gpu-map", "0,0,0", "--stream-rows", "5", "--plink-decode", "host"],
            "in_memory": ["--no-stream"],
            "in_memory_two_engines": ["--no-stream", "--gpus", "2", "--gpu-map", "0,0", "--reduce", "peer"],
        }
        for tag, extra in runs.items():
            s, res = _similarity(base + extra, n, tmp_path, tag)
            assert np.array_equal(s, g["similarity"]), (tag, flip)
            if "engines" in tag:
                assert "peer reduction" in res.stderr, res.stderr
            if tag.startswith("stream") or "small_blocks" in tag or "host_decode" in tag:
                assert "Streamed" in res.stderr and "peak RSS" in res.stderr


@pytest.mark.parametrize("name", ["tile260", "pops40"])
def test_goldens_as_vcf_through_two_engines_give_the_reference_matrix_and_the_same_output(name, tmp_path):
    g = load_golden(name)
    n = int(g["n_samples"])
    path = str(tmp_path / "golden.vcf")
    write_golden_vcf(g, path)
    s1, r1 = _similarity(["--input-path", path], n, tmp_path, "one")
    s2, r2 = _similarity(["--input-path", path, "--gpus", "2", "--gpu-map", "0,0"], n, tmp_path, "two")
    s3, r3 = _similarity(["--input-path", path, "--no-stream"], n, tmp_path, "mem")
    assert np.array_equal(s1, g["similarity"]) and np.array_equal(s2, s1) and np.array_equal(s3, s1)
    assert r1.stdout == r2.stdout == r3.stdout   # same S, same engine for computePca: the printed coordinates are identical
    # r05: one engine streams a single VCF block by block; --gpus k and --no-stream hold the data set first
    assert "streamed block by block" in r1.stderr and "streamed block by block" not in r2.stderr + r3.stderr


def test_more_engines_than_devices_fails_loudly(tmp_path):
    """--gpus 2 with the default map (devices 0 and 1) on a one-GPU box must not run one engine and call it two."""
    import torch
    if torch.cuda.device_count() > 1:
        pytest.skip("this box has a second GPU")
    g = load_golden("kat5")
    prefix = str(tmp_path / "p")
    write_golden_plink(g, prefix)
    res = subprocess.run([_exe(), "--input-path", prefix + ".bed", "--gpus", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         universal_newlines=True)
    assert res.returncode != 0 and "pcoa_create on device 1" in res.stderr


def test_peer_reduction_and_device_side_bed_decode_through_the_c_abi(P):
    """pcoa_gram_reduce_from: S of two engines fed disjoint variant ranges, reduced, equals the single engine's; and
    pcoa_accumulate_plink_bed (raw 2-bit rows, host and device pointers, both reference-allele conventions) equals the bitsets
    the Python ingest builds from the same bytes."""
    import torch
    rng = np.random.default_rng(5)
    n, v = 301, 1500                       # n % 4 == 1, n % 32 != 0: ragged last byte and last word
    bpv = (n + 3) // 4
    raw = rng.integers(0, 256, size=(v, bpv), dtype=np.uint8)
    codes = np.stack([(raw >> (2 * q)) & 3 for q in range(4)], axis=2).reshape(v, bpv * 4)[:, :n]
    for ref_a1 in (False, True):
        carrier = (codes == 2) | (codes == (3 if ref_a1 else 0))
        want = carrier.T.astype(np.int64) @ carrier.astype(np.int64)
        with P.PcoaEngine(n) as a, P.PcoaEngine(n) as b, P.PcoaEngine(n) as c:
            a.accumulate_plink_bed(raw[:700], ref_is_a1=ref_a1)
            b.accumulate_plink_bed(torch.from_numpy(raw[700:]).cuda(), ref_is_a1=ref_a1)
            c.accumulate_plink_bed(raw, ref_is_a1=ref_a1)
            assert np.array_equal(c.gram(), want)
            a.reduce_from(b)
            assert np.array_equal(a.gram(), want)
            assert np.array_equal(b.gram(), carrier[700:].T.astype(np.int64) @ carrier[700:].astype(np.int64))   # src unchanged
            a.accumulate_plink_bed(raw[:10], ref_is_a1=ref_a1)      # the reduced engine keeps accumulating
            assert np.array_equal(a.gram(), want + carrier[:10].T.astype(np.int64) @ carrier[:10].astype(np.int64))
        with P.PcoaEngine(n) as a, P.PcoaEngine(n + 1) as b:
            with pytest.raises(P.PcoaError):
                a.reduce_from(b)
            with pytest.raises(P.PcoaError):
                a.reduce_from(a)


def test_queued_bed_blocks_from_three_rotating_page_locked_buffers(P):
    """PCOA_BED_HOST_ASYNC (r05, the streaming host's feed): the call only queues a page-locked block; three buffers rotate and
    each is REWRITTEN as soon as the rule of pcoa.h allows (after the second later call has returned) -- so a copy that had
    not finished by then would contract the wrong bytes.  Blocks of different sizes, one of them larger than the 2^17-row
    chunk of the device slots; the result must equal the bitset path on the same bytes."""
    import torch
    rng = np.random.default_rng(9)
    n = 2504
    bpv = (n + 3) // 4
    sizes = [5000, 70000, 131072 + 777, 1, 4096, 65536, 33333, 9]
    blocks = [rng.integers(0, 256, size=(v, bpv), dtype=np.uint8) | np.uint8(0xAA) for v in sizes]   # codes 10 / 11: sparse carriers
    pins = [torch.empty((max(sizes), bpv), dtype=torch.uint8).pin_memory() for _ in range(3)]
    with P.PcoaEngine(n) as ref, P.PcoaEngine(n) as eng:
        for i, blk in enumerate(blocks):
            ref.accumulate_plink_bed(blk)
            buf = pins[i % 3]
            buf[:blk.shape[0]].copy_(torch.from_numpy(blk))          # overwrites the block handed over three calls ago
            eng.accumulate_plink_bed(buf[:blk.shape[0]], asynchronous=True)
        eng.sync()
        for b in pins:
            b.fill_(0xFF)                                            # after a synchronising call the buffers are the caller's
        assert np.array_equal(eng.gram(), ref.gram())


def test_streamed_plink_with_a_references_filter_that_drops_whole_blocks_equals_the_in_memory_path(tmp_path):
    """The queued feed (r05: four rotating page-locked blocks, a reader thread) when --references keeps two disjoint ranges:
    blocks in front, between and behind them hold no kept row at all and are never handed over -- the rule that frees a
    block counts CALLS, not blocks -- and blocks at the range ends are squeezed.  Same S and output as --no-stream."""
    g = load_golden("pops40")
    n = int(g["n_samples"])
    prefix = str(tmp_path / "p")
    write_golden_plink(g, prefix)
    pos = lambda k: 41196312 + 7 * k                                  # noqa: E731  (the .bim of write_golden_plink)
    refs = "17:%d:%d,17:%d:%d" % (pos(20) - 1, pos(61) - 1, pos(110) - 1, pos(150) - 1)
    base = ["--input-path", prefix + ".bed", "--references", refs]
    s_mem, r_mem = _similarity(base + ["--no-stream"], n, tmp_path, "mem")
    assert 0 < int(np.trace(s_mem)) < int(np.trace(g["similarity"]))   # a proper subset of the variants
    for tag, extra in (("rows3", ["--stream-rows", "3"]), ("rows16", ["--stream-rows", "16"]),
                       ("rows3_two_engines", ["--stream-rows", "3", "--gpus", "2", "--gpu-map", "0,0"])):
        s, r = _similarity(base + extra, n, tmp_path, tag)
        assert np.array_equal(s, s_mem), tag
        assert r.stdout == r_mem.stdout, tag
