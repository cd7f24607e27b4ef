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


def _synthetic_halves(P, n, v, seed=411):
    synth = load_pkg("synth")
    offs = synth.pop_offsets(n)
    thr = synth.thresholds(seed, 0, v)
    return synth, offs, thr


def test_a_reduced_matrix_keeps_the_int32_forms_and_the_upper_triangle_matvec_at_large_n(P):
    """VERDICT r05 Weak 3: after pcoa_gram_reduce_from (two engines on the one GPU), after export -> import (the torch
    all-reduce path) and after a checkpoint load, S used to live in the int64 matrix and N >= 16,384 silently fell back to the
    row form of the mat-vec (8 N^2 bytes per product instead of 2 N^2).  Now: the peer reduction adds the int32 partials in place
    (4 N^2 bytes cross), an int64 matrix that fits int32 is moved back (narrow_s64), and computePca runs the upper-triangle
    forms -- with the SAME eigenpairs as the single engine that saw all variants (reduceByKey, VariantsPca.scala:190, then
    :224-227)."""
    import torch
    n, v, seed = 16384 + 4, 6000, 411
    synth, offs, thr = _synthetic_halves(P, n, v, seed)
    half = 2944   # not a multiple of 128: the halves end in part-filled operand blocks
    with P.PcoaEngine(n) as whole, P.PcoaEngine(n) as a, P.PcoaEngine(n) as b:
        whole.accumulate_synthetic(seed, offs, thr, 0)
        comps0, lam0, nz0 = whole.compute(2)
        t0 = whole.timings()
        assert t0["matvec_form"] == 1 and t0["gram_i64_live"] == 0 and t0["eig_method"] == 1
        a.accumulate_synthetic(seed, offs, thr[:half], 0)
        b.accumulate_synthetic(seed, offs, thr[half:], half)
        a.reduce_from(b)
        ta = a.timings()
        assert ta["reduce_int32_calls"] == 1 and ta["gram_i64_live"] == 0
        for (r0, c0) in ((0, 0), (16000, 100), (100, 16000), (n - 200, n - 200)):
            assert np.array_equal(a.gram_block(r0, c0, 200, 200), whole.gram_block(r0, c0, 200, 200)), (r0, c0)
        comps1, lam1, nz1 = a.compute(2)
        ta = a.timings()
        assert ta["matvec_form"] == 1 and ta["eig_method"] == 1
        assert nz1 == nz0 and np.max(np.abs(lam1 - lam0) / np.abs(lam0)) < 1e-13
        assert np.max(np.abs(comps1 - comps0)) < 1e-13
        # the torch all-reduce path's export -> import on b (here: b <- a's total): the int64 hand-over fits int32 again
        scratch = torch.empty((n, n), dtype=torch.int64, device="cuda:0")
        a.export_device(scratch.data_ptr())
        a.sync()
        b.import_device(scratch.data_ptr())
        b.sync()
        tb = b.timings()
        assert tb["narrowed_to_int32"] == 1 and tb["gram_i64_live"] == 0
        comps2, lam2, nz2 = b.compute(2)
        tb = b.timings()
        assert tb["matvec_form"] == 1
        assert np.max(np.abs(lam2 - lam0) / np.abs(lam0)) < 1e-13 and np.max(np.abs(comps2 - comps0)) < 1e-13
        del scratch
        # accumulation goes on after a narrowing: the books carry max |entry| as the bound of every int32 partial
        b.accumulate_synthetic(seed, offs, thr[:half], 0)
        a.accumulate_synthetic(seed, offs, thr[:half], 0)
        assert np.array_equal(a.gram_block(5, 9000, 300, 300), b.gram_block(5, 9000, 300, 300))


def test_the_int64_reduction_and_kernels_stay_reachable(tmp_path):
    """PCOA_NO_NARROW=1 (r05 behaviour): the peer reduction widens to int64 and computePca reads the int64 matrix -- the path a
    cohort beyond 2^31 variants takes.  Same S, same eigenpairs as the int32 path."""
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import importlib
P = importlib.import_module("spark-examples_amd")
synth = importlib.import_module("spark-examples_amd.synth")
n, v, seed = 300, 3000, 5
offs = synth.pop_offsets(n); thr = synth.thresholds(seed, 0, v)
with P.PcoaEngine(n) as w, P.PcoaEngine(n) as a, P.PcoaEngine(n) as b:
    w.accumulate_synthetic(seed, offs, thr, 0)
    a.accumulate_synthetic(seed, offs, thr[:1000], 0)
    b.accumulate_synthetic(seed, offs, thr[1000:], 1000)
    a.reduce_from(b)
    t = a.timings()
    c1, l1, _ = a.compute(2)
    c0, l0, _ = w.compute(2)
    np.savez(sys.argv[1], same=np.array_equal(a.gram(), w.gram()), i64=t["gram_i64_live"], r32=t["reduce_int32_calls"],
             dl=np.max(np.abs(l1 - l0) / np.abs(l0)), dc=np.max(np.abs(c1 - c0)))
""" % ROOT
    for tag, env, want64 in (("wide", {"PCOA_NO_NARROW": "1"}, 1), ("narrow", {}, 0)):
        out = str(tmp_path / (tag + ".npz"))
        subprocess.check_call([os.sys.executable, "-c", code, out], env=dict(os.environ, **env))
        r = np.load(out)
        assert bool(r["same"]) and int(r["i64"]) == want64 and int(r["r32"]) == 1 - want64
        assert float(r["dl"]) < 1e-12 and float(r["dc"]) < 1e-10


@pytest.mark.parametrize("name", ["tile260", "pops40"])
def test_python_host_with_gpus_2_shards_reduces_and_prints_on_rank_0(name, tmp_path):
    """The Python twin's --gpus K end to end with two REAL ranks (VERDICT r05 Missing 4): `variants_pca.py --gpus 2` re-executes
    itself under torch.distributed.run, rank r accumulates shard_range(r, 2, rows) of the RDD[Seq[Int]] rows on its engine, the
    partial matrices are summed (reduceByKey, VariantsPca.scala:190), rank 0 runs computePca and prints.  One GPU here, so both
    ranks share cuda:0 and the wire is gloo (--rank-devices 0,0 --dist-backend gloo; on a node: RCCL, one GPU per rank).
    S must equal the reference's own similarity matrix and the output the single-process run's."""
    import socket
    g = load_golden(name)
    n = int(g["n_samples"])
    path = str(tmp_path / "golden.vcf")
    write_golden_vcf(g, path)
    script = os.path.join(ROOT, "spark-examples_amd", "variants_pca.py")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_PORT=str(port))
    outs = {}
    for tag, extra in (("one", []), ("two", ["--gpus", "2", "--rank-devices", "0,0", "--dist-backend", "gloo"])):
        dump = str(tmp_path / (tag + ".bin"))
        res = subprocess.run([os.sys.executable, script, "--input-path", path, "--all-references", "--dump-similarity", dump] + extra,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, env=env, timeout=600)
        assert res.returncode == 0, res.stderr[-3000:]
        assert np.array_equal(np.fromfile(dump, dtype="<i8").reshape(n, n), g["similarity"]), tag
        outs[tag] = [l for l in res.stdout.splitlines() if "\t" in l or l.startswith(("Matrix size", "Non zero rows"))]
        if tag == "two":
            assert "Reduced over 2 ranks" in res.stderr, res.stderr[-2000:]
    assert len(outs["one"]) == n + 2 and [l.split("\t")[:2] for l in outs["one"]] == [l.split("\t")[:2] for l in outs["two"]]
    a = np.array([[float(x) for x in l.split("\t")[2:4]] for l in outs["one"] if "\t" in l])
    b = np.array([[float(x) for x in l.split("\t")[2:4]] for l in outs["two"] if "\t" in l])
    assert np.abs(a - b).max() < 1e-12
