"""CPU tests of the strip-owner computePca (spark-examples_amd/strips.py): the host-driven Lanczos over
column strips of S, with numpy stand-ins for the GPU strip owners (tests/strip_standins.py: the same interface, .lanczos included),
single-process and over two gloo ranks."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, align_sign, load_oracle, load_pkg, planted_callsets
from strip_standins import HostStrip


def _cohort(seed, n, v):
    rng = np.random.default_rng(seed)
    return planted_callsets(rng, n, v)


def test_strip_ranges_tile_the_samples():
    strips = load_pkg("strips")
    for n, g in ((2504, 8), (250000, 8), (6000, 3), (10, 10), (7, 2), (300, 4)):
        r = strips.strip_ranges(n, g)
        assert len(r) == g and r[0][0] == 0 and sum(w for _, w in r) == n
        assert all(r[i][0] + r[i][1] == r[i + 1][0] for i in range(g - 1)) and all(w > 0 for _, w in r)
    assert all(c % 256 == 0 for c, _ in strips.strip_ranges(250000, 8))   # cut at tile edges when there is room
    with pytest.raises(ValueError):
        strips.strip_ranges(3, 5)


@pytest.mark.parametrize("ranges", [[(0, 300)], [(0, 100), (100, 57), (157, 143)], [(0, 1), (1, 298), (299, 1)]])
def test_lanczos_over_strips_matches_the_oracle(ranges):
    strips = load_pkg("strips")
    oracle = load_oracle()
    x = _cohort(11, 300, 1500)
    s = oracle.similarity_from_dense(x, 300)
    ref = oracle.compute_pca(s, 3)
    owners = [HostStrip(s, c0, w) for c0, w in ranges]
    trace = []
    comps, lam, nz = strips.compute_pca_over_strips(owners, 3, trace=trace)
    assert nz == ref["nonzero_rows"] and len(trace) >= 1
    assert np.max(np.abs(lam - ref["eigenvalues"]) / np.abs(ref["eigenvalues"])) < 1e-9
    assert np.abs(align_sign(comps, ref["components"]) - ref["components"]).max() < 1e-8
    assert np.allclose(np.linalg.norm(comps, axis=0), 1.0, atol=1e-12)
    # sign convention: the largest-magnitude entry of every component is positive
    assert all(comps[np.argmax(np.abs(comps[:, c])), c] > 0 for c in range(3))
    with pytest.raises(ValueError):
        strips.compute_pca_over_strips(owners[:1] if len(owners) > 1 else [HostStrip(s, 0, 299)], 2)   # strips do not tile N
    with pytest.raises(ValueError):
        strips.compute_pca_over_strips(owners, 0)


def test_no_verified_pair_is_an_error_not_a_guess():
    """identical samples: B = 0, every Ritz value is 0 with no gap -- nothing can be verified and nothing is returned"""
    strips = load_pkg("strips")
    s = np.full((40, 40), 7, dtype=np.int64)
    with pytest.raises(RuntimeError):
        strips.compute_pca_over_strips([HostStrip(s, 0, 40)], 2)


def _free_port():
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    p = sk.getsockname()[1]
    sk.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    strips = load_pkg("strips")
    oracle = load_oracle()
    oracle.set_num_threads(1)
    x = _cohort(23, 257, 1200)
    s = oracle.similarity_from_dense(x, 257)
    ranges = strips.strip_ranges(257, 3, align=64)
    mine = ranges[:2] if rank == 0 else ranges[2:]            # rank 0 owns two strips, rank 1 one: ragged all-gather
    owners = [HostStrip(s, c0, w) for c0, w in mine]
    comps, lam, nz = strips.compute_pca_over_strips(owners, 2)
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), comps=comps, lam=lam, nz=nz)
    td.barrier()
    td.destroy_process_group()


def _feed_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    strips = load_pkg("strips")
    ingest = load_pkg("ingest")
    n = 200
    x = _cohort(29, n, 1200)
    shard = x[:700] if rank == 0 else x[700:]                 # ragged variant shards: 700 and 500 rows
    ranges = strips.strip_ranges(n, 3, align=64)
    mine = ranges[:1] if rank == 0 else ranges[1:]
    owners = [HostStrip.empty(n, c0, w) for c0, w in mine]
    fed = strips.feed_owners_from_variant_shards(owners, ingest.pack_bits(shard), chunk_variants=256)   # 3 rounds
    comps, lam, nz = strips.compute_pca_over_strips(owners, 2)
    np.savez(os.path.join(out_dir, "f%d.npz" % rank), comps=comps, lam=lam, nz=nz, fed=fed,
             s=np.concatenate([o.s for o in owners], axis=1), c0=mine[0][0])
    td.barrier()
    td.destroy_process_group()


def test_variant_shards_are_all_gathered_to_every_strip_owner(tmp_path):
    """SURVEY 8e, the C5 layout end to end on two gloo ranks: variants sharded over the ranks (ragged), columns of S
    tiled over the ranks (rank 0 one strip, rank 1 two), bitsets exchanged in three rounds, then the Lanczos over strips."""
    import torch.multiprocessing as mp
    strips = load_pkg("strips")
    ingest = load_pkg("ingest")
    port = _free_port()
    mp.spawn(_feed_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    oracle = load_oracle()
    x = _cohort(29, 200, 1200)
    s = oracle.similarity_from_dense(x, 200)
    ref = oracle.compute_pca(s, 2)
    got = [np.load(os.path.join(str(tmp_path), "f%d.npz" % r)) for r in range(2)]
    assert int(got[0]["fed"]) == int(got[1]["fed"]) == 1200
    tiled = np.concatenate([got[0]["s"], got[1]["s"]], axis=1)
    assert np.array_equal(tiled, s.astype(np.float64))                         # every owner saw every variant once
    assert np.array_equal(got[0]["comps"], got[1]["comps"])
    assert np.max(np.abs(got[0]["lam"] - ref["eigenvalues"]) / np.abs(ref["eigenvalues"])) < 1e-9
    assert np.abs(align_sign(got[0]["comps"], ref["components"]) - ref["components"]).max() < 1e-8
    # single process: the same call is a chunked feed
    o = HostStrip.empty(200, 0, 200)
    assert strips.feed_owners_from_variant_shards([o], ingest.pack_bits(x), chunk_variants=500) == 1200
    assert np.array_equal(o.s, s.astype(np.float64))
    with pytest.raises(ValueError):
        strips.feed_owners_from_variant_shards([o], ingest.pack_bits(x).astype(np.int64))


def test_two_ranks_all_gather_their_strips(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    oracle = load_oracle()
    x = _cohort(23, 257, 1200)
    ref = oracle.compute_pca(oracle.similarity_from_dense(x, 257), 2)
    got = [np.load(os.path.join(str(tmp_path), "r%d.npz" % r)) for r in range(2)]
    assert np.array_equal(got[0]["comps"], got[1]["comps"]) and np.array_equal(got[0]["lam"], got[1]["lam"])   # replicated
    assert int(got[0]["nz"]) == ref["nonzero_rows"]
    assert np.max(np.abs(got[0]["lam"] - ref["eigenvalues"]) / np.abs(ref["eigenvalues"])) < 1e-9
    assert np.abs(align_sign(got[0]["comps"], ref["components"]) - ref["components"]).max() < 1e-8


def _config5_worker(rank, world, port, out_dir, exchange):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    import importlib
    tool = importlib.import_module("config5_strips")
    out = tool.main(["--standin", "--samples", "260", "--variants", "1500", "--chunk", "400", "--exchange", exchange])
    if rank == 0:
        import json
        json.dump(out, open(os.path.join(out_dir, "c5_%s.json" % exchange), "w"))


@pytest.mark.parametrize("exchange", ["bits", "none"])
def test_config5_driver_control_flow_on_two_gloo_ranks(tmp_path, exchange):
    """tools/config5_strips.py (one strip owner per rank, the launcher of the configs[4] layout) with numpy stand-in
    owners: variants sharded and exchanged as bitsets / regenerated by every owner, Lanczos over the strips; the result
    is held to the oracle on the whole cohort."""
    import json
    import torch.multiprocessing as mp
    synth = load_pkg("synth")
    port = _free_port()
    mp.spawn(_config5_worker, args=(2, port, str(tmp_path), exchange), nprocs=2, join=True)
    out = json.load(open(os.path.join(str(tmp_path), "c5_%s.json" % exchange)))
    oracle = load_oracle()
    n, v = 260, 1500
    x = synth.genotypes(1005, 0, synth.thresholds(1005, 0, v), synth.pop_offsets(n), dtype=np.uint8)
    ref = oracle.compute_pca(oracle.similarity_from_dense(x, n), 2)
    assert out["variants_fed_to_every_owner"] == v and out["nonzero_rows"] == ref["nonzero_rows"]
    assert np.max(np.abs(np.array(out["eigenvalues"]) - ref["eigenvalues"]) / np.abs(ref["eigenvalues"])) < 1e-9
    assert max(out["relative_residuals"]) < 1e-8
