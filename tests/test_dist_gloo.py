"""World-size-2 test of the multi-GPU path on CPU (gloo): variant sharding + all-reduce of the
partial Gram gives exactly the single-shard result.  The per-rank partial comes from the oracle
here (no GPU in this container); on the GPU box the same dist.* functions carry the HIP partials."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist_t
import torch.multiprocessing as mp

from conftest import ROOT, load_oracle, load_pkg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, seed, v, n, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist_t.init_process_group("gloo", rank=rank, world_size=world)
    oracle = load_oracle()
    oracle.set_num_threads(1)
    dist = load_pkg("dist")
    synth = load_pkg("synth")
    offsets = synth.pop_offsets(n)
    v0, v1 = dist.shard_range(rank, world, v)
    thr = synth.thresholds(seed, v0, v1 - v0)
    x = synth.genotypes(seed, v0, thr, offsets)           # counter-based: shard-invariant bits
    s_local = oracle.similarity_from_dense(x, n) if v1 > v0 else np.zeros((n, n), dtype=np.int64)
    s = dist.allreduce_gram_numpy(s_local)
    np.save(os.path.join(out_dir, "s_rank%d.npy" % rank), s)
    dist_t.barrier()
    dist_t.destroy_process_group()


def test_two_rank_allreduce_matches_single_shard(tmp_path):
    seed, v, n, world = 77, 301, 45, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, seed, v, n, str(tmp_path)), nprocs=world, join=True)
    oracle = load_oracle()
    synth = load_pkg("synth")
    x = synth.genotypes(seed, 0, synth.thresholds(seed, 0, v), synth.pop_offsets(n))
    full = oracle.similarity_from_dense(x, n)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "s_rank%d.npy" % r))
        assert np.array_equal(got, full)


def test_single_process_allreduce_is_identity():
    dist = load_pkg("dist")
    s = np.arange(9, dtype=np.int64).reshape(3, 3)
    assert np.array_equal(dist.allreduce_gram_numpy(s), s)
    t = torch.ones(2, 2, dtype=torch.int64)
    assert dist.allreduce_gram_tensor(t) is t


def test_gpus_n_without_a_rendezvous_spawns_n_ranks_or_fails_loudly():
    """bench.py --gpus N started the way the driver starts --gpus 1 must not run one rank and print n_gpus = 1."""
    dist = load_pkg("dist")
    assert dist.launch_plan(1, {}, 1, ["bench.py"]) == ("run", None)
    assert dist.launch_plan(8, {"WORLD_SIZE": "8", "RANK": "3"}, 8, ["bench.py"]) == ("run", None)
    what, cmd = dist.launch_plan(8, {}, 8, ["bench.py", "--gpus", "8", "--steps", "5"], python="py")
    assert what == "spawn" and cmd[:3] == ["py", "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-5:] == ["bench.py", "--gpus", "8", "--steps", "5"]
    assert dist.launch_plan(8, {}, 1, ["bench.py"])[0] == "error"          # fewer devices than ranks
    assert dist.launch_plan(8, {"WORLD_SIZE": "2"}, 8, ["bench.py"])[0] == "error"   # the launcher disagrees with --gpus
    assert dist.launch_plan(0, {}, 8, ["bench.py"])[0] == "error"


def _tele_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist_t.init_process_group("gloo", rank=rank, world_size=world)
    dist = load_pkg("dist")
    tele = dist.collective_telemetry(1.0 + rank, 0.01 * (rank + 1), {"allreduce_seconds": 0.004, "allreduce_calls": 1,
                                                                    "comm_ranks": world, "allreduce_int32": 1})
    import json
    with open(os.path.join(out_dir, "tele%d.json" % rank), "w") as f:
        json.dump(tele, f)
    dist_t.barrier()
    dist_t.destroy_process_group()


def test_multi_rank_record_explains_itself(tmp_path):
    """the fields bench.py prints for N > 1 -- per-rank elapsed (min / max), the reduction step's wall
    (max over ranks) and HIP-event time, the communicator's own rank count -- gathered over a world-size-2 gloo group."""
    import json
    world = 2
    mp.spawn(_tele_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        t = json.load(open(os.path.join(str(tmp_path), "tele%d.json" % r)))
        assert t["world_size"] == 2 and t["rank"] == r and t["rccl_ranks"] == 2
        assert t["rank_elapsed_s"] == [1.0, 2.0] and t["rank_elapsed_min_s"] == 1.0 and t["rank_elapsed_max_s"] == 2.0
        assert abs(t["allreduce_ms"] - 20.0) < 1e-9 and abs(t["allreduce_event_ms"] - 4.0) < 1e-9
        assert t["allreduce_int32_in_place"] is True
    dist = load_pkg("dist")
    solo = dist.collective_telemetry(0.5, 0.0, {})
    assert solo["world_size"] == 1 and solo["rccl_ranks"] is None and solo["allreduce_event_ms"] is None
    # bench.py's line carries them
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ("rccl_ranks", "allreduce_ms", "allreduce_event_ms", "rank_elapsed_min_s", "rank_elapsed_max_s", "n1_equivalent_value"):
        assert '"%s"' % key in src, key


def test_python_host_gpus_flag_plans_one_rank_per_gpu_and_shards_like_the_reference_partitions():
    """The Python twin's --gpus K (VERDICT r05 Missing 4): without a rendezvous it re-executes itself under
    torch.distributed.run with K ranks (or fails loudly), and a rank's partition of every RDD[Seq[Int]] form is the contiguous
    range dist.shard_range gives -- the shards concatenate to the input (VariantsPca.scala:184, GenomicsConf.scala:42-45)."""
    vp = load_pkg("variants_pca")
    conf = vp.PcaConf(["--gpus", "2", "--synthetic", "10,8,1"])
    assert conf.gpus == 2 and conf.allreduce == "native"
    what, cmd = vp.multi_gpu_plan(conf, ["--gpus", "2", "--synthetic", "10,8,1"], {}, 2, python="py")
    assert what == "spawn" and cmd[:3] == ["py", "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5].endswith(os.path.join("spark-examples_amd", "variants_pca.py")) and cmd[-4:] == ["--gpus", "2", "--synthetic", "10,8,1"]
    assert vp.multi_gpu_plan(conf, [], {}, 1)[0] == "error"                      # fewer devices than ranks
    assert vp.multi_gpu_plan(conf, [], {"WORLD_SIZE": "2", "RANK": "1"}, 2) == ("run", None)
    assert vp.multi_gpu_plan(conf, [], {"WORLD_SIZE": "4"}, 4)[0] == "error"     # the launcher disagrees with --gpus
    assert vp.PcaConf(["--synthetic", "10,8,1"]).gpus == 1
    rng = np.random.default_rng(3)
    lists = [sorted(rng.choice(9, size=rng.integers(0, 5), replace=False).tolist()) for _ in range(11)]
    offs = np.concatenate([[0], np.cumsum([len(c) for c in lists])]).astype(np.int64)
    idx = np.array([i for c in lists for i in c], dtype=np.int32)
    bits = rng.integers(0, 2 ** 32, size=(11, 1), dtype=np.uint64).astype(np.uint32)
    geno = rng.integers(0, 256, size=(11, 3), dtype=np.uint8)
    keep = rng.random(11) < 0.7
    for world in (1, 2, 3, 4):
        parts = [vp.shard_calls(lists, r, world) for r in range(world)]
        assert [c for p in parts for c in p] == lists
        got = [vp.shard_calls((idx, offs), r, world) for r in range(world)]
        assert np.array_equal(np.concatenate([g[0] for g in got]), idx)
        assert all(g[1][0] == 0 and g[1][-1] == g[0].size for g in got) and sum(g[1].size - 1 for g in got) == 11
        gb = [vp.shard_calls(("bits", bits), r, world) for r in range(world)]
        assert all(g[0] == "bits" for g in gb) and np.array_equal(np.concatenate([g[1] for g in gb]), bits)
        gd = [vp.shard_calls(("bed", geno, keep, True), r, world) for r in range(world)]
        assert np.array_equal(np.concatenate([g[1] for g in gd]), geno) and np.array_equal(np.concatenate([g[2] for g in gd]), keep)
        assert all(g[3] is True for g in gd)
