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
