"""Multi-GPU sharding of the Gram accumulation: one process per GPU, variants partitioned, one
all-reduce of the partial N x N matrix (replaces reduceByKey(_ + _), VariantsPca.scala:190).

S = sum_v x_v x_v^T is a sum over variants, so rank r takes the contiguous variant range
shard_range(r, world, V) exactly as the reference's partitions do (VariantsPca.scala:184-190);
integer partials make the result independent of the number of ranks and of the reduction order.
The eigendecomposition runs on rank 0 (every rank holds the reduced S, so any rank could).

torch.distributed is used for rendezvous and the collective only (backend "nccl" is RCCL on ROCm;
"gloo" in the CPU tests).  The C ABI also offers a torch-free path: pcoa_comm_* +
pcoa_gram_allreduce_rccl.
"""
import numpy as np


def shard_range(rank, world_size, n_variants):
    """Contiguous, balanced partition of [0, n_variants): the first (V mod W) ranks get one extra."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside [0, %d)" % (rank, world_size))
    base, extra = divmod(int(n_variants), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def launch_plan(gpus, env, device_count, argv, python="python"):
    """What a script asked for `--gpus N` has to do (bench.py, tools/config5_strips.py).  Returns
      ("run", None)        this process IS a rank (WORLD_SIZE set, e.g. under torch.distributed.run) or N == 1
      ("spawn", command)   N > 1 and no rendezvous in the environment: re-execute under torch.distributed.run with one
                           rank per GPU (rendezvous on 127.0.0.1: the container's host name may not resolve)
      ("error", message)   the request cannot be honoured: fewer than N devices, or WORLD_SIZE disagrees with N.
    A `--gpus 8` run that silently used one rank would report n_gpus = 1 (Missing 2)."""
    gpus = int(gpus)
    if gpus < 1:
        return "error", "--gpus must be >= 1"
    world = env.get("WORLD_SIZE")
    if world is not None:
        if int(world) != gpus:
            return "error", "--gpus %d but WORLD_SIZE=%s: launch one rank per GPU" % (gpus, world)
        return "run", None
    if gpus == 1:
        return "run", None
    if device_count < gpus:
        return "error", "--gpus %d but only %d GPU(s) are visible" % (gpus, device_count)
    port = env.get("MASTER_PORT", "29531")
    cmd = [python, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + list(argv)
    return "spawn", cmd


def allreduce_gram_tensor(t, group=None):
    """In-place sum of an int64 [N][N] tensor over the process group (CPU/gloo or GPU/RCCL)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def allreduce_engine(engine, group=None, scratch=None):
    """All-reduce the finalized S of a PcoaEngine across ranks through a torch int64 device tensor
    (export -> all_reduce over RCCL/xGMI -> import).  Returns the scratch tensor for reuse."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        engine.finalize()
        return scratch
    if scratch is None:
        scratch = torch.empty((engine.n, engine.n), dtype=torch.int64, device="cuda:%d" % engine.device)
    engine.export_device(scratch.data_ptr())
    engine.sync()  # the export ran on the engine's stream; the collective runs on torch's
    if dist.get_backend(group) == "gloo":   # CPU wire (one-GPU test boxes: several ranks share a device, which RCCL refuses)
        host = scratch.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        scratch.copy_(host)
    else:
        dist.all_reduce(scratch, op=dist.ReduceOp.SUM, group=group)
    torch.cuda.current_stream(scratch.device).synchronize()
    engine.import_device(scratch.data_ptr())
    engine.sync()
    return scratch


class NativeComm(object):
    """RCCL communicator owned by the C ABI (pcoa_comm_init): the all-reduce then runs on the engine's
    own stream, in place on the int32 partial (25 MB at N = 2504) whenever the summed counts fit int32,
    with no export/import copies.  The 128-byte unique id travels over torch.distributed once."""

    def __init__(self, engine, group=None):
        import torch
        import torch.distributed as dist
        self.engine = engine
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        dev = torch.device("cuda", engine.device)
        if self.rank == 0:
            uid = torch.tensor(list(engine.comm_unique_id()), dtype=torch.uint8, device=dev)
        else:
            uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        dist.broadcast(uid, src=0, group=group)
        self.comm = engine.comm_init(bytes(uid.cpu().tolist()), self.rank, self.world)

    def allreduce(self):
        self.engine.allreduce_rccl(self.comm)

    def count(self):
        """ncclCommCount of the communicator (pcoa_comm_count): the rank count RCCL itself reports."""
        return self.engine.comm_count(self.comm)

    def close(self):
        if self.comm is not None:
            self.engine.comm_destroy(self.comm)
            self.comm = None


def collective_telemetry(elapsed_local_s, allreduce_wall_s, timings=None, rccl_ranks=None, group=None, device=None):
    """What a multi-rank record must carry to explain itself: the per-rank elapsed times (all-gathered),
    their min / max, the wall time of the reduction step (max over ranks), the collective's own HIP-event time when the
    library ran it (pcoa_timings.allreduce_seconds), and the rank count the communicator reports.  Works on any backend
    (gloo in the CPU tests); without an initialised process group it describes the single rank."""
    import torch
    import torch.distributed as dist
    world, rank = 1, 0
    per_rank = [float(elapsed_local_s)]
    red = float(allreduce_wall_s)
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        dev = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")
        mine = torch.tensor([float(elapsed_local_s), float(allreduce_wall_s)], dtype=torch.float64, device=dev)
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine, group=group)
        per_rank = [float(g[0]) for g in got]
        red = max(float(g[1]) for g in got)
    t = timings or {}
    return {"world_size": world, "rank": rank,
            "rccl_ranks": int(rccl_ranks) if rccl_ranks is not None else (int(t.get("comm_ranks", 0)) or None),
            "allreduce_ms": 1e3 * red,
            "allreduce_event_ms": 1e3 * float(t.get("allreduce_seconds", 0.0)) if t.get("allreduce_calls", 0) else None,
            "allreduce_int32_in_place": bool(t.get("allreduce_int32", 0)) if t.get("allreduce_calls", 0) else None,
            "rank_elapsed_s": per_rank, "rank_elapsed_min_s": min(per_rank), "rank_elapsed_max_s": max(per_rank)}


def allreduce_gram_numpy(s_local, group=None):
    """CPU twin used by the gloo tests: numpy int64 partial -> summed numpy int64."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(s_local, dtype=np.int64).copy())
    allreduce_gram_tensor(t, group)
    return t.numpy()
