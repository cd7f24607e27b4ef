#!/usr/bin/env python3
"""BASELINE configs[4] layout ("Gram tiled across HBM"): one strip owner per rank, every rank sees every variant,
computePca as a Lanczos iteration over the strips (spark-examples_amd/strips.py; DESIGN_HISTORY.md 4.5).

  torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/config5_strips.py --samples 250000 --variants 10000000

Every rank owns the column strip strips.strip_ranges(N, world)[rank] of S and generates its shard of the VARIANTS
(dist.shard_range) on its device; --exchange bits turns the shard into carrier bitsets and all-gathers them to every
owner (the path real data takes: 31 KB per variant at N = 250,000), --exchange none lets every owner regenerate every
variant itself (the synthetic cohort is counter-based, so this needs no communication at all -- the upper bound of what
the exchange can cost).  One JSON line on rank 0.

--standin runs the same control flow on the CPU with numpy strip owners (gloo; what tests/test_strips_cpu.py spawns);
without it the engine is the HIP library and there is no CPU fallback.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def NumpyStrip(n, col0, cols):
    """CPU stand-in for PcoaEngine(strip=...) (tests / --standin only): tests/strip_standins.py"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from strip_standins import HostStrip
    return HostStrip.empty(n, col0, cols)


def device_bitsets(owner, seed, offs, thr, first, n, dev):
    """This rank's variants [first, first + len(thr)) as carrier bitsets on the device: the engine's generator fills an
    fp32 tile (the same Philox stream as synth.genotypes), torch packs it 32 samples to a word (sample i -> bit i & 31
    of word i >> 5, the layout of pcoa_accumulate_bits)."""
    import torch
    cnt = int(thr.shape[0])
    words = (n + 31) // 32
    if cnt == 0:
        return torch.zeros((0, words), dtype=torch.int32, device=dev)
    xf = torch.empty((cnt, n), dtype=torch.float32, device=dev)
    owner.synth_fill(seed, offs, thr, first, xf.data_ptr(), n)
    owner.sync()
    xb = xf > 0
    del xf
    if words * 32 != n:
        xb = torch.nn.functional.pad(xb, (0, words * 32 - n))
    w = (xb.view(cnt, words, 32).to(torch.int64) << torch.arange(32, device=dev, dtype=torch.int64)).sum(dim=2)
    w = torch.where(w >= (1 << 31), w - (1 << 32), w)          # the same 32 bits as a signed word
    return w.to(torch.int32).contiguous()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=250000)
    ap.add_argument("--variants", type=int, default=10000000)
    ap.add_argument("--seed", type=int, default=1005)
    ap.add_argument("--chunk", type=int, default=8192,
                    help="variants per generation / exchange round (the fp32 tile the generator fills is chunk x N x 4 "
                         "bytes: 8 GB at N = 250,000)")
    ap.add_argument("--exchange", choices=("bits", "none"), default="bits")
    ap.add_argument("--num-pc", type=int, default=2)
    ap.add_argument("--standin", action="store_true", help="numpy strip owners on the CPU (gloo); tests only")
    args = ap.parse_args(argv)

    import torch
    import torch.distributed as td
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1 and not td.is_initialized():
        if args.standin:
            td.init_process_group("gloo", rank=rank, world_size=world)
        else:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            torch.cuda.set_device(local_rank)
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    strips = importlib.import_module("spark-examples_amd.strips")
    synth = importlib.import_module("spark-examples_amd.synth")
    dist = importlib.import_module("spark-examples_amd.dist")
    ingest = importlib.import_module("spark-examples_amd.ingest")
    n, v, seed = args.samples, args.variants, args.seed
    offs = synth.pop_offsets(n)
    c0, w = strips.strip_ranges(n, world)[rank]
    if args.standin:
        owner = NumpyStrip(n, c0, w)
    else:
        P = importlib.import_module("spark-examples_amd")
        owner = P.PcoaEngine(n, device=local_rank, strip=(c0, w))

    t0 = time.perf_counter()
    fed = 0
    if args.exchange == "none":
        for v0 in range(0, v, args.chunk):
            cnt = min(args.chunk, v - v0)
            thr = synth.thresholds(seed, v0, cnt)
            if args.standin:
                owner.accumulate_bits(ingest.pack_bits(synth.genotypes(seed, v0, thr, offs, dtype=np.uint8)))
            else:
                owner.accumulate_synthetic(seed, offs, thr, v0)
            fed += cnt
    else:
        s0, s1 = dist.shard_range(rank, world, v)               # this rank's variants (VariantsPca.scala:184)
        rounds = (max(dist.shard_range(r, world, v)[1] - dist.shard_range(r, world, v)[0] for r in range(world))
                  + args.chunk - 1) // args.chunk
        for r in range(rounds):
            a = min(s1, s0 + r * args.chunk)
            b = min(s1, a + args.chunk)
            thr = synth.thresholds(seed, a, b - a)
            if args.standin:
                bits = ingest.pack_bits(synth.genotypes(seed, a, thr, offs, dtype=np.uint8)) if b > a else \
                    np.zeros((0, (n + 31) // 32), dtype=np.uint32)
            else:
                bits = device_bitsets(owner, seed, offs, thr, a, n, torch.device("cuda", local_rank))
            fed += strips.feed_owners_from_variant_shards([owner], bits, chunk_variants=args.chunk)
    if not args.standin:
        owner.sync()
    t_gram = time.perf_counter() - t0

    t1 = time.perf_counter()
    trace = []
    comps, lam, nz = strips.compute_pca_over_strips([owner], args.num_pc, trace=trace)
    t_pcoa = time.perf_counter() - t1
    # residual of the returned pairs against the strips (one more mat-vec each): the check that needs no N x N matrix
    rs = strips.gather_concat([owner.strip_col_sums()])
    means, mm = rs / float(n), float(rs.sum()) / n / n
    res = []
    for c in range(args.num_pc):
        bu = strips.gather_concat([owner.strip_matvec(comps[:, c], means, mm)])
        res.append(float(np.linalg.norm(bu - lam[c] * comps[:, c]) / abs(lam[c])))
    out = None
    if rank == 0:
        out = {"workload": "configs[4] layout: %d samples x %d variants, S tiled by columns over %d strip owners, exchange=%s%s"
                           % (n, v, world, args.exchange, " (numpy stand-in)" if args.standin else ""),
               "strip_of_rank0": [int(c0), int(w)], "variants_fed_to_every_owner": int(fed),
               "gram_wall_s": t_gram, "variants_per_s": fed / t_gram if t_gram > 0 else None,
               "pcoa_wall_s": t_pcoa, "lanczos_checks": [int(t[0]) for t in trace],
               "eigenvalues": [float(x) for x in lam], "nonzero_rows": int(nz), "relative_residuals": res}
        print(json.dumps(out), flush=True)
    owner.close()
    if world > 1:
        td.barrier()
        td.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
