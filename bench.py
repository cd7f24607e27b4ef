#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, on N GPUs of one node.

Workload (config.workload): BASELINE configs[1] per GPU -- synthetic 2,504 samples x 1,000,000 variants
fp32 (10.0 GB) per step, resident in HBM before the timed region.  Weak scaling (default): every rank holds
min(K, 5) DISTINCT 1M-variant batches of its own shard (up to 5M variants = 50 GB per GPU: BASELINE configs[2]'s
shape, 40M variants at 8 GPUs; the generator is counter-based, so the cohort does not depend on N); step i takes
batch i mod 5.  --scaling strong: a FIXED cohort (--cohort, default 8M variants) is sharded over the ranks with
dist.shard_range and the K steps are one pass over the rank's resident shard.

One step = one pass of the hot path over one resident batch: the Gram accumulation of its variants
(re-layout pre-pass + matrix-core contraction: MX-FP4 for binary tiles, int8 for multiplicities; exact).  The job =
K steps, then ONE finalize (mirror) and, for N > 1, ONE RCCL all-reduce of S -- the reference reduces once per job too
(reduceByKey after all partitions, VariantsPca.scala:190).  The finalize/all-reduce is INSIDE the timed region.
value = variants of ALL ranks over the K steps / max-over-ranks wall time of the timed region.
The PCoA wall-clock (centring + eigensolve + D2H of N x 2, rank 0) is reported in `pcoa_wall_ms`.

Contract: one JSON line on stdout from rank 0.  Launched by the driver as
  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes itself under torch.distributed.run
with N ranks (dist.launch_plan); it exits with an error if fewer than N GPUs are visible or WORLD_SIZE != N.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "variants/sec into N×N Gram + PCoA wall-clock, 2504 samples, 1/2/4/8 GPU"
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: 256 CU x 2.4 GHz x 256 flop/clk/CU
PEAK_I8_MFMA_TOPS = 5000.0      # MI355X_MICROARCH.md: i8 MFMA ~2x the bf16 rate (~2.5 PF dense) => ~5 POP/s dense
PEAK_FP4_MFMA_TFLOPS = 10000.0  # MI355X_MICROARCH.md: MX-FP4 dense ~2x the fp8 rate (~5 PF dense) => ~10 PF dense
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
N_SAMPLES = 2504
SEED = 1002                     # seed of configs[1]


def pkg(sub=None):
    return importlib.import_module("spark-examples_amd" + ("." + sub if sub else ""))


def cpu_baseline(x_dev, n, budget_s=20.0):
    """The oracle's faithful pair loop (reference VariantsPca.scala:184-190 restated in C/OpenMP) timed
    on this box's host cores, on a bounded sample of the same workload.  Baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    oracle = importlib.import_module("variants_pca_oracle")
    cores = oracle.num_threads()
    probe = min(8192, x_dev.shape[0])
    xs = x_dev[:probe].cpu().numpy()
    oracle.similarity_from_dense(xs[:512], n)  # thread pool + page warm-up
    t0 = time.perf_counter()
    oracle.similarity_from_dense(xs, n)
    dt = max(time.perf_counter() - t0, 1e-4)
    rate = probe / dt
    sample = int(min(x_dev.shape[0], max(probe, rate * budget_s), 400000))
    xs = x_dev[:sample].cpu().numpy()
    t0 = time.perf_counter()
    s_ref = oracle.similarity_from_dense(xs, n)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    nb = min(sample, 100000)
    oracle.similarity_from_dense_blas(xs[:nb])
    dtb = time.perf_counter() - t0
    return {"value": sample / dt, "unit": "variants/s", "cores": cores, "kind": "port",
            "sample": "first %d variants of rank 0's shard, faithful pair loop (OpenMP, %d threads)" % (sample, cores),
            "seconds": dt, "blas_value": nb / dtb,
            "blas_note": "best-effort numpy/OpenBLAS sgemm X^T X on %d variants" % nb}, s_ref, sample


def pmc_for(key):
    """HBM traffic per 10^6 variants from rocprofv3 --pmc passes.  k-bits kernels: profiles/gram_pmc_live.json, written by
    `tools/gpu_round.sh <tag> pmclive` from the tree it ran on and named by that tree's source hash; a file of another
    tree is REFUSED (traffic: null).  The older kernels (f32 / i8 / fp4 operand) keep their static record."""
    P = pkg("_lib")
    here = P.source_hash()
    live = os.path.join(ROOT, "profiles", "gram_pmc_live.json")
    if key.startswith("kbits"):
        try:
            rec = json.load(open(live))
        except Exception:  # noqa: BLE001
            return {}, "no profiles/gram_pmc_live.json for this tree (source hash %s): run tools/gpu_round.sh <tag> pmclive" % here
        if rec.get("source_hash") != here:
            return {}, ("REFUSED: profiles/gram_pmc_live.json was measured on source hash %s, this tree is %s "
                        "(tools/gpu_round.sh <tag> pmclive makes a new one)" % (rec.get("source_hash"), here))
        return rec.get(key, {}), ("profiles/gram_pmc_live.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                                  "(--no-extras) on THIS source tree (hash %s, %s), per 10^6 variants, scaled to this run's "
                                  "launch size" % (here, rec.get("made", "?")))
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "gram_pmc_latest.json")))
        return rec.get(key, {}), "profiles/gram_pmc_latest.json (static record of an earlier round; NOT measured on this tree)"
    except Exception:  # noqa: BLE001
        return {}, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--variants", type=int, default=1000000, help="variants per step and GPU (default = BASELINE configs[1])")
    ap.add_argument("--distinct-batches", type=int, default=5,
                    help="weak scaling: distinct resident batches per GPU (5 x 1M variants = configs[2]'s 5M per GPU)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--cohort", type=int, default=8000000, help="--scaling strong: variants of the fixed cohort")
    ap.add_argument("--samples", type=int, default=N_SAMPLES)
    ap.add_argument("--ld", type=int, default=0,
                    help="row pitch of the resident fp32 tile in floats (0 = packed: ld = samples).  pcoa.h: a pitch that is a "
                         "multiple of 32 floats starts every row on a 128-byte line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pcoa-reps", type=int, default=3)
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the reference-only extras (uint8 input, config-0 CSR job, PCIe rates, Householder timing)")
    ap.add_argument("--allreduce", choices=["native", "torch"], default="native",
                    help="N>1: native = RCCL communicator inside libpcoa_hip (in-place int32), torch = "
                         "export -> torch.distributed.all_reduce -> import")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="N>1: torch.distributed backend (nccl = RCCL; gloo = CPU wire for a box where several ranks have to share "
                         "one GPU: it implies --allreduce torch -- the self-test of the multi-rank code path, not a measurement)")
    ap.add_argument("--rank-devices", type=str, default=None,
                    help="N>1: device ordinal of each rank, comma-separated (default: LOCAL_RANK); ordinals may repeat with --dist-backend gloo")
    ap.add_argument("--gram-kernel", choices=["auto", "fp4", "i8", "f32"], default="auto",
                    help="auto (default): binary tiles -> pack to MX-FP4 + v_mfma_f32_32x32x64_f8f6f4, tiles with "
                         "multiplicities -> int8; fp4 / i8: force one; f32: v_mfma_f32_32x32x2_f32")
    ap.add_argument("--operand", choices=["bits", "fp4"], default="bits",
                    help="form of the binary-tile operand in HBM: bits (default; 1 bit per genotype, expanded to MX-FP4 inside "
                         "the contraction) or fp4 (PCOA_FLAG_OPERAND_FP4, the r01 / r02 form)")
    ap.add_argument("--config2-variants", type=int, default=40000000,
                    help="extras: variants of the configs[2] cohort accumulated on this one GPU as bitsets (0 = skip)")
    ap.add_argument("--config3-samples", type=int, default=100000,
                    help="extras: sample count of the configs[3] job (100,000 samples x --config3-variants on this one GPU; 0 = skip)")
    ap.add_argument("--config3-variants", type=int, default=1000000)
    ap.add_argument("--sustained-seconds", type=float, default=2.0,
                    help="extras: length of the additional sustained run (same step, >= this many seconds of timed region)")
    args = ap.parse_args()

    import torch
    import torch.distributed as td

    # --gpus N without a rendezvous in the environment: become the launcher of N ranks (or fail loudly) instead of
    # running one rank and reporting n_gpus = 1
    visible = torch.cuda.device_count()
    rank_devices = [int(t) for t in args.rank_devices.split(",")] if args.rank_devices else None
    if rank_devices is not None:
        if len(rank_devices) != args.gpus or min(rank_devices) < 0 or max(rank_devices) >= max(visible, 1):
            sys.exit("bench.py: --rank-devices must name --gpus visible devices")
        if len(set(rank_devices)) < len(rank_devices) and args.dist_backend != "gloo":
            sys.exit("bench.py: --rank-devices repeats a device: RCCL needs one GPU per rank (use --dist-backend gloo for the self-test)")
        visible = max(visible, args.gpus)   # the map names what exists; the rank count is what the plan checks
    what, detail = pkg("dist").launch_plan(args.gpus, os.environ, visible, [os.path.abspath(__file__)] + sys.argv[1:],
                                           python=sys.executable)
    if what == "error":
        sys.exit("bench.py: " + detail)
    if what == "spawn":
        import subprocess
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(detail, env=env))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank_devices is not None and world > 1:
        local_rank = rank_devices[rank]
    gloo = world > 1 and args.dist_backend == "gloo"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if gloo:
            td.init_process_group("gloo", rank=rank, world_size=world)
        else:
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if gloo else dev   # where the small collectives' tensors live

    P = pkg()
    dist = pkg("dist")
    synth = pkg("synth")
    n, v = args.samples, args.variants
    eng = P.PcoaEngine(n, device=local_rank, gram_kernel=args.gram_kernel, operand=args.operand)
    dev_name, cus = eng.device_info()
    eng.reserve(v, 2)   # operand buffers + computePca workspace now, not inside the first calls (pcoa_reserve)

    # ---- resident input: this rank's shard of the cohort, generated on device --------------------------
    offs = synth.pop_offsets(n)
    steps_n = max(args.steps, 1)
    if args.scaling == "strong":
        s0, s1 = dist.shard_range(rank, world, args.cohort)      # the reference's partitioning (VariantsPca.scala:184)
        first, resident = s0, s1 - s0
        cuts = [resident * i // steps_n for i in range(steps_n + 1)]
        batch_of = lambda i: (cuts[i % steps_n], cuts[i % steps_n + 1])   # noqa: E731
        variants_per_job = args.cohort                             # fixed total work
    else:
        nb = max(1, min(args.distinct_batches, steps_n))
        first, resident = rank * nb * v, nb * v
        batch_of = lambda i: ((i % nb) * v, (i % nb) * v + v)      # noqa: E731
        variants_per_job = world * v * steps_n
    ld = args.ld if args.ld >= n else n
    x_store = torch.empty((resident, ld), dtype=torch.float32, device=dev)
    if ld > n:
        x_store[:, n:].fill_(float("nan"))        # pitch padding is never read as data: NaN would fail the 0/1 check
    x = x_store[:, :n]
    step_rows = 1 << 18
    for v0 in range(0, resident, step_rows):
        v1 = min(resident, v0 + step_rows)
        eng.synth_fill(SEED, offs, synth.thresholds(SEED, first + v0, v1 - v0), first + v0, x[v0:v1].data_ptr(), ld)
    eng.sync()

    scratch = None
    native = None
    allreduce_mode = "none"
    if world > 1:
        allreduce_mode = "torch" if gloo else args.allreduce   # (the library's communicator is RCCL)
        if allreduce_mode == "native":
            # every rank must end up on the same path: agree on success before using it
            ok = 1
            try:
                native = dist.NativeComm(eng)
            except Exception as exc:  # noqa: BLE001
                sys.stderr.write("rank %d: native RCCL communicator failed (%s); using torch.distributed\n" % (rank, exc))
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=cdev)
            td.all_reduce(flag, op=td.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if native is not None:
                    native.close()
                native = None
                allreduce_mode = "torch"

    def one_step(i=0):
        a, b = batch_of(i)
        eng.accumulate_dense(x[a:b])

    def finish_job():
        nonlocal scratch
        if world > 1:
            if native is not None:
                native.allreduce()
            else:
                scratch = dist.allreduce_engine(eng, scratch=scratch)
        else:
            eng.finalize()

    def fence():
        eng.sync()
        torch.cuda.synchronize(dev)
        if world > 1:
            td.barrier()

    eng.reset()
    for _ in range(args.warmup):
        one_step()
    if native is not None:
        # first use of the native communicator: if it fails on ANY rank, every rank moves to torch.distributed
        ok = 1
        try:
            native.allreduce()
            eng.sync()
        except Exception as exc:  # noqa: BLE001
            sys.stderr.write("rank %d: native RCCL all-reduce failed (%s); using torch.distributed\n" % (rank, exc))
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=cdev)
        td.all_reduce(flag, op=td.ReduceOp.MIN)
        if int(flag.item()) == 0:
            try:
                native.close()
            except Exception:  # noqa: BLE001
                pass
            native = None
            allreduce_mode = "torch"
            eng.reset()
            one_step()
    finish_job()  # also warms the collective
    fence()
    eng.reset()
    eng.reset_timings()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(i)
    t_local = t_red = None
    if world > 1:
        # (multi-rank only) this rank's own share ends when its partial S is finalized; the reduction step is timed apart
        eng.finalize()
        eng.sync()
        t_local = time.perf_counter() - t0
    t_r0 = time.perf_counter()
    finish_job()
    if world > 1:
        eng.sync()
        t_red = time.perf_counter() - t_r0
    fence()
    elapsed = time.perf_counter() - t0
    elapsed_rank = elapsed
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        elapsed = float(tt.item())
    tim = eng.timings()
    # what a multi-rank record needs to explain itself: per-rank elapsed, the reduction step, the communicator's own rank count
    tele = dist.collective_telemetry(t_local if t_local is not None else elapsed_rank, t_red if t_red is not None else 0.0, tim,
                                     native.count() if native is not None else None, device=cdev if world > 1 else None)

    out = None
    if rank == 0:
        steps = max(args.steps, 1)
        ms_per_step = 1e3 * elapsed / steps
        value = variants_per_job / elapsed
        launches = max(int(tim["gram_kernel_launches"]), 1)
        kbits = int(tim["operand_bits"]) == 1
        info = {"pipeline": bool(tim["pipeline_launches"] > 0), "lockstep": bool(tim["lockstep_launches"] > 0),
                "even_split": bool(tim["evensplit_launches"] > 0), "operand_bits_per_genotype": int(tim["operand_bits"]),
                "pipeline_launches": int(tim["pipeline_launches"]), "lockstep_launches": int(tim["lockstep_launches"]),
                "evensplit_launches": int(tim["evensplit_launches"]),
                "pipeline_pre_pass_cus": int(tim["pipeline_pre_pass_cus"]),
                "pipeline_contraction_cus": int(tim["pipeline_contraction_cus"]),
                "what": "fp32 tiles at this N: the pre-pass of operand buffer k+1 runs beside the contraction of buffer k "
                        "(one workgroup per CU on pipeline_contraction_cus CUs, placed first; the pre-pass cannot share a CU "
                        "with it and takes the rest) -- DESIGN_HISTORY.md 4.1; PCOA_PIPELINE=0 disables it"}
        # k-bits operand: the two kernels SHARE the CUs (persistent ring pre-pass, 40 VGPRs, beside a contraction held to 224)
        cores = bool(tim["pipeline_pre_pass_cus"] + tim["pipeline_contraction_cus"] > cus and tim["pipeline_launches"] > 0)
        info["co_resident"] = cores
        if cores:
            info["what"] = ("fp32 tiles at this N, k-bits operand: the pre-pass of operand buffer k+1 (pack_kbits_ring_kernel: "
                            "persistent, LDS-DMA ring, two workgroups of 4 waves per CU, 40 VGPRs) runs on the SAME CUs as the "
                            "contraction of buffer k (gram_kbits_kernel held to 224 VGPRs per wave, one workgroup on each of "
                            "pipeline_contraction_cus CUs, placed first) -- DESIGN_HISTORY.md 4.1, profiles/r03s..u_coreside.txt; "
                            "PCOA_KBITS_CORESIDE=0 gives the disjoint-CU form back, PCOA_PIPELINE=0 the serial order")
        kern_s = tim["gram_kernel_seconds"] / launches           # average Gram-kernel launch duration (HIP events)
        vpl = tim["gram_variants"] / launches                    # variants per contraction launch
        flops_per_launch = 2.0 * vpl * n * n                     # algorithmic 2*V*N^2
        achieved = flops_per_launch / kern_s / 1e12 if kern_s > 0 else 0.0
        pmc, pmc_src = pmc_for("kbits" if kbits else {1: "f32", 2: "i8", 3: "fp4"}[tim["gram_kernel_kind"]])
        kind = tim["gram_kernel_kind"]          # what actually ran: 1 fp32, 2 int8, 3 MX-FP4
        pipe = bool(info.get("pipeline"))
        gram_cus = info.get("pipeline_contraction_cus", cus) if pipe else cus
        if kind == 3 and kbits:
            tile, peak = 256, PEAK_FP4_MFMA_TFLOPS
            w4_name = bool(not (bool(info.get("pipeline")) and info.get("co_resident")) and os.environ.get("PCOA_KBITS_W4", "1") != "0")
            kname = "%s (k-bits operand expanded to MX-FP4 in registers; %s launch)" % (
                "gram_kbits_w4_kernel<4, 0, 69> (one wave per SIMD, hand-placed pipeline)" if w4_name else
                "gram_kbits_kernel<3, 2, 2> (two waves per SIMD, ping-pong)",
                "even-split" if info.get("even_split") else "lock-step" if info.get("lockstep") else "split-K")
            kdesc = ("pack fp32->k-bits (1 bit per genotype, HBM-bound, verifies values are 0/1) + MX-FP4 MFMA "
                     "v_mfma_f32_32x32x64_f8f6f4 (unscaled form, exact) fed from bit words expanded in registers (0.5 x 2.0 "
                     "conjugate E2M1 weights), upper-triangular 256x256 tiles, fp32 accumulators (< 2^24 per launch) -> "
                     "int32 atomics")
        elif kind == 3:
            tile, peak = 256, PEAK_FP4_MFMA_TFLOPS
            kname = "gram_packed_kernel<1, 2, 2, 4, 3, true, 2> (FMT 1 = MX-FP4, ping-pong; %s launch)" % (
                "lock-step" if info.get("lockstep") or pipe else "split-K")
            kdesc = ("pack fp32->k-blocked FP4 E2M1 (HBM-bound, verifies values are 0/1) + MX-FP4 MFMA "
                     "v_mfma_f32_32x32x64_f8f6f4 (unscaled form, exact), upper-triangular 256x256 tiles, "
                     "fp32 accumulators (< 2^24 per launch) -> int32 atomics")
        elif kind == 2:
            tile, peak, kname = 256, PEAK_I8_MFMA_TOPS, "gram_packed_kernel<0, 2, 2, 4, 3, true, 2> (FMT 0 = int8, ping-pong)"
            kdesc = "pack fp32->k-blocked int8 (HBM-bound) + i8 MFMA v_mfma_i32_32x32x32_i8, upper-triangular 256x256 tiles, split-K, int32 accumulators"
        else:
            tile, peak, kname = 128, PEAK_FP32_MFMA_TFLOPS, "gram_f32_kernel"
            kdesc = "fp32 MFMA v_mfma_f32_32x32x2_f32, upper-triangular 128x128 tiles, split-K"
        # which k-bits contraction ran: beside the ring pre-pass the two-waves-per-SIMD kernel (one idle wave pair per diagonal
        # tile), everywhere else gram_kbits_w4_kernel (wave roles on diagonal tiles) unless PCOA_KBITS_W4=0
        w4_ran = bool(kind == 3 and kbits and not (pipe and info.get("co_resident")) and os.environ.get("PCOA_KBITS_W4", "1") != "0")
        frac_issued = issued_fraction(n, tile, diag=w4_diag_share() if w4_ran else None)
        issued = achieved * frac_issued
        roof_gram = {"bound": "mfma", "achieved": issued, "peak": peak, "unit": "TFLOP/s", "frac": issued / peak,
                     "convention": "ISSUED matrix-core work: only upper-triangular tiles are computed and the below-diagonal "
                                   "waves of diagonal tiles issue nothing (%.3f of the 2*V*N^2 of SURVEY 8d)" % frac_issued,
                     "algorithmic_tflops": achieved, "algorithmic_frac": achieved / peak,
                     "useful_frac": achieved * useful_fraction(n) / peak,
                     "useful_note": "the upper triangle with its diagonal, V*N*(N+1) flops: what S = X^T X needs at all",
                     "algorithmic_note": "2*V*N^2 per launch / launch duration (both triangles credited: can exceed 1)",
                     "cus_used": gram_cus, "frac_of_cus_used": issued / (peak * gram_cus / float(cus)),
                     # PMC traffic is stored per 10^6 variants and scaled to this run's average launch, like `achieved`
                     "traffic": (pmc["gram_hbm_bytes_per_mvariants"] * vpl / 1e6
                                 if "gram_hbm_bytes_per_mvariants" in pmc else pmc.get("gram_hbm_bytes_per_launch")),
                     "traffic_source": pmc_src, "variants_per_launch": vpl,
                     "kernel": kname, "avg_launch_ms": 1e3 * kern_s, "launches": launches}
        roof_pack = None
        if kind != 1 and tim["pack_launches"] > 0:
            pl = int(tim["pack_launches"])
            pack_s = tim["pack_seconds"] / pl
            read_bytes = 4.0 * (tim["gram_variants"] / pl) * n               # X read once, 4*V*N
            read_gbs = read_bytes / pack_s / 1e9 if pack_s > 0 else 0.0
            all_gbs = tim["pack_bytes"] / pl / pack_s / 1e9 if pack_s > 0 else 0.0
            pack_cus = info.get("pipeline_pre_pass_cus", cus) if pipe else cus
            roof_pack = {"bound": "hbm", "achieved": read_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": read_gbs / PEAK_HBM_GBS,
                         "bytes_convention": "SURVEY 8(d): 4*V*N bytes of X read once per launch",
                         "achieved_incl_operand_writes": all_gbs, "frac_incl_operand_writes": all_gbs / PEAK_HBM_GBS,
                         "incl_note": "also credits the V*Npad%s bytes of packed operand the pre-pass writes" % (
                             "/8" if kbits else "/2" if kind == 3 else ""),
                         "cus_used": pack_cus,
                         "traffic": (pmc["pack_hbm_bytes_per_mvariants"] * (tim["gram_variants"] / pl) / 1e6
                                     if "pack_hbm_bytes_per_mvariants" in pmc else pmc.get("pack_hbm_bytes_per_launch")),
                         "traffic_source": pmc_src,
                         "kernel": (("pack_kbits_ring_kernel<8, 0, 0>" if info.get("co_resident") else
                                     "pack_kbits_kernel<float, 4, true>") if kbits else "pack_fp4_kernel<float, 4, true>")
                         if kind == 3 else "pack_f32_i8_kernel<4>",
                         "avg_launch_ms": 1e3 * pack_s, "launches": pl}
            if pipe and info.get("co_resident"):
                roof_pack["note"] = ("fp32 pipeline, co-resident form: this kernel's waves share all %d CUs with the %d workgroups "
                                     "of the previous buffer's contraction, and have the chip to themselves once that is done -- "
                                     "its launch duration ~ the step; alone on the chip: roofline_standalone" % (cus, gram_cus))
            elif pipe:
                roof_pack["note"] = ("fp32 pipeline: this kernel runs BESIDE the contraction of the previous buffer -- on the "
                                     "%d of %d CUs the contraction's %d workgroups leave, and on all of them once it is done -- "
                                     "so its launch duration ~ the step; alone on the whole chip it takes ~2.0 ms per 10^6 "
                                     "variants (roofline_standalone; profiles/r03*_kbits_harness.txt)" % (pack_cus, cus, gram_cus))
        # the dominant kernel (larger share of the step) goes into `roofline`, the other into `roofline_other`
        if roof_pack is not None and tim["pack_seconds"] > tim["gram_kernel_seconds"]:
            roofline, roofline_other = roof_pack, roof_gram
        else:
            roofline, roofline_other = roof_gram, roof_pack
        # whole step against the HBM roofline: SURVEY 8(d) algorithmic bytes (4*N per variant + 4*N^2 of S per job)
        step_bytes = 4.0 * n * (variants_per_job / world) / steps + 4.0 * n * n / steps
        step_hbm_frac = step_bytes / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS
        # PCoA wall-clock on rank 0 (S of the last step is in place)
        pcoa = []
        for _ in range(max(args.pcoa_reps, 1)):
            t1 = time.perf_counter()
            comps, lam, nz = eng.compute(2)
            pcoa.append(1e3 * (time.perf_counter() - t1))
        tim2 = eng.timings()
        # the dense (gap-independent) eigensolver on the same S, for reference
        pcoa_hh = []
        with P.PcoaEngine(n, device=local_rank, eig="householder") as eng_hh:
            eng_hh.load_gram(eng.gram())
            for _ in range(0 if args.no_extras else 2):
                t1 = time.perf_counter()
                comps_hh, lam_hh, _ = eng_hh.compute(2)
                pcoa_hh.append(1e3 * (time.perf_counter() - t1))
            tim_hh = eng_hh.timings()
        # the gap-independent Krylov path (r06): Lanczos as the band iteration from the start (block width num_pc + 2), which is
        # what the engine falls back to for clustered leading eigenvalues before the dense solver
        pcoa_band, band_steps, agree_band = [], None, None
        if not args.no_extras:
            with P.PcoaEngine(n, device=local_rank, eig="band") as eng_b:
                eng_b.load_gram(eng.gram())
                for _ in range(3):
                    t1 = time.perf_counter()
                    comps_b, lam_b, _ = eng_b.compute(2)
                    pcoa_band.append(1e3 * (time.perf_counter() - t1))
                band_steps = int(eng_b.timings()["lanczos_block_steps"])
            agree_band = float(max(np.linalg.norm(comps[:, c] - comps_b[:, c] * np.sign(np.dot(comps[:, c], comps_b[:, c]))) for c in range(2)))
        agree = None if args.no_extras else float(max(
            np.linalg.norm(comps[:, c] - comps_hh[:, c] * np.sign(np.dot(comps[:, c], comps_hh[:, c]))) for c in range(2)))
        out = {
            "metric": METRIC, "value": value, "unit": "variants/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": {1: "f32", 2: "i8->i32", 3: "fp4(e2m1)->f32->i32"}[kind], "data": "synthetic",
            "config": {"workload": ("configs[1]: synthetic %d samples x %d variants fp32 per step and GPU, resident in HBM "
                                    "(Gram + eig on rank 0); %s" % (
                                        n, v if args.scaling == "weak" else resident // steps, ("weak scaling, %d distinct resident batches per GPU (configs[2]: 5M variants per "
                                               "GPU), step i takes batch i mod %d" % (resident // v, resident // v))
                                        if args.scaling == "weak" else
                                        "STRONG scaling: fixed cohort of %d variants sharded over %d ranks, K steps = one pass"
                                        % (args.cohort, world))),
                       "n_samples": n, "row_pitch_floats": ld, "variants_per_step_per_gpu": v if args.scaling == "weak" else resident // steps,
                       "resident_variants_per_gpu": resident, "seed": SEED,
                       "parallelism": "variant-sharded x%d" % world,
                       "allreduce": allreduce_mode, "dist_backend": args.dist_backend if world > 1 else None,
                       "gram_kernel": kdesc, "gram_kernel_mode": args.gram_kernel, "operand": args.operand,
                       "fp4_fallback_chunks": int(tim["fp4_fallbacks"])},
            "roofline": roofline, "roofline_other": roofline_other,
            # multi-GPU telemetry (present at N = 1 too, so that the line has one shape): did RCCL see N ranks, what did the
            # collective cost, how uneven were the ranks, and the per-GPU rate to hold against the N = 1 line
            "rccl_ranks": tele["rccl_ranks"], "allreduce_ms": tele["allreduce_ms"] if world > 1 else None,
            "allreduce_event_ms": tele["allreduce_event_ms"], "allreduce_int32_in_place": tele["allreduce_int32_in_place"],
            "rank_elapsed_s": tele["rank_elapsed_s"], "rank_elapsed_min_s": tele["rank_elapsed_min_s"],
            "rank_elapsed_max_s": tele["rank_elapsed_max_s"], "n1_equivalent_value": value / world,
            "multi_gpu_note": "rank_elapsed_s = each rank's time to its finalized partial S; allreduce_ms = wall of the reduction "
                              "step (max over ranks), allreduce_event_ms = the collective's HIP-event time on rank 0's stream; "
                              "rccl_ranks = ncclCommCount of the library's communicator (None with --allreduce torch); "
                              "n1_equivalent_value = value / n_gpus",
            "step_hbm_frac": step_hbm_frac,
            "step_hbm_note": "SURVEY 8(d) algorithmic bytes of a step (4*N per variant, X read once) / ms_per_step / 8 TB/s",
            "pipeline": info,
            "gram_ms_per_step": 1e3 * tim["gram_kernel_seconds"] / steps,
            "pack_ms_per_step": 1e3 * tim["pack_seconds"] / steps,
            "finalize_ms_per_step": 1e3 * tim["finalize_seconds"] / steps,
            "pcoa_wall_ms": float(np.median(pcoa)), "pcoa_wall_ms_all": pcoa,
            "pcoa_breakdown_ms": {k: 1e3 * tim2[k] / max(args.pcoa_reps, 1) for k in
                                  ("center_seconds", "lanczos_seconds", "tridiag_seconds", "eig_seconds",
                                   "backtransform_seconds")},
            "pcoa_method": {1: "lanczos (verified residual)", 2: "householder"}.get(tim2["eig_method"], "?"),
            "lanczos_steps": tim2["lanczos_steps"],
            "pcoa_wall_ms_householder": float(min(pcoa_hh)) if pcoa_hh else None,
            "pcoa_householder_breakdown_ms": {k: 1e3 * tim_hh[k] / 2 for k in
                                              ("center_seconds", "tridiag_seconds", "eig_seconds",
                                               "backtransform_seconds")},
            "pcoa_paths_max_vector_diff": agree,
            "pcoa_wall_ms_band_lanczos": float(min(pcoa_band)) if pcoa_band else None, "band_lanczos_basis_vectors": band_steps,
            "pcoa_band_vs_default_max_vector_diff": agree_band,
            "eigenvalues": [float(t) for t in lam], "nonzero_rows": int(nz),
            "device": dev_name, "cu_count": cus,
            # what tools/pmc_live.py needs to turn per-process counter totals into per-variant traffic
            "pmc_manifest": {"source_hash": pkg("_lib").source_hash(), "no_extras": bool(args.no_extras),
                             "fp32_variants_through_the_engine": int((args.warmup + args.steps) * v) if args.scaling == "weak" else None,
                             "pipeline": bool(info["pipeline"])},
        }
        if not args.no_extras and info["pipeline"]:
            # the two kernels STANDALONE (each alone on the whole chip, strictly serial on one stream): what their own
            # rooflines look like without the other kernel beside them -- three steps on a PCOA_FLAG_NO_PIPELINE engine
            with P.PcoaEngine(n, device=local_rank, gram_kernel=args.gram_kernel, pipeline=False, operand=args.operand) as es:
                a0, b0 = batch_of(0)
                es.reserve(b0 - a0, 0)
                es.accumulate_dense(x[a0:b0]); es.finalize(); es.sync()
                es.reset(); es.reset_timings(); es.sync()
                t1 = time.perf_counter()
                for i in range(3):
                    a0, b0 = batch_of(i)
                    es.accumulate_dense(x[a0:b0])
                es.finalize(); es.sync()
                dts = time.perf_counter() - t1
                ts = es.timings()
            vs = (batch_of(0)[1] - batch_of(0)[0])
            ps = ts["pack_seconds"] / max(int(ts["pack_launches"]), 1)
            gs = ts["gram_kernel_seconds"] / max(int(ts["gram_kernel_launches"]), 1)
            gv = ts["gram_variants"] / max(int(ts["gram_kernel_launches"]), 1)
            alone_w4 = os.environ.get("PCOA_KBITS_W4", "1") != "0"
            alone_diag = w4_diag_share() if alone_w4 else 0.75
            out["roofline_standalone"] = {
                "note": "pre-pass and contraction without the pipeline (PCOA_FLAG_NO_PIPELINE engine, 3 steps): each alone on all "
                        "%d CUs -- the pre-pass as pack_kbits_kernel (the ring form only runs beside a contraction; alone the two "
                        "are equally fast, profiles/r03s..w), the contraction as the even-split launch; ms_per_step is what the "
                        "serial order costs on this box" % cus,
                "ms_per_step": 1e3 * dts / 3, "variants_per_s": 3 * vs / dts,
                "pre_pass": {"bound": "hbm", "avg_launch_ms": 1e3 * ps, "achieved": 4.0 * vs * n / ps / 1e9, "peak": PEAK_HBM_GBS,
                             "unit": "GB/s", "frac": 4.0 * vs * n / ps / 1e9 / PEAK_HBM_GBS,
                             "bytes_convention": "SURVEY 8(d): 4*V*N bytes of X read once per launch"},
                "contraction": {"bound": "mfma", "avg_launch_ms": 1e3 * gs, "unit": "TFLOP/s", "peak": PEAK_FP4_MFMA_TFLOPS,
                                "achieved": 2.0 * gv * n * n / gs / 1e12 * issued_fraction(n, 256, diag=alone_diag),
                                "frac": 2.0 * gv * n * n / gs / 1e12 * issued_fraction(n, 256, diag=alone_diag) / PEAK_FP4_MFMA_TFLOPS,
                                "useful_frac": 2.0 * gv * n * n / gs / 1e12 * useful_fraction(n) / PEAK_FP4_MFMA_TFLOPS,
                                "algorithmic_frac": 2.0 * gv * n * n / gs / 1e12 / PEAK_FP4_MFMA_TFLOPS,
                                "convention": "frac = ISSUED matrix-core work (%.4f of 2*V*N^2: upper-triangular 256 x 256 tiles, %.4f of the "
                                              "MFMAs of a diagonal tile); useful_frac = V*N*(N+1) (the upper triangle itself, %.4f); "
                                              "algorithmic_frac = SURVEY 8(d)'s 2*V*N^2" % (
                                                  issued_fraction(n, 256, diag=alone_diag), alone_diag, useful_fraction(n)),
                                "kernel": "gram_kbits_w4_kernel<4, 0, 69>" if alone_w4 else "gram_kbits_kernel<3, 2, 2>",
                                "variants_per_launch": gv}}
            pmc_alone, pmc_alone_src = pmc_for("kbits_standalone")
            for k2, fld, per in (("pre_pass", "pack_hbm_bytes_per_mvariants", vs), ("contraction", "gram_hbm_bytes_per_mvariants", gv)):
                out["roofline_standalone"][k2]["traffic"] = pmc_alone[fld] * per / 1e6 if fld in pmc_alone else None
                out["roofline_standalone"][k2]["traffic_source"] = pmc_alone_src
            # the same two figures next to the co-running ones, where a reader of `roofline` looks first
            for obj in (out.get("roofline"), out.get("roofline_other")):
                if not isinstance(obj, dict):
                    continue
                alone = out["roofline_standalone"]["pre_pass" if obj.get("bound") == "hbm" else "contraction"]
                obj["alone_on_the_chip"] = {"avg_launch_ms": alone["avg_launch_ms"], "achieved": alone["achieved"],
                                            "frac": alone["frac"], "source": "roofline_standalone (same run, same box)"}
        if not args.no_extras and world == 1:
            # the same step, sustained: a timed region of >= --sustained-seconds (boost clocks, thermal state and a 53-ms
            # burst are different things); S would pass int32 on the way, so the int64 fold is part of it, as in a real
            # whole-genome stream
            reps_s = max(int(args.sustained_seconds / max(ms_per_step * 1e-3, 1e-6)) + 1, steps)
            eng.reset(); eng.reset_timings(); fence()
            t1 = time.perf_counter()
            for i in range(reps_s):
                one_step(i)
            finish_job()
            fence()
            dts = time.perf_counter() - t1
            out["sustained"] = {"steps": reps_s, "seconds": dts, "ms_per_step": 1e3 * dts / reps_s,
                                "value": reps_s * (variants_per_job / steps) / dts, "unit": "variants/s",
                                "vs_timed_region": (reps_s * (variants_per_job / steps) / dts) / value,
                                "note": "the timed region of `value` repeated for >= %.1f s (same batches, same finalize)" % args.sustained_seconds}
            eng.reset()
            one_step(steps - 1)
            finish_job()
            fence()
        if not args.no_extras:
            # north_star's literal kernel: the fp32-MFMA Gram (v_mfma_f32_32x32x2_f32) on a slice of the same batch, so that
            # the record carries its roofline fraction too ("at >= 40 % fp32-MFMA roofline")
            vf = int(min(131072, resident))
            with P.PcoaEngine(n, device=local_rank, gram_kernel="f32") as ef:
                ef.accumulate_dense(x[:vf]); ef.finalize(); ef.sync()
                ef.reset(); ef.reset_timings()
                ef.accumulate_dense(x[:vf]); ef.finalize(); ef.sync()
                tf = ef.timings()
            f32_s = tf["gram_kernel_seconds"] / max(int(tf["gram_kernel_launches"]), 1)
            f32_alg = 2.0 * vf * n * n / f32_s / 1e12 if f32_s > 0 else 0.0
            f32_iss = f32_alg * issued_fraction(n, 128)
            out["roofline_f32_mfma"] = {
                "bound": "mfma", "achieved": f32_iss, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": f32_iss / PEAK_FP32_MFMA_TFLOPS, "convention": "issued (upper-triangular 128x128 tiles)",
                "algorithmic_tflops": f32_alg, "algorithmic_frac": f32_alg / PEAK_FP32_MFMA_TFLOPS,
                "variants_per_s": vf / f32_s if f32_s > 0 else 0.0, "variants": vf, "kernel": "gram_f32_kernel<4>",
                "avg_launch_ms": 1e3 * f32_s,
                "note": "gram_kernel=f32 (PCOA_FLAG_GRAM_F32_MFMA) on the first %d variants of the batch; not the default path" % vf}
        if world == 1 and not args.no_extras:
            # the int8-MFMA path (tiles with carrier multiplicities; forced here on the binary cohort: same kernels, same bytes):
            # pre-pass and contraction in series (r06 tried the pre-pass on a stream of its own with two workspaces: the two
            # kernels do not co-reside -- the contraction's waves hold every register of a CU -- and the step stayed the sum)
            with P.PcoaEngine(n, device=local_rank, gram_kernel="i8") as e8:
                xb = x[:min(v, resident)]
                for _ in range(2):
                    e8.accumulate_dense(xb)
                e8.finalize(); e8.sync(); e8.reset(); e8.reset_timings(); e8.sync()
                t1 = time.perf_counter()
                for _ in range(steps):
                    e8.accumulate_dense(xb)
                e8.finalize(); e8.sync()
                dti = time.perf_counter() - t1
                ti8 = e8.timings()
            out["int8_path"] = {"value": xb.shape[0] * steps / dti, "unit": "variants/s", "ms_per_step": 1e3 * dti / steps,
                                "pack_ms_per_step": 1e3 * ti8["pack_seconds"] / steps, "gram_ms_per_step": 1e3 * ti8["gram_kernel_seconds"] / steps,
                                "note": "PCOA_FLAG_GRAM_I8_MFMA on the same fp32 batch: pack fp32 -> int8 (HBM-bound) + v_mfma_i32_32x32x32_i8; "
                                        "in series on the ctx stream (pcoa.h: this path is not pipelined)"}
        if world == 1 and not args.no_extras:
            # the production input format (one byte per genotype): same contraction, 4x cheaper pre-pass
            x1 = x[:min(v, resident)]          # the extras below work on the first resident batch
            x8 = x1.to(torch.uint8)
            torch.cuda.synchronize(dev)
            eng.reset()
            for _ in range(2):
                eng.accumulate_dense_u8(x8)
            eng.finalize(); eng.sync()
            eng.reset(); eng.reset_timings(); eng.sync()
            t1 = time.perf_counter()
            for _ in range(steps):
                eng.accumulate_dense_u8(x8)
            eng.finalize(); eng.sync()
            dt8 = time.perf_counter() - t1
            t8 = eng.timings()
            s_dense = eng.gram()     # S of `steps` passes over the batch: what the other boundaries are held to below
            sus8 = sustained_leg(eng, lambda: eng.accumulate_dense_u8(x8), v, dt8 / steps, args.sustained_seconds / 2)
            out["alt_input_u8"] = {"value": v * steps / dt8, "unit": "variants/s", "ms_per_step": 1e3 * dt8 / steps, "sustained": sus8,
                                   "pack_ms_per_step": 1e3 * t8["pack_seconds"] / steps,
                                   "gram_ms_per_step": 1e3 * t8["gram_kernel_seconds"] / steps,
                                   "pipeline_launches": int(t8["pipeline_launches"]),
                                   "note": "same cohort handed over as uint8 [V][N] (pcoa_accumulate_dense_u8); "
                                           "not the BASELINE configs[1] fp32 boundary, reported for reference.  "
                                           "pipeline_launches > 0: the pre-pass of one operand buffer ran beside the "
                                           "contraction of the previous one (co-resident form), so pack + gram > step"}
            del x8
            # the bit-packed boundary (1 bit per genotype, 313 B per variant): expand to FP4 + the same contraction
            words = (n + 31) // 32
            bits = torch.empty((x1.shape[0], words), dtype=torch.int32, device=dev)
            wts = (1 << torch.arange(32, device=dev, dtype=torch.int64))
            for r0 in range(0, x1.shape[0], 1 << 16):
                xb = torch.nn.functional.pad(x1[r0:r0 + (1 << 16)] > 0, (0, words * 32 - n))
                val = (xb.view(-1, words, 32).to(torch.int64) * wts).sum(dim=2)
                bits[r0:r0 + val.shape[0]] = torch.where(val >= 2 ** 31, val - 2 ** 32, val).to(torch.int32)
            del xb, val
            torch.cuda.synchronize(dev)
            eng.reset()
            for _ in range(2):
                eng.accumulate_bits(bits)
            eng.finalize(); eng.sync()
            eng.reset(); eng.reset_timings(); eng.sync()
            t1 = time.perf_counter()
            for _ in range(steps):
                eng.accumulate_bits(bits)
            eng.finalize(); eng.sync()
            dtb = time.perf_counter() - t1
            tb = eng.timings()
            same_bits = bool(np.array_equal(eng.gram(), s_dense))
            susb = sustained_leg(eng, lambda: eng.accumulate_bits(bits), v, dtb / steps, args.sustained_seconds / 2)
            out["alt_input_bits"] = {"value": v * steps / dtb, "unit": "variants/s", "ms_per_step": 1e3 * dtb / steps, "sustained": susb,
                                     "expand_ms_per_step": 1e3 * tb["pack_seconds"] / steps,
                                     "gram_ms_per_step": 1e3 * tb["gram_kernel_seconds"] / steps,
                                     "pipeline_launches": int(tb["pipeline_launches"]),
                                     "bytes_per_variant": 4 * words,
                                     "same_gram_as_dense_input": same_bits,
                                     "note": "same cohort as carrier bitsets [V][ceil(N/32)] uint32 "
                                             "(pcoa_accumulate_bits, SURVEY 8d '1-bit-packed twin'); reported separately"}
            del bits
            out["csr_boundary"] = csr_boundary(P, torch, dev, local_rank, n, x1, s_dense, steps, args.operand)
            out["plink_bed_boundary"] = plink_bed_boundary(P, torch, dev, local_rank, n, x1, s_dense, steps, args.operand)
        if world == 1 and not args.no_extras and args.config2_variants > 0:
            # BASELINE configs[2] on ONE GPU: the whole 40 M-variant cohort as carrier bitsets (12.6 GB, resident), one job
            out["config2_one_gpu_bits"] = config2_one_gpu_bits(P, synth, torch, dev, local_rank, n, args.config2_variants,
                                                               args.operand)
        if world == 1 and not args.no_extras:
            # BASELINE configs[0] stand-in: BRCA1-sized region (2,500 variants) through the faithful CSR boundary
            # (pcoa_accumulate_calls, host arrays), end to end: H2D + densify + Gram + finalize + PCoA + D2H
            v1 = 2500
            thr1 = synth.thresholds(1001, 0, v1)
            x1 = synth.genotypes(1001, 0, thr1, offs, dtype=np.uint8)
            x1 = x1[x1.any(axis=1)]
            offs1 = np.concatenate([[0], np.cumsum(x1.sum(axis=1, dtype=np.int64))]).astype(np.int64)
            idx1 = np.nonzero(x1)[1].astype(np.int32)
            with P.PcoaEngine(n, device=local_rank) as e1:
                walls = []
                for _ in range(4):
                    e1.reset()
                    t1 = time.perf_counter()
                    e1.accumulate_calls(idx1, offs1)
                    e1.finalize()
                    c1, l1, _ = e1.compute(2)
                    walls.append(1e3 * (time.perf_counter() - t1))
                tt1 = e1.timings()
            # the same job on the CPU restatement at 4 threads (= the reference's local[4]) -- baseline only
            cpu_c0 = None
            if not args.no_cpu_baseline:
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                oracle = importlib.import_module("variants_pca_oracle")
                prev = oracle.num_threads()
                oracle.set_num_threads(4)
                t1 = time.perf_counter()
                s_c0 = oracle.calculate_similarity_matrix((idx1, offs1), n, n_partitions=4)
                t_gram = time.perf_counter() - t1
                ref_c0 = oracle.compute_pca(s_c0, 2)
                t_all = time.perf_counter() - t1
                oracle.set_num_threads(prev)
                cpu_c0 = {"threads": 4, "similarity_seconds": t_gram, "total_seconds": t_all, "kind": "port",
                          "pc_max_abs_diff_vs_gpu": float(np.abs(
                              np.abs(ref_c0["components"]) - np.abs(c1)).max())}
            out["config0_csr_end_to_end"] = {
                "workload": "configs[0] stand-in: synthetic BRCA1-sized region, %d variants x %d samples, %d carriers, "
                            "host CSR through pcoa_accumulate_calls" % (x1.shape[0], n, idx1.size),
                "wall_ms": float(min(walls[1:])), "wall_ms_all": walls,
                "eig_method": {1: "lanczos", 2: "householder"}.get(tt1["eig_method"], "?"),
                "eigenvalues": [float(t) for t in l1], "cpu_local4": cpu_c0}
        if world == 1 and not args.no_extras:
            # PCIe-inclusive rates (host tiles through the staging path): noted, never `value`
            hv = min(v, resident, 131072)
            xh = x[:hv].cpu().numpy()
            xh8 = xh.astype(np.uint8)
            pcie = {}
            xhb = pkg("ingest").pack_bits(xh8)
            for name, arr, fn in (("host_fp32", xh, eng.accumulate_dense), ("host_uint8", xh8, eng.accumulate_dense_u8),
                                  ("host_bits", xhb, eng.accumulate_bits)):
                eng.reset(); fn(arr); eng.finalize(); eng.sync()
                t1 = time.perf_counter()
                eng.reset(); fn(arr); eng.finalize(); eng.sync()
                pcie[name + "_variants_per_s"] = hv / (time.perf_counter() - t1)
            pcie["note"] = ("%d variants from pageable host memory, H2D included (pcoa_accumulate_dense_f32 / _u8 / "
                            "pcoa_accumulate_bits with is_device_ptr=0)" % hv)
            out["pcie_inclusive"] = pcie
            del xh, xh8, xhb
        if world == 1 and not args.no_cpu_baseline:
            base, s_ref, sample = cpu_baseline(x[:min(v, resident)], n)
            out["cpu_baseline"] = base
            # the CPU sample doubles as an end-of-run parity check of the measured path
            eng.reset()
            eng.accumulate_dense(x[:sample])
            out["parity_vs_cpu_sample"] = bool(np.array_equal(eng.gram(), s_ref))
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_extras and args.config3_samples > 0:
            # BASELINE configs[3] at FULL size on this one GPU, by wall-clock (the reference cannot hold N > 46,340 at all,
            # VariantsPca.scala:176-177, :185).  Last of the extras: the resident fp32 batches are released first (S alone is 40 GB).
            x1 = None   # (a view of the resident batches)
            del x, x_store
            torch.cuda.empty_cache()
            try:
                out["config3_one_gpu"] = config3_one_gpu(P, synth, torch, dev, local_rank, args.config3_samples, args.config3_variants)
            except Exception as exc:  # noqa: BLE001 -- an extra must never take the headline line with it (e.g. HBM held by another process)
                out["config3_one_gpu"] = {"skipped": "failed: %s: %s" % (type(exc).__name__, exc)}
        print(json.dumps(out), flush=True)
    if native is not None:
        native.close()
    if world > 1:
        td.barrier()
        td.destroy_process_group()
    eng.close()
    return 0


def sustained_leg(eng, step, variants_per_step, est_step_s, seconds):
    """The same leg over a timed region of >= `seconds`: a 20-step job pays its pipeline fill, its last exposed contraction and the
    finalize out of ~25 ms; a stream of batches does not (tools/soak_pipeline.py: 20,000 steps per format, exact)."""
    reps = max(int(seconds / max(est_step_s, 1e-6)) + 1, 20)
    eng.reset(); eng.reset_timings(); eng.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    eng.finalize(); eng.sync()
    dt = time.perf_counter() - t0
    return {"steps": reps, "seconds": dt, "value": variants_per_step * reps / dt, "unit": "variants/s", "ms_per_step": 1e3 * dt / reps}


def config3_one_gpu(P, synth, torch, dev, local_rank, n, v):
    """BASELINE configs[3]: synthetic 100,000 samples x 10^6 variants (seed 1004), Gram + eig on ONE GPU, timed as WALL from the
    first accumulate call to the finalized S, then pcoa_compute(2).  The genotypes are generated on the device inside the
    timed region (the fp32 input would be 400 GB; pcoa_accumulate_synthetic writes the k-bits operand directly); the model's
    thresholds (20 MB of parameters) are made on the host before it.  No CPU oracle can hold this job: the checks are the ones of
    tools/config4_biobank.py -- the top-left 2504 x 2504 block against an INDEPENDENT N = 2504 engine fed the same variants
    restricted to those samples, a far off-diagonal block against its mirror, the diagonal dominating its rows -- and the
    engine's own on-device residual test of the eigenpairs; tests/test_gpu_baseline_sizes.py holds blocks of S at this N to the
    CPU oracle at 65,536 variants and runs this job at full size."""
    seed = 1004
    now = time.perf_counter
    free0, total = torch.cuda.mem_get_info(dev)
    need = 4.0 * n * n + 16e9
    if free0 < need:
        return {"skipped": "needs %.0f GB of free HBM (S = %.0f GB int32 + operand buffers + workspaces), %.0f GB are free"
                           % (need / 1e9, 4.0 * n * n / 1e9, free0 / 1e9)}
    offs = synth.pop_offsets(n)
    n_small = min(2504, int(offs[1]))
    t = now()
    thr = synth.thresholds(seed, 0, v)
    t_thr = now() - t
    t = now()
    eng = P.PcoaEngine(n, device=local_rank)
    t_create = now() - t
    t = now()
    eng.reserve(1 << 20, 2)
    t_reserve = now() - t
    t = now()
    eng.accumulate_synthetic(seed, offs, thr[:4096], 0); eng.finalize(); eng.compute(2); eng.reset(); eng.reset_timings(); eng.sync()
    t_warm = now() - t
    chunk = 1 << 17
    t0 = now()
    for v0 in range(0, v, chunk):
        eng.accumulate_synthetic(seed, offs, thr[v0:v0 + chunk], v0)
    t_queued = now() - t0
    eng.finalize()
    eng.sync()
    wall = now() - t0
    tim = eng.timings()
    free1, _ = torch.cuda.mem_get_info(dev)
    t1 = now()
    comps, lam, nz = eng.compute(2)
    t_pcoa = now() - t1
    tim1 = eng.timings()
    t1 = now()
    comps, lam, nz = eng.compute(2)
    t_pcoa2 = now() - t1
    tim2 = eng.timings()
    # ---- checks (outside the timed region)
    with P.PcoaEngine(n_small, device=local_rank) as small:
        offs_small = np.array([0, n_small], dtype=np.int32)
        thr0 = np.ascontiguousarray(thr[:, :1])
        for v0 in range(0, v, chunk):
            small.accumulate_synthetic(seed, offs_small, thr0[v0:v0 + chunk], v0)
        ok_block = bool(np.array_equal(eng.gram_block(0, 0, n_small, n_small), small.gram()))
    a = eng.gram_block(10, n - 300, 200, 256)
    b = eng.gram_block(n - 300, 10, 256, 200)
    ok_mirror = bool(np.array_equal(a, b.T)) and int(a.sum()) > 0
    dg = eng.gram_block(n - 64, n - 64, 64, 64)
    ok_diag = bool(np.array_equal(dg, dg.T)) and bool((np.diag(dg) >= dg.max(axis=1)).all())
    # the eigensolver's passes over S in the second pcoa_compute: every Lanczos step and every true-residual check is one
    # mat-vec over the upper triangle (2 N^2 bytes), the row sums one more pass of the same bytes; the Lanczos time also holds
    # the re-orthogonalisation kernels, so the rate is a lower bound of the mat-vec's own
    passes = int(tim2["lanczos_steps"]) + 2
    lz_s = (tim2["lanczos_seconds"] - tim1["lanczos_seconds"])
    mv_s = lz_s / passes if (passes > 0 and lz_s > 0 and tim2["matvec_form"] == 1) else None
    gk = tim["gram_kernel_seconds"]
    issued = 2.0 * v * n * n / gk / 1e12 * issued_fraction(n, 256, diag=w4_diag_share()) if gk > 0 else 0.0
    res = {"workload": "configs[3]: synthetic %d samples x %d variants (seed %d), 1x MI355X (Gram + eig on one GPU), generated on "
                       "the device inside the timed region" % (n, v, seed),
           "gram_wall_s": wall, "variants_per_s_wall": v / wall, "pcoa_wall_s": min(t_pcoa, t_pcoa2), "pcoa_wall_s_all": [t_pcoa, t_pcoa2],
           "gram_plus_pcoa_wall_s": wall + min(t_pcoa, t_pcoa2),
           "wall_breakdown_s": {"contraction_kernels": gk, "generation_kernels": tim["synth_seconds"], "pre_pass_kernels": tim["pack_seconds"],
                                "finalize_kernels": tim["finalize_seconds"], "calls_returned_after": t_queued,
                                "unaccounted": wall - gk - tim["synth_seconds"] - tim["pack_seconds"] - tim["finalize_seconds"]},
           "outside_the_timed_region_s": {"thresholds_on_host": t_thr, "create_engine_40GB_S": t_create, "reserve": t_reserve,
                                          "first_use_warm_up": t_warm},
           "contraction": {"bound": "mfma", "kernel": "gram_kbits_w4_kernel<4, 0, 69> (banded split-K)", "launches": int(tim["gram_kernel_launches"]),
                           "achieved": issued, "peak": PEAK_FP4_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": issued / PEAK_FP4_MFMA_TFLOPS,
                           "algorithmic_pflops": 2.0 * v * n * n / gk / 1e15 if gk > 0 else None,
                           "convention": "issued matrix-core work (upper-triangular 256 x 256 tiles)"},
           "matvec": None if mv_s is None else {"bound": "hbm", "seconds_per_pass_incl_reorthogonalisation": mv_s, "passes": passes, "bytes": 2.0 * n * n,
                                                "achieved": 2.0 * n * n / mv_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                                "frac": 2.0 * n * n / mv_s / 1e9 / PEAK_HBM_GBS,
                                                "kernel": "symv_sym_tiles_kernel (upper-triangular 1024 x 1024 tiles of S, 2 N^2 bytes)"},
           "pcoa_method": {1: "lanczos (verified residual)", 2: "householder"}.get(tim2["eig_method"], "?"),
           "lanczos_steps": int(tim2["lanczos_steps"]), "matvec_form": int(tim2["matvec_form"]),
           "eigenvalues": [float(t) for t in lam], "nonzero_rows": int(nz),
           "unit_norm": [float(np.linalg.norm(comps[:, c])) for c in range(2)], "orthogonality": float(abs(comps[:, 0] @ comps[:, 1])),
           "hbm_in_use_gb": (total - free1) / 1e9, "hbm_total_gb": total / 1e9,
           "check_block_vs_independent_engine": ok_block, "check_mirror": ok_mirror, "check_diagonal": ok_diag}
    eng.close()
    return res


def config2_one_gpu_bits(P, synth, torch, dev, local_rank, n, v, operand):
    """BASELINE configs[2] (2,504 samples x 40 M variants, seed 1003) as ONE job on one GPU through the bit-packed boundary:
    the cohort is generated on the device chunk by chunk and kept resident as bitsets (313 B per variant), then
    accumulate + finalize + computePca are timed.  Reported beside the headline, never as `value`."""
    seed, chunk = 1003, 1 << 18
    offs = synth.pop_offsets(n)
    words = (n + 31) // 32
    t0 = time.perf_counter()
    bits = torch.empty((v, words), dtype=torch.int32, device=dev)
    tile = torch.empty((chunk, n), dtype=torch.float32, device=dev)
    shifts = torch.arange(32, device=dev, dtype=torch.int32)
    with P.PcoaEngine(n, device=local_rank) as gen:
        for v0 in range(0, v, chunk):
            c = min(chunk, v - v0)
            gen.synth_fill(1003, offs, synth.thresholds(seed, v0, c), v0, tile.data_ptr(), n)
            for r0 in range(0, c, 1 << 16):
                r1 = min(c, r0 + (1 << 16))
                xb = torch.nn.functional.pad(tile[r0:r1] > 0, (0, words * 32 - n)).view(r1 - r0, words, 32)
                bits[v0 + r0:v0 + r1] = (xb.to(torch.int32) << shifts).sum(dim=2, dtype=torch.int32)
    del tile
    torch.cuda.synchronize(dev)
    t_gen = time.perf_counter() - t0
    with P.PcoaEngine(n, device=local_rank, operand=operand) as e2:
        e2.reserve(1 << 20, 2)
        e2.accumulate_bits(bits[:1 << 20]); e2.finalize(); e2.compute(2)       # warm-up
        e2.reset(); e2.reset_timings(); e2.sync()
        t1 = time.perf_counter()
        e2.accumulate_bits(bits)
        e2.finalize()
        e2.sync()
        t_gram = time.perf_counter() - t1
        comps, lam, nz = e2.compute(2)
        t_all = time.perf_counter() - t1
        tt = e2.timings()
        smax = int(e2.gram().max())
    del bits
    return {"workload": "configs[2]: %d samples x %d variants (seed %d) as carrier bitsets, resident on one GPU (%.1f GB), one job"
                        % (n, v, seed, 4.0 * words * v / 1e9),
            "gram_wall_s": t_gram, "gram_variants_per_s": v / t_gram, "gram_plus_pcoa_wall_s": t_all,
            "pre_pass_s": tt["pack_seconds"], "contraction_s": tt["gram_kernel_seconds"],
            "contraction_launches": int(tt["gram_kernel_launches"]), "largest_entry_of_S": smax,
"eigenvalues": [float(t) for t in lam],
            "generation_s_outside_the_timed_region": t_gen,
            "note": "parity of this job (shard sum, oracle blocks, eigenpairs): tests/test_gpu_baseline_sizes.py::"
                    "test_config2_full_size_one_cohort_on_one_gpu_bitset_boundary"}


def csr_boundary(P, torch, dev, local_rank, n, x1, s_dense_steps, steps, operand):
    """The reference's own seam at scale: the configs[1] batch as the carrier lists an RDD[Seq[Int]]
    partition holds (getCallsRdd, VariantsPca.scala:153-168) through pcoa_accumulate_calls_ex, whole job = H2D + scatter into
    the operand + contraction + finalize.  One leg per place the arrays can live."""
    v = int(x1.shape[0])
    counts = torch.zeros(v, dtype=torch.int64, device=dev)
    cols = []
    for r0 in range(0, v, 1 << 17):
        nzr = (x1[r0:r0 + (1 << 17)] != 0)
        counts[r0:r0 + nzr.shape[0]] = nzr.sum(dim=1)
        cols.append(nzr.nonzero()[:, 1].to(torch.int32))
    idx_dev = torch.cat(cols)
    del cols, nzr
    offs_dev = torch.zeros(v + 1, dtype=torch.int64, device=dev)
    offs_dev[1:] = torch.cumsum(counts, 0)
    nnz = int(offs_dev[-1])
    bytes_per_variant = (4.0 * nnz + 8.0 * (v + 1)) / v
    pcie_bound = 63e9 / bytes_per_variant
    idx_cpu, offs_cpu = idx_dev.cpu(), offs_dev.cpu()
    idx_pin, offs_pin = idx_cpu.pin_memory(), offs_cpu.pin_memory()
    half = v // 2 // 128 * 128
    # what the link of THIS box delivers: the same pinned array through one plain async copy (best of 3)
    land = torch.empty_like(idx_dev)
    link = 0.0
    for _ in range(3):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        land.copy_(idx_pin, non_blocking=True)
        torch.cuda.synchronize(dev)
        link = max(link, 4.0 * nnz / (time.perf_counter() - t0))
    del land
    link_bound = link / bytes_per_variant
    res = {"workload": "configs[1] batch 0 as carrier lists: %d variants, %d carriers (%.1f per variant), %.0f B per variant"
                       % (v, nnz, nnz / float(v), bytes_per_variant),
           "pcie_bound_variants_per_s": pcie_bound, "pcie_gbs_assumed": 63.0,
           "link_gbs_measured": link / 1e9, "link_bound_variants_per_s": link_bound,
           "link_note": "link_gbs_measured = one hipMemcpyAsync of the pinned index array (%.2f GB) on this box, best of 3; "
                        "frac_of_measured_link = a leg's variants/s against that" % (4e-9 * nnz)}
    with P.PcoaEngine(n, device=local_rank, operand=operand) as e:
        e.reserve(v, 0)
        legs = [("pageable", idx_cpu, offs_cpu, False, 1), ("pinned", idx_pin, offs_pin, False, 1),
                ("pinned_async_two_halves", idx_pin, offs_pin, True, 2), ("device", idx_dev, offs_dev, False, 1)]
        for name, ti, to, asyn, parts in legs:
            for timed in (False, True):
                e.reset(); e.reset_timings(); e.sync()
                t0 = time.perf_counter()
                if parts == 1:
                    e.accumulate_calls_tensors(ti, to, asynchronous=asyn)
                else:   # what a host does that builds batch k+1 while batch k travels: two calls, no wait in between
                    oh = to[half:] - to[half]
                    oh = oh.pin_memory() if to.is_pinned() else oh
                    t0 = time.perf_counter()
                    e.accumulate_calls_tensors(ti[:int(to[half])], to[:half + 1], asynchronous=asyn)
                    e.accumulate_calls_tensors(ti[int(to[half]):], oh, asynchronous=asyn)
                t_call = time.perf_counter() - t0
                e.finalize(); e.sync()
                dt = time.perf_counter() - t0
            tt = e.timings()
            same = bool(np.array_equal(e.gram() * steps, s_dense_steps))
            res[name] = {"variants_per_s": v / dt, "frac_of_pcie_bound": (v / dt) / pcie_bound,
                         "frac_of_measured_link": (v / dt) / link_bound, "seconds": dt,
                         "seconds_until_the_calls_returned": t_call,
                         "scatter_kernel_ms": 1e3 * tt["densify_seconds"], "contraction_ms": 1e3 * tt["gram_kernel_seconds"],
                         "host_staging_copy_s": tt["csr_stage_seconds"], "host_wait_for_device_check_s": tt["csr_wait_seconds"],
                         "chunks": int(tt["csr_fast_chunks"]), "same_gram_as_dense_input": same}
        # r06 (VERDICT r05 Weak 7): at this density (mean list longer than N / 32 entries) a host sends the rows as carrier
        # BITSETS -- N / 8 bytes per variant instead of 4 per carrier: the compiled host's parser threads pack them for dense
        # blocks (--carrier-format auto), the Scala host per batch.  The same batch as bitsets in page-locked host memory
        # through pcoa_accumulate_bits, H2D included; the host-side packing is not in this leg (a few threads pack 10^6 rows in
        # ~40 ms beside the transfer: profiles/r07*_carrier_format*.txt)
        words = (n + 31) // 32
        bits_dev = torch.zeros((v, words), dtype=torch.int32, device=dev)
        rows_of = torch.repeat_interleave(torch.arange(v, device=dev), counts)
        flat = rows_of * words + (idx_dev.to(torch.int64) >> 5)
        vals = (torch.ones_like(flat) << (idx_dev.to(torch.int64) & 31))
        acc64 = torch.zeros(v * words, dtype=torch.int64, device=dev)
        acc64.index_add_(0, flat, vals)            # distinct bits of a word: a sum is an OR
        bits_dev.view(-1).copy_(torch.where(acc64 >= 2 ** 31, acc64 - 2 ** 32, acc64).to(torch.int32))
        del rows_of, flat, vals, acc64
        bits_pin = bits_dev.cpu().pin_memory()
        del bits_dev
        ptr = ctypes.c_void_p(bits_pin.data_ptr())
        for timed in (False, True):
            e.reset(); e.reset_timings(); e.sync()
            t0 = time.perf_counter()
            e._check(e._lib.pcoa_accumulate_bits(e._ctx, ptr, v, words, 0))
            e.finalize(); e.sync()
            dt = time.perf_counter() - t0
        tt = e.timings()
        bits_bound = 63e9 / (4.0 * words)
        res["auto_bits"] = {"variants_per_s": v / dt, "seconds": dt, "bytes_per_variant": 4 * words,
                            "frac_of_pcie_bound": (v / dt) / bits_bound, "frac_of_measured_link": (v / dt) / (link / (4.0 * words)),
                            "transpose_ms": 1e3 * tt["pack_seconds"], "contraction_ms": 1e3 * tt["gram_kernel_seconds"],
                            "same_gram_as_dense_input": bool(np.array_equal(e.gram() * steps, s_dense_steps)),
                            "what": "the batch as carrier bitsets in page-locked host memory (what --carrier-format auto / the Scala "
                                    "host send for a block this dense) through pcoa_accumulate_bits"}
        del bits_pin
    res["note"] = ("device-validated path (range + repeats checked by the scatter kernel; a call reaches S only after its check): "
                   "pageable arrays are copied into pinned staging by host threads beside the H2D of the previous chunk; "
                   "PCOA_CSR_LEGACY=1 gives the r03 path (host-serial validation, one staging buffer) for comparison")
    return res


def plink_bed_boundary(P, torch, dev, local_rank, n, x1, s_dense_steps, steps, operand):
    """The configs[1] batch as the rows of a variant-major PLINK 1 .bed (2 bits per genotype, ceil(N / 4) bytes per variant;
    carriers heterozygous, the rest homozygous A2 = reference) through pcoa_accumulate_plink_bed: the boundary a cohort stored
    as a PLINK fileset crosses (the compiled host streams files this way), whole job = H2D + device decode + transpose +
    contraction + finalize."""
    v = int(x1.shape[0])
    bpv = (n + 3) // 4
    raw = torch.empty((v, bpv), dtype=torch.uint8, device=dev)
    sh = torch.tensor([0, 2, 4, 6], dtype=torch.uint8, device=dev)
    for r0 in range(0, v, 1 << 16):
        c = torch.nn.functional.pad(x1[r0:r0 + (1 << 16)] > 0, (0, bpv * 4 - n))
        codes = (3 - c.to(torch.uint8)).view(-1, bpv, 4)            # carrier -> 10, non-carrier -> 11
        raw[r0:r0 + codes.shape[0]] = (codes << sh).sum(dim=2).to(torch.uint8)
    del c, codes
    block = 131072
    raw_pin = raw.cpu().pin_memory()
    res = {"workload": "configs[1] batch 0 as PLINK .bed rows: %d variants x %d samples, %d B per variant" % (v, n, bpv),
           "link_bound_variants_per_s_at_63_gbs": 63e9 / bpv}
    with P.PcoaEngine(n, device=local_rank, operand=operand) as e:
        e.reserve(v, 0)
        for name in ("device_rows", "pinned_rows_queued"):
            for timed in (False, True):
                e.reset(); e.reset_timings(); e.sync()
                t0 = time.perf_counter()
                if name == "device_rows":
                    e.accumulate_plink_bed(raw)
                else:   # page-locked rows, queued block by block (PCOA_BED_HOST_ASYNC): what the streaming host's feed does
                    for r0 in range(0, v, block):
                        e.accumulate_plink_bed(raw_pin[r0:r0 + block], asynchronous=True)
                e.finalize(); e.sync()
                dt = time.perf_counter() - t0
            tt = e.timings()
            res[name] = {"variants_per_s": v / dt, "seconds": dt, "gb_per_s_of_rows": v * bpv / dt / 1e9,
                         "decode_ms": 1e3 * tt["densify_seconds"], "transpose_ms": 1e3 * tt["pack_seconds"],
                         "contraction_ms": 1e3 * tt["gram_kernel_seconds"],
                         "same_gram_as_dense_input": bool(np.array_equal(e.gram() * steps, s_dense_steps))}
    res["note"] = ("pinned_rows_queued: the rows lie in page-locked memory already (no file, no host copy) -- file -> reduced S "
                   "with the compiled host: profiles/r06h_plink_stream_final.txt")
    return res


def issued_fraction(n, bm, idle_diag=True, diag=None):
    """Share of the 2*V*N^2 of SURVEY 8(d) the Gram kernels issue on the matrix cores: upper-triangular tiles only,
    and in a diagonal tile the waves that lie wholly below the diagonal skip their MFMAs (2 of 8 waves of a
    256 x 256 tile: 0.75 of it; 1 of 4 waves of a 128 x 128 tile of the fp32 kernel).  diag: the share of a diagonal
    tile's MFMAs that is issued, where the kernel is more selective (gram_kbits_w4_kernel, r04: 36 of 64 MFMA tiles)."""
    t = (n + bm - 1) // bm
    if diag is None:
        diag = 0.75 if idle_diag else 1.0
    return (t * (t - 1) / 2.0 + t * diag) / float(t * t)


def w4_diag_share():
    """gram_kbits_w4_kernel on a diagonal 256 x 256 tile: with wave roles (default) the two blocks on the diagonal issue 10 of
    their 16 MFMA tiles and the block above it all 16 -- 36 of 64; PCOA_KBITS_W4_DIAG=0 (r04a form): 48 of 64."""
    return 0.75 if os.environ.get("PCOA_KBITS_W4_DIAG", "-1") == "0" else 36.0 / 64.0


def useful_fraction(n):
    """Share of 2*V*N^2 that S = X^T X needs at all: the upper triangle with its diagonal, V*N*(N+1)."""
    return (n + 1.0) / (2.0 * n)


if __name__ == "__main__":
    sys.exit(main())
