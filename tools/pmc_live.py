#!/usr/bin/env python3
"""Writes <out>/gram_pmc_live.json from the `pmclive` passes of tools/gpu_round.sh: HBM bytes per 10^6 variants of the
k-bits pre-pass and contraction kernels, (a) in the co-resident pipeline's kernels and (b) in the kernels of the serial
order (PCOA_PIPELINE=0), from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --no-extras` ON THE TREE THE
CALL RAN ON.  The file carries that tree's source hash (_lib.source_hash); bench.py refuses a file of another tree.

Counters (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies the 128-B
requests of wide coalesced reads at 64 B, so reads are doubled: bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.
usage: tools/pmc_live.py <gpurun_out/tag>"""
import csv, glob, json, os, re, sys, time

out = sys.argv[1]
GRAM = r"(gram_kbits_w4_kernel|gram_kbits_kernel)"
PACK = r"(pack_kbits_ring_kernel|pack_kbits_kernel)"


def totals(sub):
    """counter totals over every dispatch of the process, per kernel family"""
    acc = {"gram": {}, "pack": {}}
    names = {"gram": set(), "pack": set()}
    n_disp = {"gram": set(), "pack": set()}
    for f in glob.glob(os.path.join(sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            fam = "gram" if re.search(GRAM, k) else "pack" if re.search(PACK, k) else None
            if fam is None:
                continue
            acc[fam][row["Counter_Name"]] = acc[fam].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            names[fam].add(re.search(GRAM if fam == "gram" else PACK, k).group(1))
            n_disp[fam].add(row.get("Dispatch_Id"))
    return acc, {k: sorted(v) for k, v in names.items()}, {k: len(v) for k, v in n_disp.items()}


res = {"made": time.strftime("%Y-%m-%d"), "made_by": "tools/gpu_round.sh %s pmclive" % os.path.basename(out.rstrip("/")),
       "convention": "bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (KiB counters; gfx950 tallies wide reads at half), summed "
                     "over every dispatch of the kernel family in the process / the variants bench.py put through the engine "
                     "(pmc_manifest), x 10^6; rocprofv3 serialises kernels, so each ran without the other beside it"}
for key, tag in (("kbits", "pipe"), ("kbits_standalone", "serial")):
    rec, man = {}, None
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        sub = os.path.join(out, "pmclive_%s_%s" % (tag, c))
        try:
            man = json.load(open(sub + ".json"))["pmc_manifest"]
        except Exception as exc:  # noqa: BLE001
            print("no manifest for %s: %s" % (sub, exc)); continue
        acc, names, nd = totals(sub)
        v = float(man["fp32_variants_through_the_engine"])
        for fam in ("gram", "pack"):
            if c in acc[fam]:
                rec["%s_%s_bytes_per_mvariants" % (fam, "read" if c == "FETCH_SIZE" else "write")] = \
                    acc[fam][c] * 1024.0 * (2.0 if c == "FETCH_SIZE" else 1.0) / v * 1e6
                rec["%s_kernels" % fam] = names[fam]
                rec["%s_dispatches" % fam] = nd[fam]
        res["source_hash"] = man["source_hash"]
        rec["variants"] = v
    for fam in ("gram", "pack"):
        if "%s_read_bytes_per_mvariants" % fam in rec and "%s_write_bytes_per_mvariants" % fam in rec:
            rec["%s_hbm_bytes_per_mvariants" % fam] = rec["%s_read_bytes_per_mvariants" % fam] + rec["%s_write_bytes_per_mvariants" % fam]
    res[key] = rec
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(out, "gram_pmc_live.json"), "w"), indent=1)
