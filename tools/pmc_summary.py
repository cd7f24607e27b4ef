#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (per-launch averages).
usage: tools/pmc_summary.py <gpurun_out/tag>"""
import csv, glob, json, os, re, sys
from collections import defaultdict

out = sys.argv[1]
res = {}
for sub in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(sub):
        continue
    for f in glob.glob(os.path.join(sub, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            m = re.search(r"(gram_kbits_kernel|pack_kbits_ring_kernel|pack_kbits_kernel|pack_u8x8_kbits_kernel|transpose_bits_kbits_kernel|gram_packed_kernel|gram_i8_kernel|gram_f32_kernel|pack_u8x8_fp4_kernel|expand_bits_fp4_kernel|pack_f32_i8_kernel|pack_u8_i8_kernel|pack_fp4_kernel|"
                          r"tridiag_update_kernel|symv_kernel|symv_sym_tiles_kernel|rowsums_sym_tiles_kernel|symv_centered_kernel|row_sums_kernel|"
                          r"densify_csr_kbits_lds_kernel|densify_csr_kbits_kernel|plink_bed_to_bits_kernel|gram_kbits_w4_kernel)", k)
            if not m:
                continue
            name = m.group(1)
            if name in ("gram_i8_kernel", "gram_packed_kernel"):   # template: <FMT, ...>, FMT 1 = MX-FP4 operands
                name = "gram_packed_kernel_fp4" if re.search(r"gram_(i8|packed)_kernel<1", k) else "gram_packed_kernel_i8"
            acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):   # KiB; gfx950 tallies wide reads at half (MI355X_MICROARCH.md)
                acc[name][row["Counter_Name"] + "_bytes"] += float(row["Counter_Value"]) * 1024.0 * (2.0 if row["Counter_Name"] == "FETCH_SIZE" else 1.0)
            cnt[name].add(row.get("Dispatch_Id"))
        for name, d in acc.items():
            n = max(len(cnt[name]), 1)
            for c, v in d.items():
                res.setdefault(name, {})[c] = v / n
            res[name].setdefault("launches_seen", {})[os.path.basename(sub)] = n
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
