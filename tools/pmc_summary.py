#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs for the Gram kernel (per-launch averages)."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
res = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "gram_" not in k:
                continue
            name = k.split("(")[0][-40:]
            acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[name].add(row.get("Dispatch_Id"))
        for name, d in acc.items():
            n = max(len(cnt[name]), 1)
            for c, v in d.items():
                res.setdefault(name, {})[c] = v / n
            res[name]["launches_seen"] = n
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
