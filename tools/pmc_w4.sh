#!/bin/bash
# PMC passes over tools/exp_w4 (one rocprofv3 run per counter set; --pmc only, never with a trace domain).
# usage: tools/pmc_w4.sh <tag> "<exp_w4 args>"
TAG=$1; ARGS=$2
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_REQ_sum TCC_TAG_STALL_sum TCC_READ_sum" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$i -o pmc -- $OLDPWD/tools/exp_w4 $ARGS > $OUT/pmc_$i.log 2>&1 )
  echo "pmc set $i exit $?" | tee -a $OUT/summary.txt
done
python3 - <<PY | tee -a $OUT/summary.txt
import csv, glob, json, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    per = defaultdict(lambda: defaultdict(float))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gram_kbits" not in k: continue
        m = re.search(r"gram_kbits\w*<[^>]*>", k)
        name = (m.group(0) if m else k[:60]) + " grid " + row.get("Grid_Size", "?")
        per[(name, row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
    for (name, _), d in per.items():
        for c, v in d.items(): acc[name][c].append(v)
res = {n: {c: sum(v) / len(v) for c, v in d.items()} for n, d in acc.items()}
for n, d in sorted(res.items()):
    print(n)
    for c, v in sorted(d.items()): print("   %-28s %.4g" % (c, v))
    if "TCC_HIT_sum" in d: print("   L2 hit rate %.3f" % (d["TCC_HIT_sum"] / max(1, d["TCC_HIT_sum"] + d["TCC_MISS_sum"])))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d: print("   MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs) = %.3f" % (d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (d["GRBM_GUI_ACTIVE"] / 8)))
    if "SQ_INSTS_VALU" in d and "SQ_INSTS_MFMA" in d: print("   SQ_INSTS_VALU : SQ_INSTS_MFMA %.2f (SQ_INSTS_VALU counts the MFMAs too: %.2f other VALU per MFMA)" % (d["SQ_INSTS_VALU"] / d["SQ_INSTS_MFMA"], d["SQ_INSTS_VALU"] / d["SQ_INSTS_MFMA"] - 1))
    if "TCC_EA0_RDREQ_LEVEL_sum" in d and "TCC_EA0_RDREQ_sum" in d: print("   mean EA read latency %.0f TCC cycles" % (d["TCC_EA0_RDREQ_LEVEL_sum"] / max(1, d["TCC_EA0_RDREQ_sum"])))
json.dump(res, open("$OUT/pmc_summary.json", "w"), indent=1)
PY
find $OUT -name "*counter_collection*" -size +4M -delete
