#!/bin/bash
# Per-channel L2 -> fabric read requests of the ring pre-pass on "fast" and "slow" resident batches (VERDICT r05 Next 8):
# tools/placement_probe.py under rocprofv3 --pmc, one pass per counter set.  usage: tools/placement_pmc.sh <tag>
TAG=${1:-placement}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -E "TCC_EA0_RDREQ|TCC_EA0_RD_|TCC_BUBBLE|TCC_TAG_STALL|MALL|TCC_EA0_RDREQ_DRAM|UTCL2|TCP_UTCL1" | sort -u | head -60 > $OUT/counters_avail.txt
i=0
for set in "TCC_EA0_RDREQ" "TCC_EA0_RDREQ_DRAM TCC_TAG_STALL" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd /tmp && timeout 420 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$i -o pmc -- python $OLDPWD/tools/placement_probe.py 2 > $OUT/probe_$i.txt 2>&1 )
  echo "set $i ($set) exit $?" | tee -a $OUT/summary.txt
done
python3 - <<PY | tee -a $OUT/summary.txt
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    if not rows: continue
    print("==", f.split("/")[-3], "columns:", list(rows[0].keys()))
    per = collections.OrderedDict()
    for r in rows:
        if "pack_kbits_ring" not in r["Kernel_Name"]: continue
        per.setdefault(r["Dispatch_Id"], collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
    ids = list(per.keys())
    print("   ring pre-pass dispatches:", len(ids))
    for d in ids[:: max(1, len(ids) // 60)]:
        line = "   dispatch %s" % d
        for c, v in per[d].items():
            line += "  %s n=%d sum=%.4g min=%.4g max=%.4g max/mean=%.3f" % (c, len(v), sum(v), min(v), max(v), max(v) / (sum(v) / len(v)) if sum(v) else 0)
        print(line)
PY
find $OUT -name "*counter_collection*" -size +6M -delete
