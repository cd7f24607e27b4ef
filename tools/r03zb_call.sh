OUT=gpurun_out/r03zb; mkdir -p $OUT
export TMPDIR=/tmp
# 1. PMC passes of the driver's command (co-resident default)
bash tools/gpu_round.sh r03zb pmc
# 2. kernel trace with timestamps: gaps between consecutive pre-passes
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$OUT/prof -o trace -- python $OLDPWD/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --pcoa-reps 1 > $OLDPWD/$OUT/prof_bench.json 2> $OLDPWD/$OUT/prof.err; echo "prof exit $?" | tee -a $OLDPWD/$OUT/summary.txt )
python - <<'PY' | tee -a gpurun_out/r03zb/summary.txt
import csv, glob
f = glob.glob("gpurun_out/r03zb/prof/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    k = "ring" if "pack_kbits_ring" in n else "gram" if "gram_kbits" in n else "pack" if "pack_kbits_kernel" in n else "delay" if "delay_kernel" in n else None
    if k: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
ev.sort()
t0 = ev[0][0]
rings = [e for e in ev if e[2] == "ring"]
print("ring launches", len(rings), "gram", sum(1 for e in ev if e[2] == "gram"))
for i in range(max(1, len(rings) - 8), len(rings)):
    a, b = rings[i - 1], rings[i]
    grams = [g for g in ev if g[2] == "gram" and a[1] - 3000000 < g[0] < b[1]]
    gs = ", ".join("gram %.3f..%.3f" % ((g[0] - a[0]) / 1e6, (g[1] - a[0]) / 1e6) for g in grams)
    print("ring %2d: dur %.3f ms, gap to next start %.3f ms (period %.3f); %s" % (i - 1, (a[1] - a[0]) / 1e6, (b[0] - a[1]) / 1e6, (b[0] - a[0]) / 1e6, gs))
PY
find $OUT/prof -name "*kernel_trace*" -size +6M -delete
# 3. the fp32-MFMA kernel: is the matrix pipe busy?
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA --output-format csv -d $OLDPWD/$OUT/pmc_f32sq -o pmc -- python $OLDPWD/bench.py --gpus 1 --steps 2 --warmup 1 --gram-kernel f32 --variants 131072 --no-cpu-baseline --no-extras --pcoa-reps 1 > /dev/null 2> $OLDPWD/$OUT/pmc_f32sq.err; echo "pmc f32 sq exit $?" | tee -a $OLDPWD/$OUT/summary.txt )
python tools/pmc_summary.py $OUT | tee -a $OUT/summary.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_f32 -o trace -- python $OLDPWD/bench.py --gpus 1 --steps 2 --warmup 1 --gram-kernel f32 --variants 131072 --no-cpu-baseline --no-extras --pcoa-reps 1 > /dev/null 2> $OLDPWD/$OUT/prof_f32.err )
find $OUT/prof_f32 -name "*kernel_stats*" | head -1 | while read f; do head -4 "$f" | cut -c1-200; done | tee -a $OUT/summary.txt
find $OUT -name "*counter_collection*" -size +4M -delete; find $OUT -name "*kernel_trace*" -size +6M -delete
