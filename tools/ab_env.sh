#!/bin/bash
# A/B of bench.py under different environments, two interleaved rounds: tools/ab_env.sh <tag> "ENV1=a ENV2=b;ENV1=c;..." [extra bench args]
OUT=gpurun_out/$1; mkdir -p $OUT
IFS=';' read -ra SETS <<< "$2"
for round in 1 2; do
  for set in "${SETS[@]}"; do
    env $set timeout 600 python bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --pcoa-reps 1 ${3:---no-extras} > $OUT/bench_ab.json 2>> $OUT/bench.err
    python - <<PY | tee -a $OUT/summary.txt
import json
d=json.load(open("$OUT/bench_ab.json"))
x=" | bits %.1f u8 %.1f config2 %.1f" % (d['alt_input_bits']['value']/1e6, d['alt_input_u8']['value']/1e6, d['config2_one_gpu_bits']['gram_variants_per_s']/1e6) if 'alt_input_bits' in d else ""
print("[%s] value %.1f M/s, ms/step %.3f (gram %.3f, pack %.3f), sustained %.1f%s" % ("$set", d['value']/1e6, d['ms_per_step'], d['gram_ms_per_step'], d['pack_ms_per_step'], d.get('sustained',{}).get('value',0)/1e6, x))
PY
  done
done
