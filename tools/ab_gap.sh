#!/bin/bash
# A/B of the fp32 step's launch-gap knobs and row pitch: tools/ab_gap.sh <tag>
OUT=gpurun_out/$1; mkdir -p $OUT
run() { # label, env, args
  env $2 timeout 600 python bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --pcoa-reps 1 --no-extras $3 > $OUT/b.json 2>> $OUT/bench.err
  python - <<PY | tee -a $OUT/summary.txt
import json
d=json.load(open("$OUT/b.json"))
print("[%s] value %.1f M/s, ms/step %.3f (gram %.3f, pack %.3f) ring frac %.3f" % ("$1", d['value']/1e6, d['ms_per_step'], d['gram_ms_per_step'], d['pack_ms_per_step'], d['roofline']['frac']))
PY
}
for round in 1 2; do
  run r03-forks "PCOA_FORK_LAZY=0 PCOA_RING_ALIGNED_WINDOWS=0" ""
  run lazy-only "PCOA_RING_ALIGNED_WINDOWS=0" ""
  run default "" ""
  run default+hs5 "PCOA_HEADSTART_US=5" ""
  run default+hs20 "PCOA_HEADSTART_US=20" ""
  run ld2528 "" "--ld 2528"
  run ld2512 "" "--ld 2512"
done
