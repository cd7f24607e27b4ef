OUT=gpurun_out/$1; mkdir -p $OUT
for round in 1 2; do
for w in 1 2 0; do
  PCOA_KBITS_W4=$w timeout 600 python bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --pcoa-reps 1 > $OUT/bench_w$w.json 2>> $OUT/bench.err
  python - <<PY | tee -a $OUT/summary.txt
import json
d=json.load(open("$OUT/bench_w$w.json"))
print("W4=$w: value %.1f M/s, ms/step %.3f (gram %.3f, pack %.3f), sustained %.1f | standalone contraction %.3f ms pre-pass %.3f | bits %.1f M/s (gram %.3f) | u8 %.1f M/s (gram %.3f) | config2 %.1f M/s" % (d['value']/1e6, d['ms_per_step'], d['gram_ms_per_step'], d['pack_ms_per_step'], d['sustained']['value']/1e6, d['roofline_standalone']['contraction']['avg_launch_ms'], d['roofline_standalone']['pre_pass']['avg_launch_ms'], d['alt_input_bits']['value']/1e6, d['alt_input_bits']['gram_ms_per_step'], d['alt_input_u8']['value']/1e6, d['alt_input_u8']['gram_ms_per_step'], d['config2_one_gpu_bits']['gram_variants_per_s']/1e6))
PY
done
done
