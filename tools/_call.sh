set +e
bash tools/gpu_round.sh r06z smoke tests bench prof pmclive
OUT=gpurun_out/r06z
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err
echo "driver-command bench exit $?" | tee -a $OUT/summary.txt
python - <<'PY' | tee -a gpurun_out/r06z/summary.txt
import json
d=json.load(open('gpurun_out/r06z/bench_driver_command.json'))
print("value %.1f M/s ms/step %.3f frac %.3f traffic %s sustained %.1f" % (d['value']/1e6, d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['sustained']['value']/1e6))
print("standalone contraction", d['roofline_standalone']['contraction']['avg_launch_ms'], d['roofline_standalone']['contraction']['frac'])
print("bits", d['alt_input_bits']['value']/1e6, "u8", d['alt_input_u8']['value']/1e6)
print("csr", {k:(round(v['variants_per_s']/1e6,1), round(v.get('frac_of_pcie_bound',0),3)) for k,v in d['csr_boundary'].items() if isinstance(v,dict)})
print("plink", json.dumps(d.get('plink_bed_boundary'))[:900])
print("pcie", d['pcie_inclusive'])
print("config2", d['config2_one_gpu_bits']['gram_wall_s'], d['config2_one_gpu_bits']['gram_variants_per_s']/1e6)
print("pcoa", d['pcoa_wall_ms'])
PY
timeout 900 python tools/config4_biobank.py --samples 250000 --variants 65536 > $OUT/biobank_250k.json 2> $OUT/biobank_250k.err
tail -1 $OUT/biobank_250k.json | cut -c1-900 | tee -a $OUT/summary.txt
