set +e
OUT=gpurun_out/r06a; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "carrier or csr or repeat or calls or multiplicity" > $OUT/tests_csr.log 2>&1
echo "tests exit $?" > $OUT/summary.txt; tail -3 $OUT/tests_csr.log >> $OUT/summary.txt
for g in 1 0 1 0; do
  echo "== PCOA_CSR_GLOBAL_ATOMICS=$g" >> $OUT/summary.txt
  PCOA_CSR_GLOBAL_ATOMICS=$g timeout 300 python tools/csr_probe.py 1000000 >> $OUT/summary.txt 2>&1
done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" >> $OUT/summary.txt
python - >> $OUT/summary.txt <<'PY'
import json
d=json.load(open('gpurun_out/r06a/bench.json'))
print("value %.1f M/s ms/step %.3f frac %.3f sustained %.1f" % (d['value']/1e6, d['ms_per_step'], d['roofline']['frac'], d['sustained']['value']/1e6))
print(json.dumps(d['csr_boundary'], indent=1))
print("standalone contraction", d['roofline_standalone']['contraction']['avg_launch_ms'], d['roofline_standalone']['contraction']['frac'])
print("bits", d['alt_input_bits']['value']/1e6, "u8", d['alt_input_u8']['value']/1e6)
PY
cat $OUT/summary.txt
