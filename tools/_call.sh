set +e
OUT=gpurun_out/r06d; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x -k "upper_triangle or lanczos or implicit or strip or large_n or pcoa or compute or center or centr" > $OUT/tests.log 2>&1
echo "tests exit $?" > $OUT/summary.txt; tail -3 $OUT/tests.log >> $OUT/summary.txt
timeout 600 python tools/bed_probe.py 32 > $OUT/bed_probe.txt 2>&1
echo "== bed probe" >> $OUT/summary.txt; grep -v amdgpu.ids $OUT/bed_probe.txt >> $OUT/summary.txt
for lib in build/libpcoa_hip_old.so spark-examples_amd/libpcoa_hip.so build/libpcoa_hip_old.so spark-examples_amd/libpcoa_hip.so; do
  echo "== $lib N=100000" >> $OUT/summary.txt
  PCOA_LIB=$PWD/$lib timeout 600 python tools/config4_biobank.py --samples 100000 --variants 131072 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('pcoa_wall_s','lanczos_steps','matvec_upper_triangle_form_s','matvec_forms_max_rel_diff','eigenvalues','check_block_vs_independent_engine')})" >> $OUT/summary.txt 2>&1
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof100k -o trace -- python $OLDPWD/tools/config4_biobank.py --samples 100000 --variants 131072 > $OLDPWD/$OUT/prof100k.json 2> $OLDPWD/$OUT/prof100k.err )
find $OUT/prof100k -name "*kernel_stats*" | head -1 | while read f; do head -12 "$f" | cut -c1-220; done >> $OUT/summary.txt
find $OUT/prof100k -name "*kernel_trace*" -size +8M -delete
cat $OUT/summary.txt
