set +e
OUT=gpurun_out/r06c; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/bed_probe.py 32 > $OUT/bed_probe.txt 2>&1
echo "== bed probe" > $OUT/summary.txt; grep -v amdgpu.ids $OUT/bed_probe.txt >> $OUT/summary.txt
for rnd in 1 2; do
for set in "X=0" "PCOA_KBITS_W4=2" "PCOA_BITS_PIPELINE=1 PCOA_KBITS_W4=2"; do
  echo "== [$set]" >> $OUT/summary.txt
  env $set timeout 300 python tools/alt_inputs_ab.py 20 2>&1 | grep -E "^pipeline|S equal" | head -3 >> $OUT/summary.txt
done
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof100k -o trace -- python $OLDPWD/tools/config4_biobank.py --samples 100000 --variants 131072 > $OLDPWD/$OUT/prof100k.json 2> $OLDPWD/$OUT/prof100k.err )
find $OUT/prof100k -name "*kernel_stats*" | head -1 | while read f; do head -16 "$f" | cut -c1-200; done >> $OUT/summary.txt
find $OUT/prof100k -name "*kernel_trace*" -size +8M -delete
cat $OUT/summary.txt
