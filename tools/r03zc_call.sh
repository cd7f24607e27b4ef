OUT=gpurun_out/r03zc; mkdir -p $OUT
for b in exp_bits exp_bits_p1 exp_bits exp_bits_p1; do
  echo "=== $b" >> $OUT/coreside.txt
  timeout 300 tools/$b --coreside 2>&1 | grep -E "^pipe|MISMATCH|RESULT|contraction alone, even split 256: shipped" >> $OUT/coreside.txt
done
cat $OUT/coreside.txt
