mkdir -p gpurun_out/r03c
( timeout 900 tools/exp_bits > gpurun_out/r03c/exp_bits.txt 2>&1; echo "exp_bits exit $?" >> gpurun_out/r03c/exp_bits.txt )
grep -E "MISMATCH|RESULT|full size|time |pipe " gpurun_out/r03c/exp_bits.txt | tail -60
grep -E "MISMATCH|RESULT|full size|time |pipe " gpurun_out/r03c/exp_bits.txt | tail -70
