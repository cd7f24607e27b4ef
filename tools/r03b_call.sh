mkdir -p gpurun_out/r03b
( timeout 900 tools/exp_bits > gpurun_out/r03b/exp_bits.txt 2>&1; echo "exp_bits exit $?" >> gpurun_out/r03b/exp_bits.txt )
( timeout 300 python tools/rccl_probe.py > gpurun_out/r03b/rccl_probe.txt 2>&1; echo "exit $?" >> gpurun_out/r03b/rccl_probe.txt )
grep -E "MISMATCH|RESULT|full size|time |pipe " gpurun_out/r03b/exp_bits.txt | tail -60
tail -8 gpurun_out/r03b/rccl_probe.txt
