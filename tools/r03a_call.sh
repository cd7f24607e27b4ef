mkdir -p gpurun_out/r03a
export TMPDIR=/tmp
( timeout 900 tools/exp_bits > gpurun_out/r03a/exp_bits.txt 2>&1; echo "exp_bits exit $?" >> gpurun_out/r03a/exp_bits.txt )
( timeout 300 python tools/rccl_probe.py > gpurun_out/r03a/rccl_probe.txt 2>&1; echo "exit $?" >> gpurun_out/r03a/rccl_probe.txt )
( timeout 120 tools/exp_vmm > gpurun_out/r03a/vmm.txt 2>&1; echo "exit $?" >> gpurun_out/r03a/vmm.txt )
( timeout 120 tools/exp_vmm fault > gpurun_out/r03a/vmm_fault.txt 2>&1; echo "exit $?" >> gpurun_out/r03a/vmm_fault.txt )
( timeout 60 rocm-smi --showuse > gpurun_out/r03a/after.txt 2>&1 )
tail -5 gpurun_out/r03a/exp_bits.txt
