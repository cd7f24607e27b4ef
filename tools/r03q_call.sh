# PMC passes (FETCH_SIZE / WRITE_SIZE / SQ) of the driver's command on the k-bits default, + TCC hit/miss + the bits boundary
export PMC_SETS="TCC_HIT_sum,TCC_MISS_sum,TCC_REQ_sum SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_INSTS_VALU,SQ_INSTS_MFMA,SQ_INST_CYCLES_VMEM,SQ_ACTIVE_INST_LDS"
bash tools/gpu_round.sh r03q pmc pmcx
