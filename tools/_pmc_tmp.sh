G="SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_INSTS_VALU;SQ_INSTS_MFMA,SQ_VALU_MFMA_BUSY_CYCLES,SQ_INSTS_VMEM_RD,SQ_INSTS_LDS,SQ_WAIT_INST_LDS,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,GRBM_GUI_ACTIVE;FETCH_SIZE;TCC_HIT_sum,TCC_MISS_sum,TCC_REQ_sum"
python3 tools/pmc_kernel.py "enc_gemm_rows64_kernel" "$G" -- --graphs 512 --nodes 128 > gpurun_out/r2_pmc_r64.log 2>&1
cp gpurun_out/pmc_kernel/summary.json gpurun_out/r2_pmc_r64.json
rm -rf gpurun_out/pmc_kernel
