set -x
G="SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_INSTS_VALU;SQ_INSTS_MFMA,SQ_VALU_MFMA_BUSY_CYCLES,SQ_INSTS_VMEM_RD,SQ_INSTS_LDS,SQ_WAIT_INST_LDS,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,GRBM_GUI_ACTIVE;FETCH_SIZE;WRITE_SIZE"
python3 tools/pmc_kernel.py "enc_gemm_split_lds_kernel<true" "$G" -- --graphs 512 --nodes 128 > gpurun_out/r2_pmc_enc_a.log 2>&1
cp gpurun_out/pmc_kernel/summary.json gpurun_out/r02_encgemm_lds_fused_512x128.json
python3 tools/pmc_kernel.py "enc_gemm_split_lds_kernel<false" "$G" -- --graphs 64 --nodes 256 > gpurun_out/r2_pmc_enc_b.log 2>&1
cp gpurun_out/pmc_kernel/summary.json gpurun_out/r02_encgemm_lds_split4_64x256.json
python3 tools/pmc_kernel.py "enc_gemm_split_lds_kernel<false" "$G" -- --graphs 64 --nodes 128 > gpurun_out/r2_pmc_enc_c.log 2>&1
cp gpurun_out/pmc_kernel/summary.json gpurun_out/r02_encgemm_lds_split8_64x128.json
python3 tools/pmc_kernel.py "enc_tail_mfma_kernel" "$G" -- --graphs 64 --nodes 128 > gpurun_out/r2_pmc_enc_d.log 2>&1
cp gpurun_out/pmc_kernel/summary.json gpurun_out/r02_enctail_mfma_64x128.json
rm -rf gpurun_out/pmc_kernel
