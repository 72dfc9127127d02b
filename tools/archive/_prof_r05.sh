# Round 5 collection (run on the GPU box from the repo root): the workloads of rounds 3-4 with the round-5 kernels (fp16-split encoder GEMMs), plus a kernel
# trace of the headline's HIP-graph block form (GraphedForward.block: what the driver's line times).
set -x
python3 tools/collect_profiles.py r05_dense256 -- --steps 20 --warmup 5 --mode eager
python3 tools/collect_profiles.py r05_dense256_graphblock --trace-only -- --steps 20 --warmup 5 --mode graphk
python3 tools/collect_profiles.py r05_64x256 -- --graphs 64 --nodes 256 --steps 40 --warmup 5 --mode eager
python3 tools/collect_profiles.py r05_512x128 -- --graphs 512 --nodes 128 --steps 20 --warmup 3 --mode eager
python3 tools/collect_profiles.py r05_64x128 -- --graphs 64 --nodes 128 --steps 60 --warmup 5 --mode eager
python3 tools/collect_profiles.py r05_dense1024_L8 -- --nodes 1024 --L 8 --steps 40 --warmup 5 --mode eager
python3 tools/collect_profiles.py r05_64x256_bf16 -- --graphs 64 --nodes 256 --steps 40 --warmup 5 --mode eager --edge-state bf16
python3 tools/collect_profiles.py r05_dense1024_L8_bf16 -- --nodes 1024 --L 8 --steps 40 --warmup 5 --mode eager --edge-state bf16
for d in gpurun_out/r05_*; do rm -rf $d/trace $d/pmc0 $d/pmc1 $d/pmc2 $d/pmc3; done
# counters of the two new encoder kernels (MFMA / VALU / LDS / wait shares come with collect_profiles; TCP / TA in groups of <= 4 per pass)
G2="TCP_TCC_READ_REQ_sum,TCP_PENDING_STALL_CYCLES_sum;TCP_GATE_EN1_sum,TCP_TOTAL_CACHE_ACCESSES_sum;TA_TA_BUSY_sum,TA_BUFFER_WAVEFRONTS_sum;SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,GRBM_GUI_ACTIVE"
PMC_PASS_TIMEOUT=100 python3 tools/pmc_kernel.py "enc_gemm_f16_fused_kernel" "$G2" -- --graphs 512 --nodes 128
cp gpurun_out/pmc_kernel/summary.json gpurun_out/r05_tcp_ta_encgemm_f16_fused_512x128.json
PMC_PASS_TIMEOUT=100 python3 tools/pmc_kernel.py "enc_gemm_f16_rows32_kernel" "$G2" -- --graphs 64 --nodes 128
cp gpurun_out/pmc_kernel/summary.json gpurun_out/r05_tcp_ta_encgemm_f16_rows32_64x128.json
rm -rf gpurun_out/pmc_kernel
