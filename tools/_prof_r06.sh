# Round 6 collection (run on the GPU box from the repo root, one part per gpurun call: `bash tools/_prof_r06.sh A|B|C`): the workloads of rounds 3-5
# with the round-6 kernels (the 32-row slices encoder below 4096 nodes), a kernel trace of the headline's HIP-graph block form
# (GraphedForward.block: what the driver's line times), TCP / TA counters of the new kernel, the Terrace pipeline's kernels.
set -x
part=${1:-A}
if [ $part = A ]; then
python3 tools/collect_profiles.py r06_dense256 -- --steps 20 --warmup 5 --mode eager
python3 tools/collect_profiles.py r06_dense256_graphblock --trace-only -- --steps 20 --warmup 5 --mode graphk
python3 tools/collect_profiles.py r06_64x256 -- --graphs 64 --nodes 256 --steps 40 --warmup 5 --mode eager
python3 tools/collect_profiles.py r06_512x128 -- --graphs 512 --nodes 128 --steps 20 --warmup 3 --mode eager
fi
if [ $part = B ]; then
python3 tools/collect_profiles.py r06_64x128 -- --graphs 64 --nodes 128 --steps 60 --warmup 5 --mode eager
python3 tools/collect_profiles.py r06_dense1024_L8 -- --nodes 1024 --L 8 --steps 40 --warmup 5 --mode eager
python3 tools/collect_profiles.py r06_64x256_bf16 -- --graphs 64 --nodes 256 --steps 40 --warmup 5 --mode eager --edge-state bf16
python3 tools/collect_profiles.py r06_dense1024_L8_bf16 -- --nodes 1024 --L 8 --steps 40 --warmup 5 --mode eager --edge-state bf16
fi
for d in gpurun_out/r06_*; do rm -rf $d/trace $d/pmc0 $d/pmc1 $d/pmc2 $d/pmc3; done
if [ $part = C ]; then
# counters of the new encoder kernel (MFMA / VALU / LDS / wait shares come with collect_profiles; TCP / TA in groups of <= 4 per pass)
G2="TCP_TCC_READ_REQ_sum,TCP_PENDING_STALL_CYCLES_sum;TCP_GATE_EN1_sum,TCP_TOTAL_CACHE_ACCESSES_sum;TA_TA_BUSY_sum,TA_BUFFER_WAVEFRONTS_sum;SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,GRBM_GUI_ACTIVE"
PMC_PASS_TIMEOUT=100 python3 tools/pmc_kernel.py "enc_gemm_f16_slices4_kernel" "$G2" -- --nodes 256
cp gpurun_out/pmc_kernel/summary.json gpurun_out/r06_tcp_ta_encgemm_f16_slices4_dense256.json
PMC_PASS_TIMEOUT=100 python3 tools/pmc_kernel.py "enc_gemm_f16_slices8_kernel" "$G2" -- --nodes 1024 --L 8
cp gpurun_out/pmc_kernel/summary.json gpurun_out/r06_tcp_ta_encgemm_f16_slices8_dense1024.json
rm -rf gpurun_out/pmc_kernel
# the Terrace pipeline's kernels: rocprofv3 --kernel-trace --stats over tools/prof_pipeline_host.py (352 batches of 64 frames through FramePipeline)
here=$PWD
rm -rf gpurun_out/r06_terrace_trace; (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $here/gpurun_out/r06_terrace_trace -- python3 $here/tools/prof_pipeline_host.py > $here/gpurun_out/r06_terrace_trace.log 2>&1) || true
cp gpurun_out/r06_terrace_trace/*/*kernel_stats.csv gpurun_out/terrace_pipeline_r06_kernel_stats.csv 2>/dev/null || true
python3 tools/kernel_breakdown.py gpurun_out/terrace_pipeline_r06_kernel_stats.csv 352 > gpurun_out/terrace_pipeline_r06_breakdown.txt 2>&1 || true
rm -rf gpurun_out/r06_terrace_trace
fi
