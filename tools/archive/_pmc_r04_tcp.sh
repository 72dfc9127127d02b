# Round 4: TCP / TA counters in groups of TWO per pass (the 8-counter TCP pass and the 6-counter TA pass of round 3 hung: the probe
# tools/_pmc_tcp_probe.sh showed two per pass complete in seconds), for the three kernels the roofline discussion is about; see the note at the end.
export TMPDIR=/tmp
G2="TCP_GATE_EN1_sum,TCP_GATE_EN2_sum;TCP_TA_TCP_STATE_READ_sum,TCP_TOTAL_CACHE_ACCESSES_sum;TCP_TCC_READ_REQ_sum,TCP_PENDING_STALL_CYCLES_sum;TCP_TCP_TA_DATA_STALL_CYCLES_sum,TCP_TCR_TCP_STALL_CYCLES_sum;TA_TA_BUSY_sum,TA_BUFFER_WAVEFRONTS_sum;TA_FLAT_READ_WAVEFRONTS_sum,TA_DATA_STALLED_BY_TC_CYCLES_sum;TA_ADDR_STALLED_BY_TC_CYCLES_sum,TA_ADDR_STALLED_BY_TD_CYCLES_sum;GRBM_GUI_ACTIVE"
PMC_PASS_TIMEOUT=100 python3 tools/pmc_kernel.py "enc_gemm_rows32_fused_kernel" "$G2" -- --graphs 64 --nodes 128 --enc-unsplit
cp gpurun_out/pmc_kernel/summary.json gpurun_out/r4/r04_tcp_ta_encgemm_rows32_64x128.json
PMC_PASS_TIMEOUT=100 python3 tools/pmc_kernel.py "enc_gemm_split_lds_kernel<true" "$G2" -- --graphs 512 --nodes 128
cp gpurun_out/pmc_kernel/summary.json gpurun_out/r4/r04_tcp_ta_encgemm_lds_fused_512x128.json
PMC_PASS_TIMEOUT=100 python3 tools/pmc_kernel.py "mpn_step_pipe_kernel<false, true, true" "$G2" -- --graphs 64 --nodes 256
cp gpurun_out/pmc_kernel/summary.json gpurun_out/r4/r04_tcp_ta_step_pipe_64x256.json
# (Round 4 ended this script with one pass of four and one pass of the original EIGHT TCP counters to find where the limit is: four fit,
# eight abort rocprofv3 with SIGABRT inside the profiled process -- counter-slot over-subscription, profiles/r04_logs/pmc_r04_tcp.log.  The
# finding is recorded; the pass is gone, and tools/pmc_kernel.py now re-cuts any group beyond four TCP_* / TA_* counters.)
rm -rf gpurun_out/pmc_kernel
