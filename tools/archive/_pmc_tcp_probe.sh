# Round 4, VERDICT r3 item 5: why did the TCP / TA passes of tools/_pmc_r32.sh hang?  Hypothesis: per-block slot over-subscription
# (8 TCP_* and 6 TA_* counters in one pass).  Step 1: what the tool itself lists for those blocks.  Step 2: ONE pass with two TCP
# counters, ONE with two TA counters, 100 s limit each, rocprofv3's own output kept.  No loop, no retry.
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/r4/rocprofv3_L.txt 2>&1
grep -c . gpurun_out/r4/rocprofv3_L.txt
grep -i -B1 -A4 "TCP_TOTAL_CACHE_ACCESSES\b\|TA_TA_BUSY\b\|TCP_TCC_READ_REQ\b" gpurun_out/r4/rocprofv3_L.txt | head -60
PMC_PASS_TIMEOUT=100 python3 tools/pmc_kernel.py "enc_gemm_rows32_fused_kernel" "TCP_TOTAL_CACHE_ACCESSES_sum,TCP_TCC_READ_REQ_sum;TA_TA_BUSY_sum,TA_BUFFER_WAVEFRONTS_sum" -- --graphs 64 --nodes 128 --enc-unsplit
