#!/bin/bash
# GPU box: BASELINE config 5 (dense1024, L = 8) under the step kernels' diagnostic switches, same box: waves per node, LDS gather table,
# column ranges (ab_ranges.py alternates them in-process), bf16 edge state.   bash tools/ab_config5.sh
for wps in 0 1 2 4; do
  for pdl in 801 99999; do
    echo "--- GNNCCA_WPS=$wps PD_LDS_MIN=$pdl"
    GNNCCA_DIAG=1 GNNCCA_WPS=$wps GNNCCA_PD_LDS_MIN=$pdl python3 tools/ab_ranges.py 1x1024:8
    GNNCCA_DIAG=1 GNNCCA_WPS=$wps GNNCCA_PD_LDS_MIN=$pdl AB_BF16=1 python3 tools/ab_ranges.py 1x1024:8
  done
done 2>&1 | grep -v "amdgpu.ids\|not a switch"
