#!/bin/bash
# GPU box: A/B of the step kernels on one box -- round 3's pipelined kernel (default) against round 2's (GNNCCA_DIAG=1 GNNCCA_STEP_R2=1),
# alternated so that box-to-box and clock drift show up as spread, not as a result.
#   bash tools/ab_step.sh 64x256 512x128 64x128 1x256
set -e
for rep in 1 2; do
  for spec in "$@"; do
    echo "--- r3 $spec"; python3 tools/exp_sizes.py $spec
    echo "--- r2 $spec"; GNNCCA_DIAG=1 GNNCCA_STEP_R2=1 python3 tools/exp_sizes.py $spec
  done
done
