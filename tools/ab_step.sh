#!/bin/bash
# GPU box: A/B/C of the step kernels on ONE box, alternated so that clock / box drift shows up as spread, not as a result:
#   r2      = round 2's kernel (f32 MFMA message, exec-masked global accesses)       GNNCCA_DIAG=1 GNNCCA_STEP_R2=1
#   pipe    = round 3's kernel (bf16-split message, buffer addressing): the default beyond 512 nodes
#   (the "persist" arm of the logged runs was the persistent-wave form, since shelved: profiles/r03_logs/shelved_kernels/step_persist_r3.cuh.txt)
#   bash tools/ab_step.sh 64x256 512x128 64x128
for rep in 1 2; do
  for spec in "$@"; do
    echo "--- pipe    $spec"; python3 tools/exp_sizes.py $spec
    echo "--- r2      $spec"; GNNCCA_DIAG=1 GNNCCA_STEP_R2=1 python3 tools/exp_sizes.py $spec
  done
done 2>&1 | grep -v amdgpu.ids
