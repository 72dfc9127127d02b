#!/bin/bash
# same-box alternation: the current step kernel vs round 2's (GNNCCA_DIAG=1 GNNCCA_STEP_R2=1)
for rep in 1 2 3; do
  for spec in "$@"; do
    echo "--- new $spec"; python3 tools/exp_sizes.py $spec
    echo "--- r2  $spec"; GNNCCA_DIAG=1 GNNCCA_STEP_R2=1 python3 tools/exp_sizes.py $spec
  done
done 2>&1 | grep -v amdgpu.ids
