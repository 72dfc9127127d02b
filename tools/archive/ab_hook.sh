#!/bin/bash
# same-box alternation: second round's edge state requested inside the first round (default) vs after it (GNNCCA_DIAG=1 GNNCCA_STEP_NOHOOK=1)
for rep in 1 2 3; do for spec in 64x256 200x256 1x2048; do
echo "--- hook   $spec"; python3 tools/exp_sizes.py $spec
echo "--- nohook $spec"; GNNCCA_DIAG=1 GNNCCA_STEP_NOHOOK=1 python3 tools/exp_sizes.py $spec
done; done 2>&1 | grep -v amdgpu.ids
