# GPU box: the MFMA tail below 4096 nodes?
for rep in 1 2; do
for spec in 1x1024 2x1024 1x2048 4x1000 1x3000 3x1024; do
echo "--- default (register-resident tail below 4096) $spec"; python3 tools/exp_sizes.py $spec
echo "--- mfma tail from 1024 $spec"; GNNCCA_DIAG=1 GNNCCA_TAIL_MFMA_MIN=1024 python3 tools/exp_sizes.py $spec
done; done 2>&1 | grep -v amdgpu.ids
