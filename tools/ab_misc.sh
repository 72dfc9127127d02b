# GPU box: from how many nodes should the split-bf16 128-row GEMM (plan riding along) replace the f32 MFMA GEMM?
for rep in 1 2; do
for spec in 1x256 1x384 1x512 1x640 1x768 1x896 2x256 3x256; do
echo "--- f32 GEMM (split from 1024) $spec"; python3 tools/exp_sizes.py $spec
echo "--- split-bf16 GEMM from 256 $spec"; GNNCCA_DIAG=1 GNNCCA_GEMM_SPLIT_MIN=256 python3 tools/exp_sizes.py $spec
done; done 2>&1 | grep -v amdgpu.ids
