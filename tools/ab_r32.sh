# GPU box: the 32-row fused encoder GEMM: W fragments in the blob's order against a contiguous (fragment-ordered) read (timing probe)
for rep in 1 2; do
for spec in 32x128 64x128 128x128; do
echo "--- r32 $spec"; GNNCCA_DIAG=1 python3 tools/exp_sizes.py $spec
echo "--- r32 wfrag probe $spec"; GNNCCA_DIAG=1 GNNCCA_GEMM_R32_WFRAG=1 python3 tools/exp_sizes.py $spec
done; done 2>&1 | grep -v amdgpu.ids
