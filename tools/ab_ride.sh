# GPU box: the plan inside the split-K GEMM's waves against a plan launch of its own (32-row kernel off in both arms)
for rep in 1 2 3; do
for spec in 48x128 64x128 128x128 64x256 100x256; do
echo "--- plan launch $spec"; GNNCCA_DIAG=1 GNNCCA_GEMM_R32_MIN=100000000 GNNCCA_NO_RIDE=1 python3 tools/exp_sizes.py $spec
echo "--- plan in the GEMM waves $spec"; GNNCCA_DIAG=1 GNNCCA_GEMM_R32_MIN=100000000 python3 tools/exp_sizes.py $spec
done; done 2>&1 | grep -v amdgpu.ids
