# GPU box: timing-only ablations of the 32-row un-split GEMM (results are garbage)
for rep in 1 2; do
for spec in 32x128; do
for abl in 0 2 4 5; do
echo "--- unsplit nst=8 ablation $abl (1: x loaded once, 2: W loaded once, 3: no conversion, no x) $spec"; GNNCCA_DIAG=1 GNNCCA_GEMM_R32_NST=8 GNNCCA_GEMM_R32_ABL=$abl python3 tools/exp_sizes.py $spec --unsplit
done; done; done 2>&1 | grep -v amdgpu.ids
