# GPU box: the 32-row un-split GEMM with two k groups per workgroup (two waves per SIMD) against one group and against the split-K default
for rep in 1 2; do
for spec in 32x128 48x128 64x128 96x128 128x128 64x256; do
echo "--- default (split-K) $spec"; python3 tools/exp_sizes.py $spec
echo "--- unsplit, one group (nst 8 / 4) $spec"; python3 tools/exp_sizes.py $spec --unsplit
echo "--- unsplit, two k groups, 3 ahead $spec"; GNNCCA_DIAG=1 GNNCCA_GEMM_R32_NST=24 python3 tools/exp_sizes.py $spec --unsplit
echo "--- unsplit, two k groups, 2 ahead $spec"; GNNCCA_DIAG=1 GNNCCA_GEMM_R32_NST=22 python3 tools/exp_sizes.py $spec --unsplit
done; done 2>&1 | grep -v amdgpu.ids
