# GPU box: step kernels compiled for 5 / 6 waves per SIMD (diagnostic twins, -DGNNCCA_FAST_WAVES) at the latency-bound mid sizes
for rep in 1 2; do
for spec in 32x128 64x128 128x128 16x256 1x1024; do
echo "--- default (4 waves per SIMD) $spec"; python3 tools/exp_sizes.py $spec
echo "--- forced 5 waves $spec"; GNNCCA_LIB=$PWD/gnn-cca_amd/lib/libgnncca_mpn_w5.so python3 tools/exp_sizes.py $spec
echo "--- forced 6 waves $spec"; GNNCCA_LIB=$PWD/gnn-cca_amd/lib/libgnncca_mpn_w6.so python3 tools/exp_sizes.py $spec
done; done 2>&1 | grep -v amdgpu.ids
