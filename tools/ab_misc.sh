for rep in 1 2; do
for spec in 64x128 128x128; do
echo "--- default $spec"; python3 tools/exp_sizes.py $spec
echo "--- direct, 256 workgroups (half the slabs), no ride $spec"; GNNCCA_DIAG=1 GNNCCA_NO_RIDE=1 GNNCCA_GEMM_DIRECT=1 GNNCCA_GEMM_DIRECT_WG=256 python3 tools/exp_sizes.py $spec
echo "--- direct, 512 workgroups, no ride $spec"; GNNCCA_DIAG=1 GNNCCA_NO_RIDE=1 GNNCCA_GEMM_DIRECT=1 python3 tools/exp_sizes.py $spec
echo "--- wps 2 $spec"; GNNCCA_DIAG=1 GNNCCA_WPS=2 python3 tools/exp_sizes.py $spec
done; done 2>&1 | grep -v amdgpu.ids
