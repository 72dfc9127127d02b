for rep in 1 2 3; do
for spec in 1x256 1x128 1x64; do
echo "--- default $spec"; python3 tools/exp_sizes.py $spec
for w in 1 2; do
echo "--- wps $w $spec"; GNNCCA_DIAG=1 GNNCCA_WPS=$w python3 tools/exp_sizes.py $spec
done; done; done 2>&1 | grep -v amdgpu.ids
