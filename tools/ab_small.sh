for rep in 1 2; do for spec in 1x512 1x128 1x64 4x256; do
echo "--- r3 $spec"; GNNCCA_DIAG=1 GNNCCA_STEP_NOPERSIST=1 python3 tools/exp_sizes.py $spec
echo "--- r2 $spec"; GNNCCA_DIAG=1 GNNCCA_STEP_R2=1 python3 tools/exp_sizes.py $spec
done; done 2>&1 | grep -v amdgpu.ids
