for spec in "--nodes 256" "--nodes 64"; do for rep in 1 2; do for wg in 512 128 64; do
echo "--- f32 GEMM aiming at $wg workgroups $spec"; GNNCCA_DIAG=1 GNNCCA_GEMM_F32_WG=$wg python3 bench.py $spec --no-cpu-baseline --no-scale-probe --no-config4 --profile-reps 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step']*1e3, d['config']['mode'], d['kernels_us'])"
done; done; done
