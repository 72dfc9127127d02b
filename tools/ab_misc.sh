for spec in "--nodes 32" "--nodes 64" "--nodes 128"; do for rep in 1 2; do for w in 0 12 14; do
echo "--- step kernels: forced wps code $w (0 = default, 12 / 14 = two / four waves per node uncapped) $spec"; GNNCCA_DIAG=1 GNNCCA_WPS=$w python3 bench.py $spec --no-cpu-baseline --no-scale-probe --no-config4 --profile-reps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step']*1e3, d['config']['mode'])"
done; done; done
