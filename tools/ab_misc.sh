for spec in "--nodes 1024 --L 8" "--nodes 768" "--nodes 512" "--nodes 384" "--nodes 128" "--nodes 32"; do for rep in 1 2; do
echo "--- table in LDS $spec"; python3 bench.py $spec --no-cpu-baseline --no-scale-probe --no-config4 --profile-reps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step']*1e3, d['config']['mode'])"
echo "--- gathers from L2 $spec"; GNNCCA_DIAG=1 GNNCCA_PD_LDS_MAX=0 python3 bench.py $spec --no-cpu-baseline --no-scale-probe --no-config4 --profile-reps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step']*1e3, d['config']['mode'])"
done; done
