for spec in "--nodes 256" "--nodes 64" "--nodes 512" "--nodes 1024 --L 8"; do for rep in 1 2; do for npw in 4 2 1; do
echo "--- tail: $npw nodes per workgroup $spec"; GNNCCA_DIAG=1 GNNCCA_TAIL_NPW=$npw python3 bench.py $spec --no-cpu-baseline --no-scale-probe --no-config4 --profile-reps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step']*1e3, d['config']['mode'])"
done; done; done
