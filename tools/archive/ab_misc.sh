# GPU box: the step kernels' early barrier (staging of the projection table) with several waves per node -- same box, alternating
for spec in "--nodes 256" "--nodes 128" "--nodes 512" "--nodes 2048"; do for rep in 1 2 3; do
echo "--- early barrier (rounds 1-2) $spec"; GNNCCA_DIAG=1 GNNCCA_STEP_EARLYBAR=1 python3 bench.py $spec --no-cpu-baseline --no-scale-probe --no-config4 --profile-reps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step']*1e3, d['config']['mode'])"
echo "--- combine barrier only $spec"; python3 bench.py $spec --no-cpu-baseline --no-scale-probe --no-config4 --profile-reps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step']*1e3, d['config']['mode'])"
done; done
