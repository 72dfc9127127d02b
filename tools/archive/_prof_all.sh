# GPU box: round-2 profiles of the five BASELINE workloads (+ bf16 edge state on two), summaries under gpurun_out/r02_*
set -x
python3 tools/collect_profiles.py r02_dense256 -- --steps 100 --warmup 10 --mode eager
python3 tools/collect_profiles.py r02_64x256 -- --graphs 64 --nodes 256 --steps 40 --warmup 5 --mode eager
python3 tools/collect_profiles.py r02_512x128 -- --graphs 512 --nodes 128 --steps 20 --warmup 3 --mode eager
python3 tools/collect_profiles.py r02_64x128 -- --graphs 64 --nodes 128 --steps 60 --warmup 5 --mode eager
python3 tools/collect_profiles.py r02_dense1024_L8 -- --nodes 1024 --L 8 --steps 60 --warmup 5 --mode eager
python3 tools/collect_profiles.py r02_64x256_bf16 -- --graphs 64 --nodes 256 --edge-state bf16 --steps 40 --warmup 5 --mode eager
python3 tools/collect_profiles.py r02_dense1024_L8_bf16 -- --nodes 1024 --L 8 --edge-state bf16 --steps 60 --warmup 5 --mode eager
# keep only the summaries (gpurun_out/ is merged back only while it stays under 64 MiB)
for d in gpurun_out/r02_*; do rm -rf $d/trace $d/pmc0 $d/pmc1 $d/pmc2 $d/pmc3; done
