# Round 4 collection (run on the GPU box from the repo root): the seven workloads of round 3 with the final kernels, plus a kernel
# trace of the headline's HIP-graph block form (GraphedForward.block: what the driver's line times).
set -x
python3 tools/collect_profiles.py r04_dense256 -- --steps 20 --warmup 5 --mode eager
python3 tools/collect_profiles.py r04_dense256_graphblock --trace-only -- --steps 20 --warmup 5 --mode graphk
python3 tools/collect_profiles.py r04_64x256 -- --graphs 64 --nodes 256 --steps 40 --warmup 5 --mode eager
python3 tools/collect_profiles.py r04_512x128 -- --graphs 512 --nodes 128 --steps 20 --warmup 3 --mode eager
python3 tools/collect_profiles.py r04_64x128 -- --graphs 64 --nodes 128 --steps 60 --warmup 5 --mode eager
python3 tools/collect_profiles.py r04_dense1024_L8 -- --nodes 1024 --L 8 --steps 40 --warmup 5 --mode eager
python3 tools/collect_profiles.py r04_64x256_bf16 -- --graphs 64 --nodes 256 --steps 40 --warmup 5 --mode eager --edge-state bf16
python3 tools/collect_profiles.py r04_dense1024_L8_bf16 -- --nodes 1024 --L 8 --steps 40 --warmup 5 --mode eager --edge-state bf16
for d in gpurun_out/r04_*; do rm -rf $d/trace $d/pmc0 $d/pmc1 $d/pmc2 $d/pmc3; done
