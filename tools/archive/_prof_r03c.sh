# Round 3, fourth (final) collection: every workload of the first two, with the final kernels (plan riding in the 128-row GEMM,
# the plan's up-front padded-layout check, the tail's preloaded weights, the 256-row epilogue in quarter order, plan_only_kernel, the MFMA tail from 2560 nodes).
set -x
python3 tools/collect_profiles.py r03_dense256 -- --steps 20 --warmup 5 --mode eager
python3 tools/collect_profiles.py r03_64x256 -- --graphs 64 --nodes 256 --steps 40 --warmup 5 --mode eager
python3 tools/collect_profiles.py r03_512x128 -- --graphs 512 --nodes 128 --steps 20 --warmup 3 --mode eager
python3 tools/collect_profiles.py r03_64x128 -- --graphs 64 --nodes 128 --steps 60 --warmup 5 --mode eager
python3 tools/collect_profiles.py r03_dense1024_L8 -- --nodes 1024 --L 8 --steps 40 --warmup 5 --mode eager
python3 tools/collect_profiles.py r03_64x256_bf16 -- --graphs 64 --nodes 256 --steps 40 --warmup 5 --mode eager --edge-state bf16
python3 tools/collect_profiles.py r03_dense1024_L8_bf16 -- --nodes 1024 --L 8 --steps 40 --warmup 5 --mode eager --edge-state bf16
for d in gpurun_out/r03_*; do rm -rf $d/trace $d/pmc0 $d/pmc1 $d/pmc2 $d/pmc3; done
