set -x
python3 tools/collect_profiles.py r03_dense1024_L8 -- --nodes 1024 --L 8 --steps 40 --warmup 5 --mode eager
python3 tools/collect_profiles.py r03_64x256_bf16 -- --graphs 64 --nodes 256 --steps 40 --warmup 5 --mode eager --edge-state bf16
python3 tools/collect_profiles.py r03_dense1024_L8_bf16 -- --nodes 1024 --L 8 --steps 40 --warmup 5 --mode eager --edge-state bf16
for d in gpurun_out/r03_*; do rm -rf $d/trace $d/pmc0 $d/pmc1 $d/pmc2 $d/pmc3; done
