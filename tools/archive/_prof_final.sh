set -x
python3 tools/collect_profiles.py r02f_dense256 -- --steps 20 --warmup 5
python3 tools/collect_profiles.py r02f_64x256 -- --graphs 64 --nodes 256 --steps 40 --warmup 5 --mode eager
python3 tools/collect_profiles.py r02f_512x128 -- --graphs 512 --nodes 128 --steps 20 --warmup 3 --mode eager
python3 tools/collect_profiles.py r02f_64x128 -- --graphs 64 --nodes 128 --steps 60 --warmup 5 --mode eager
for d in gpurun_out/r02f_*; do rm -rf $d/trace $d/pmc0 $d/pmc1 $d/pmc2 $d/pmc3; done
