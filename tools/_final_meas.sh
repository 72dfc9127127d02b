set -x
python tools/exp_sizes.py 1x32 1x64 1x256 2x256 4x256 8x256 16x256 64x128 64x256 512x128
python tools/exp_sizes.py 1x1024 --L 8
python tools/exp_sizes.py 1x256 64x256 512x128 --edge-state bf16
python tools/exp_sizes.py 1x1024 --L 8 --edge-state bf16
python bench.py --steps 200 --warmup 20
python bench.py --gpus 2 --single-device --backend gloo --steps 50 --warmup 5
python tools/bench_train_step.py
python tools/prof_end_to_end.py
python tools/bench_graph_build.py
