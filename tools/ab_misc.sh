for rep in 1 2 3; do python3 tools/exp_sizes.py 1x64 1x256 1x512 1x1024 1x2048; done 2>&1 | grep -v amdgpu.ids
