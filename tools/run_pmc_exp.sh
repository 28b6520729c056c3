TAG=auto python tools/exp_tiles.py 2>&1 | grep -v amdgpu.ids | grep gemm
MUDG_GEMM256=0 python tools/exp_sq.py 2>&1 | grep -v amdgpu.ids
MUDG_GEMM256=1 python tools/exp_sq.py 2>&1 | grep -v amdgpu.ids
MUDG_GEMM256=1 MUDG_GEMM256P=2 python tools/exp_sq.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu 2>&1 | tail -2
