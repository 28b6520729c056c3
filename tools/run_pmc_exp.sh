TAG=sb1 MUDG_GEMM256=0 python tools/exp_tiles.py 2>&1 | grep -v amdgpu.ids | grep "conv" > gpurun_out/tiles2.txt
TAG=sb2 MUDG_GEMM256=0 MUDG_GEMM_SB=2 python tools/exp_tiles.py 2>&1 | grep -v amdgpu.ids | grep "conv" >> gpurun_out/tiles2.txt
TAG=auto python tools/exp_tiles.py 2>&1 | grep -v amdgpu.ids | grep "conv" >> gpurun_out/tiles2.txt
