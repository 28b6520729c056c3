TAG=k128 MUDG_GEMM256=0 python tools/exp_tiles.py 2>&1 | grep -v amdgpu.ids > gpurun_out/tiles.txt
TAG=k128sb0 MUDG_GEMM256=0 MUDG_GEMM_SB=0 python tools/exp_tiles.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/tiles.txt
TAG=k128sb2 MUDG_GEMM256=0 MUDG_GEMM_SB=2 python tools/exp_tiles.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/tiles.txt
TAG=k256w16 MUDG_GEMM256=1 MUDG_GEMM256P=0 python tools/exp_tiles.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/tiles.txt
TAG=k256pp MUDG_GEMM256=1 MUDG_GEMM256P=2 python tools/exp_tiles.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/tiles.txt
TAG=auto python tools/exp_tiles.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/tiles.txt
