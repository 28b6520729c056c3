for pad in 0 64 8 72; do
PAD=$pad MUDG_GEMM256=0 python tools/exp_sq.py 2>&1 | grep -v "amdgpu.ids\|2048"
PAD=$pad MUDG_GEMM256=1 MUDG_GEMM256P=0 python tools/exp_sq.py 2>&1 | grep -v "amdgpu.ids\|2048"
PAD=$pad MUDG_GEMM256=1 MUDG_GEMM256P=2 python tools/exp_sq.py 2>&1 | grep -v "amdgpu.ids\|2048"
done
