for v in 0 1 0 1; do MUDG_ATTN_VAR=$v python tools/exp_attn.py 2>&1 | grep -v amdgpu.ids | sed "s/^/var=$v /"; done
MUDG_ATTN_VAR=1 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention or attn or softmax or transformer" 2>&1 | tail -2
