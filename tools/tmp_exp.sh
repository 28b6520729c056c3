for v in 32 64 32 64; do MUDG_ATTN_Q=$v python tools/exp_attn.py 2>&1 | grep -v amdgpu.ids | sed "s/^/Q=$v /"; done
python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -x -q -m gpu -k "attention or attn or transformer or unet or resampler" 2>&1 | tail -2
