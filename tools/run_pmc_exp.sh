for a in 0 1 0 1; do MUDG_GELU_LUT=$a MUDG_GEMM256=0 TAG=lut=$a python tools/exp_tiles.py 2>&1 | grep "geglu=1"; done
python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -x -q -m gpu -k "geglu or transformer or unet" 2>&1 | tail -2
