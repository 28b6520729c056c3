cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -x -q -m gpu -k "norm or fused or resblock or unet" 2>&1 | tail -2
rocprofv3 --kernel-trace --stats -d /tmp/pt -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-decode > /tmp/pt.log 2>&1
python tools/rocprof_summary.py trace $(find /tmp/pt -name "*.db" | head -1) | grep "gn_\|cast\|tattn\|ln_k" | cut -c1-50,95-150
