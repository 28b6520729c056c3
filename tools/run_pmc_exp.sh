for v in 0 1 2 3 0; do MUDG_ATTN_VAR=$v python tools/exp_attn.py 2>&1 | grep -v amdgpu.ids | grep 9216 | sed "s/^/var=$v /"; done
