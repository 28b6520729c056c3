cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python tools/exp_gn.py 2>&1 | grep -v amdgpu.ids
rocprofv3 --kernel-trace --stats -d /tmp/p1 -- python tools/exp_gn.py > /tmp/p1.log 2>&1
python tools/rocprof_summary.py trace $(find /tmp/p1 -name "*.db" | head -1) | grep "gn_\|ln_" | cut -c1-60,100-170
