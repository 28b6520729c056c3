mkdir -p gpurun_out/r6
( time python -m pytest tests -m gpu -q ) > gpurun_out/r6/gpu_tests_final.log 2>&1
tail -6 gpurun_out/r6/gpu_tests_final.log
MUDG_DEBUG_VARIANTS=1 MUDG_GEMM_W160=0 python bench.py --no-cpu-baseline --no-children --resolution 512 2>/dev/null | tail -1 > gpurun_out/r6/bench_m512_w160_off.json
MUDG_DEBUG_VARIANTS=1 MUDG_GEMM_W160=1 python bench.py --no-cpu-baseline --no-children --resolution 512 2>/dev/null | tail -1 > gpurun_out/r6/bench_m512_w160_rule.json
python bench.py --no-cpu-baseline --no-children --resolution 512 2>/dev/null | tail -1 > gpurun_out/r6/bench_m512.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/pt5 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-children --resolution 512 > /tmp/pt5.log 2>&1
python tools/rocprof_summary.py trace $(find /tmp/pt5 -name "*.db" | head -1) > gpurun_out/r6/kernel_trace_m512.md
MUDG_DEBUG_VARIANTS=1 python tools/exp_w160.py parity 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r6/w160_parity.txt
MUDG_DEBUG_VARIANTS=1 python tools/exp_w160.py time 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r6/w160_shapes.txt
STRESS_TIMEOUT=600 bash tools/stress_tile.sh r6 2000 > gpurun_out/stress_r6.log 2>&1
python bench.py > gpurun_out/r6/bench_final.raw 2>/dev/null; tail -1 gpurun_out/r6/bench_final.raw > gpurun_out/r6/bench_final.json
python -c "
import json
for f in ('bench_m512_w160_off','bench_m512_w160_rule','bench_m512','bench_final'):
    d=json.load(open('gpurun_out/r6/'+f+'.json')); print(f, d['value'], d['ms_per_step'])
"
tail -3 gpurun_out/r6/w160_parity.txt; cat gpurun_out/r6/stress_summary.txt | grep -c "all bit-reproducible"
