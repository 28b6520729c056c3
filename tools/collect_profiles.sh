#!/bin/bash
# Round evidence on the GPU box: kernel trace, PMC passes, bench lines -> gpurun_out/<round>/ (copy to profiles/<round>/).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r1'
R=${1:-r1}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
python bench.py > $OUT/bench_n1.raw 2> $OUT/bench_n1.err; tail -1 $OUT/bench_n1.raw > $OUT/bench_n1.json
python bench.py --no-cpu-baseline --resolution 512 2>/dev/null | tail -1 > $OUT/bench_m512.json
python bench.py --no-cpu-baseline --batch 3 --steps 4 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_b3.json
python bench.py --no-cpu-baseline --operand fp16 2>/dev/null | tail -1 > $OUT/bench_fp16.json
rocprofv3 --kernel-trace --stats -d /tmp/pt -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /tmp/pt.log 2>&1
python tools/rocprof_summary.py trace $(find /tmp/pt -name "*.db" | head -1) > $OUT/kernel_trace.md
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -- python bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-profile --no-decode > /tmp/pf.log 2>&1
python tools/rocprof_summary.py pmc $(find /tmp/pf -name "*.db" | head -1) > $OUT/pmc_fetch.md
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -- python bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-profile --no-decode > /tmp/pw.log 2>&1
python tools/rocprof_summary.py pmc $(find /tmp/pw -name "*.db" | head -1) > $OUT/pmc_write.md
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pm -- python bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-profile --no-decode > /tmp/pm.log 2>&1
python tools/rocprof_summary.py mfma $(find /tmp/pm -name "*.db" | head -1) > $OUT/pmc_mfma.md
rm -f $OUT/bench_n1.raw
ls -la $OUT
