#!/bin/bash
# Round evidence on the GPU box: kernel trace, PMC passes, bench lines -> gpurun_out/<round>/ (copy to profiles/<round>/).
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r2 [cpufull]'
R=${1:-r1}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
if [ "$2" = "proffix" ]; then SKIPBENCH=1; fi
[ -z "$SKIPBENCH" ] && python bench.py > $OUT/bench_n1.raw 2> $OUT/bench_n1.err; tail -1 $OUT/bench_n1.raw > $OUT/bench_n1.json
python bench.py --no-cpu-baseline --no-children --resolution 512 2>/dev/null | tail -1 > $OUT/bench_m512.json
python bench.py --no-cpu-baseline --no-children --batch 3 --steps 4 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_b3.json
python bench.py --no-cpu-baseline --operand fp16 2>/dev/null | tail -1 > $OUT/bench_fp16.json
# the cost of exactness: the split-operand precision modes (the ones that meet the 1e-3 decoded-frame tolerance)
python bench.py --no-cpu-baseline --operand bf16x3 --steps 4 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_bf16x3.json
python bench.py --no-cpu-baseline --operand bf16x6 --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_bf16x6.json
# BASELINE config 5: MX-fp8 scores in the long self-attention
MUDG_ATTN_FP8=1 python bench.py --no-cpu-baseline --no-children 2>/dev/null | tail -1 > $OUT/bench_fp8attn.json
# same box, same (debug-variants) library: the 128 x 128 kernels only vs the shipped rule with the 288 x 320 tile (round 5), bf16 and bf16x3
MUDG_DEBUG_VARIANTS=1 MUDG_GEMM_W288=0 python bench.py --no-cpu-baseline --no-children --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_w288_off.json
MUDG_DEBUG_VARIANTS=1 MUDG_GEMM_W288=1 python bench.py --no-cpu-baseline --no-children --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_w288_rule.json
MUDG_DEBUG_VARIANTS=1 MUDG_GEMM_W288=0 python bench.py --no-cpu-baseline --no-children --operand bf16x3 --steps 4 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_bf16x3_w288_off.json
MUDG_DEBUG_VARIANTS=1 MUDG_GEMM_W288=1 python bench.py --no-cpu-baseline --no-children --operand bf16x3 --steps 4 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_bf16x3_w288_rule.json
# round 6: the two-workgroup 144 x 256 GEGLU kernel off / by its rule (same box, same library), its per-shape timings, ablation and counters
MUDG_DEBUG_VARIANTS=1 MUDG_GEMM_H144=0 python bench.py --no-cpu-baseline --no-children --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_h144_off.json
MUDG_DEBUG_VARIANTS=1 MUDG_GEMM_H144=1 python bench.py --no-cpu-baseline --no-children --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/bench_h144_rule.json
MUDG_DEBUG_VARIANTS=1 python tools/exp_h144.py time 2>/dev/null | grep geglu > $OUT/h144.txt
MUDG_DEBUG_VARIANTS=1 python tools/exp_h144.py ablate 2>/dev/null | grep geglu > $OUT/h144_ablate.txt
MUDG_DEBUG_VARIANTS=1 python tools/exp_stamps.py > $OUT/stamps.txt 2>/dev/null
# round 6: the 160 x 320 tile (MDM512's frames): per-shape timings, parity, MDM512 with the tile off / by its rule (same box, same library), kernel trace
MUDG_DEBUG_VARIANTS=1 python tools/exp_w160.py parity > $OUT/w160_parity.txt 2>/dev/null
MUDG_DEBUG_VARIANTS=1 python tools/exp_w160.py time 2>/dev/null | grep -v amdgpu.ids > $OUT/w160_shapes.txt
MUDG_DEBUG_VARIANTS=1 MUDG_GEMM_W160=0 python bench.py --no-cpu-baseline --no-children --resolution 512 2>/dev/null | tail -1 > $OUT/bench_m512_w160_off.json
MUDG_DEBUG_VARIANTS=1 MUDG_GEMM_W160=1 python bench.py --no-cpu-baseline --no-children --resolution 512 2>/dev/null | tail -1 > $OUT/bench_m512_w160_rule.json
rocprofv3 --kernel-trace --stats -d /tmp/pt5 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-children --resolution 512 > /tmp/pt5.log 2>&1
python tools/rocprof_summary.py trace $(find /tmp/pt5 -name "*.db" | head -1) > $OUT/kernel_trace_m512.md
# every contraction shape of the step on the tile and on the 128 x 128 kernels (the measurement behind the rule in wgemm.hip), both builds
MUDG_DEBUG_VARIANTS=1 python tools/exp_w288.py time > $OUT/w288_shapes.txt 2>/dev/null
MUDG_DEBUG_VARIANTS=1 MUDG_OPERAND=bf16x3 python tools/exp_w288.py time > $OUT/w288_x3_shapes.txt 2>/dev/null
python tools/shape_profile.py > $OUT/shapes_w288.md 2>/dev/null
# the CPU baseline as a measurement: one full MDM512 oracle forward on this host
if [ "$2" = "cpufull" ]; then
  python bench.py --steps 3 --warmup 1 --cpu-baseline full --no-decode 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())['cpu_baseline']
json.dump({'tflops': d['tflops'], 'cores': d['cores'], 'seconds': d['seconds'], 'sample': d['sample']}, open('$OUT/cpu_baseline_full.json', 'w'), indent=1)"
fi
rocprofv3 --kernel-trace --stats -d /tmp/pt -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-children > /tmp/pt.log 2>&1
python tools/rocprof_summary.py trace $(find /tmp/pt -name "*.db" | head -1) > $OUT/kernel_trace.md
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -- python bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-profile --no-decode --no-children > /tmp/pf.log 2>&1
python tools/rocprof_summary.py pmc $(find /tmp/pf -name "*.db" | head -1) > $OUT/pmc_fetch.md
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -- python bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-profile --no-decode --no-children > /tmp/pw.log 2>&1
python tools/rocprof_summary.py pmc $(find /tmp/pw -name "*.db" | head -1) > $OUT/pmc_write.md
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pm -- python bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-profile --no-decode --no-children > /tmp/pm.log 2>&1
python tools/rocprof_summary.py mfma $(find /tmp/pm -name "*.db" | head -1) > $OUT/pmc_mfma.md
# the mode that meets the tolerance: kernel trace and MFMA-busy counters of the bf16x3 step
rocprofv3 --kernel-trace --stats -d /tmp/pt3 -- python bench.py --steps 2 --warmup 1 --operand bf16x3 --no-cpu-baseline --no-profile --no-children > /tmp/pt3.log 2>&1
python tools/rocprof_summary.py trace $(find /tmp/pt3 -name "*.db" | head -1) > $OUT/kernel_trace_bf16x3.md
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pm3 -- python bench.py --steps 1 --warmup 0 --operand bf16x3 --no-graph --no-cpu-baseline --no-profile --no-decode --no-children > /tmp/pm3.log 2>&1
python tools/rocprof_summary.py mfma $(find /tmp/pm3 -name "*.db" | head -1) > $OUT/pmc_mfma_bf16x3.md
# the training step of the full UNet: seconds per step, peak memory
timeout 900 python tools/train_bench.py 512 3 2>/dev/null | tail -1 > $OUT/train_bench.log
timeout 900 python tools/train_bench.py 1024 2 2>/dev/null | tail -1 >> $OUT/train_bench.log
timeout 900 python tools/train_bench.py 1024 2 ckpt 2>/dev/null | tail -1 >> $OUT/train_bench.log
timeout 900 python tools/train_bench.py 1024 2 stage2 2>/dev/null | tail -1 >> $OUT/train_bench.log
# the reference's training batch: 4 clips per GPU, activation checkpointing, two micro-batches per optimiser step
timeout 1500 python tools/train_bench.py 1024 1 ckpt b4 acc2 2>/dev/null | tail -1 >> $OUT/train_bench.log
rocprofv3 --kernel-trace --stats -d /tmp/ptr -- python tools/train_bench.py 1024 2 > /tmp/ptr.log 2>&1
python tools/rocprof_summary.py trace $(find /tmp/ptr -name "*.db" | head -1) > $OUT/train1024_kernel_trace.md
rm -f $OUT/bench_n1.raw
ls -la $OUT
