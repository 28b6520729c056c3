#!/bin/bash
# Stall / issue / MFMA counters of ONE level-0 GEGLU launch on the persistent 288 x 256 tile and on the two-workgroup 144 x 256 kernel
# (tools/exp_h144.py one; read the last dispatch of each kernel).   gpurun -- 'bash tools/pmc_h144.sh'
REPO=$(pwd); OUT=$REPO/gpurun_out/r6/pmc_h144; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $REPO
export MUDG_DEBUG_VARIANTS=1
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA"; do
  i=$((i+1))
  rm -rf /tmp/ph$i
  timeout 240 rocprofv3 --kernel-trace --pmc $set -d /tmp/ph$i -- python tools/exp_h144.py one > $OUT/run$i.log 2>&1
  python tools/rocprof_summary.py pmcd $(find /tmp/ph$i -name "*.db" | head -1) geglu > $OUT/pass$i.md 2>> $OUT/run$i.log
  python tools/rocprof_summary.py pmcd $(find /tmp/ph$i -name "*.db" | head -1) pkernel >> $OUT/pass$i.md 2>> $OUT/run$i.log
done
tail -n 12 $OUT/pass*.md
