#!/bin/bash
# Stall / LDS / MFMA counters of individual GEMM shapes (tools/exp_traffic.py: each shape twice, read the second dispatch).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/pmc_shapes.sh'
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_shapes; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $REPO
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_MFMA"; do
  i=$((i+1))
  rm -rf /tmp/pm$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pm$i -- python tools/exp_traffic.py > $OUT/run$i.log 2>&1
  python tools/rocprof_summary.py pmcd $(find /tmp/pm$i -name "*.db" | head -1) gemm_kernel > $OUT/pass$i.md 2>> $OUT/run$i.log
done
grep "M=" $OUT/run1.log > $OUT/shapes.txt
ls -la $OUT
