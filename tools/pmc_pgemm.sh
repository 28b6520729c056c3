#!/bin/bash
# LDS and stall counters of the contraction kernels over tools/exp_tiles.py's GEMM shapes (persistent rule vs one-tile kernels).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/pmc_pgemm.sh'
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_pgemm; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $REPO
for p in 1 0; do
  i=0
  for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU"; do
    i=$((i+1))
    rm -rf /tmp/pg$p$i
    MUDG_DEBUG_VARIANTS=1 MUDG_GEMM_PERSIST=$p MUDG_GEMM_WIDE=0 ONLY=gemm rocprofv3 --kernel-trace --pmc $set -d /tmp/pg$p$i -- python tools/exp_tiles.py > $OUT/run_p${p}_$i.log 2>&1
    python tools/rocprof_summary.py pmc $(find /tmp/pg$p$i -name "*.db" | head -1) > $OUT/persist${p}_pass$i.md 2>> $OUT/run_p${p}_$i.log
  done
done
ls -la $OUT
