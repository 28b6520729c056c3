#!/bin/bash
# LDS bank conflicts and wait counters of the one-stage 3x3 conv kernel without / with the shared activation stage (XSHARE), two shapes
# (debug-variants library).    /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/pmc_conv_xshare.sh'
REPO=$(pwd); OUT=$REPO/gpurun_out/r4; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $REPO
: > $OUT/pmc_conv_xshare.md
for X in 0 1; do
  for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    rm -rf /tmp/pcx
    MUDG_DEBUG_VARIANTS=1 MUDG_CONV_XSHARE=$X ONLY=conv3 rocprofv3 --kernel-trace --pmc $set -d /tmp/pcx -- python tools/exp_tiles.py > /tmp/pcx.log 2>&1
    echo "## CONV_XSHARE=$X: $set (sum over all launches of tools/exp_tiles.py ONLY=conv3, per kernel)" >> $OUT/pmc_conv_xshare.md
    python tools/rocprof_summary.py pmc $(find /tmp/pcx -name "*.db" | head -1) | grep "gemm_kernel<Geo<4, 2>, 1, true, true\|counter\|---" >> $OUT/pmc_conv_xshare.md
    echo >> $OUT/pmc_conv_xshare.md
  done
done
cat $OUT/pmc_conv_xshare.md
