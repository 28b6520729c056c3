#!/bin/bash
# LDS bank conflicts and wait counters of the temporal conv kernels on plain tiles (korder 0) / on 8-pixel x 16-frame tiles with the shared slab (korder 1)
# (shipped library).    /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/pmc_tconv_tshare.sh'
REPO=$(pwd); OUT=$REPO/gpurun_out/r4; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $REPO
: > $OUT/pmc_tconv_tshare.md
for X in 0 1; do
  for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    rm -rf /tmp/pcx
    TKORDER=$X ONLY=tconv rocprofv3 --kernel-trace --pmc $set -d /tmp/pcx -- python tools/exp_tiles.py > /tmp/pcx.log 2>&1
    echo "## temporal conv korder=$X: $set (sum over all launches of tools/exp_tiles.py ONLY=tconv, per kernel)" >> $OUT/pmc_tconv_tshare.md
    python tools/rocprof_summary.py pmc $(find /tmp/pcx -name "*.db" | head -1) | grep "gemm_kernel<Geo<4, 2>, 2, true\|counter\|---" >> $OUT/pmc_tconv_tshare.md
    echo >> $OUT/pmc_tconv_tshare.md
  done
done
cat $OUT/pmc_tconv_tshare.md
