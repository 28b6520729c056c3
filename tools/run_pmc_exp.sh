cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d /tmp/p1 -- python tools/exp_one.py > /tmp/p1.log 2>&1
DB=$(find /tmp/p1 -name "*.db" | head -1)
python tools/rocprof_summary.py pmcd $DB gemm > gpurun_out/pmc_gemm_sq.md 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d /tmp/p2 -- python tools/exp_one.py > /tmp/p2.log 2>&1
DB=$(find /tmp/p2 -name "*.db" | head -1)
python tools/rocprof_summary.py pmcd $DB gemm > gpurun_out/pmc_gemm_lds.md 2>&1
tail -3 /tmp/p2.log
