cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for V in ${VARIANTS:-0 9 17}; do
  for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
    rm -rf /tmp/pmc_$V
    timeout 120 rocprofv3 --pmc $SET --kernel-trace -d /tmp/pmc_$V -- python $R/tools/run_lowp_case.py --case i8 --rows 4000000 --variant $V --reps 2 > /tmp/pmc_run.log 2>&1
    echo "== variant $V : $SET"; python $R/tools/pmc_dump.py /tmp/pmc_$V "k_mfma_filter_"; tail -2 /tmp/pmc_run.log
  done
done
