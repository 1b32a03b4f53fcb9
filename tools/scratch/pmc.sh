cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -oE "\bSQ_[A-Z_0-9]+" | sort -u > $O/sq_counters.txt
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  PYTHONPATH=$R rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o pmc -- python $R/tools/prof_one.py free f32 > $O/run$i.log 2>&1
  db=$(find $O/p$i -name "*.db" | head -1)
  python $R/tools/pmc_report.py $db k_iterate > $O/report$i.txt 2>&1
  rm -rf $O/p$i
done
