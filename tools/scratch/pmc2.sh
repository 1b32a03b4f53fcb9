cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
mkdir -p $O
i=10
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_BRANCH" \
           "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  PYTHONPATH=$R rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o pmc -- python $R/tools/prof_one.py free f32 > $O/run$i.log 2>&1
  db=$(find $O/p$i -name "*.db" | head -1)
  python $R/tools/pmc_report.py $db k_iterate > $O/report$i.txt 2>&1
  rm -rf $O/p$i
done
