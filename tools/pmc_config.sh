#!/bin/bash
# HBM traffic and SQ busy counters of ONE other configuration (separate --pmc passes over tools/prof_one.py):
# usage (through gpurun): bash tools/pmc_config.sh <tag> <kind> <f32|f64> <batch> <nseg> <iters>   -> gpurun_out/<tag>_*.json
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$ROOT
rm -rf /tmp/pmc_cfg
python $ROOT/tools/prof_one.py "$@" > $OUT/${TAG}_plain.log 2>&1   # the un-profiled kernel time the passes are checked against
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_cfg/p$i -o pmc -- python $ROOT/tools/prof_one.py "$@" > $OUT/${TAG}_pmc_run$i.log 2>&1
done
cp $(ls $ROOT/profiles/r*_hbm_calib.json | sort | tail -1) $OUT/${TAG}_hbm_calib.json 2>/dev/null
python $ROOT/tools/pmc_collect.py /tmp/pmc_cfg $OUT/${TAG} "$@"
