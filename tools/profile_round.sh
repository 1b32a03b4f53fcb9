#!/bin/bash
# Refreshes the judged artifacts of a round on the GPU box (run through gpurun from the repo root):
#   gpurun_out/rNN_bench.json               python bench.py   (the line the driver will reproduce)
#   gpurun_out/rNN_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/rNN_hbm_calib.json           tools/hbm_calib: counter calibration for dword streams + measured HBM peak
#   gpurun_out/rNN_sq_counters.json, rNN_hbm_traffic.json   separate --pmc passes over tools/prof_one.py (same workload)
# Copy the files into profiles/ afterwards.  usage: tools/profile_round.sh r02 [quick]
R=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/${R}_bench.json 2> $OUT/${R}_bench.err
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
rm -rf /tmp/prof_stats /tmp/pmc /tmp/cal_f /tmp/cal_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o st -- python $ROOT/bench.py --no-cpu-baseline --no-secondary > $OUT/${R}_stats_run.log 2>&1
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/${R}_bench_kernel_stats.csv
# counter calibration + measured peak
$ROOT/tools/hbm_calib > $OUT/${R}_hbm_calib_plain.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/cal_f -o f -- $ROOT/tools/hbm_calib > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/cal_w -o w -- $ROOT/tools/hbm_calib > /dev/null 2>&1
python $ROOT/tools/hbm_calib_collect.py /tmp/cal_f /tmp/cal_w $OUT/${R}_hbm_calib_plain.log $OUT/${R}
# PMC passes over the workload (no torch in the process: a pass is ~10 s)
python $ROOT/tools/prof_one.py free f32 4096 100 20 > $OUT/${R}_plain.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc/p$i -o pmc -- python $ROOT/tools/prof_one.py free f32 4096 100 20 > $OUT/${R}_pmc_run$i.log 2>&1
done
python $ROOT/tools/pmc_collect.py /tmp/pmc $OUT/${R} free f32 4096 100 20
