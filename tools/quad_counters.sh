#!/bin/bash
# VALU / LDS / SALU instructions per trajectory-iteration and pipe busy fractions of the label-model kernel
# (two rocprofv3 --pmc passes over tools/quad_bench.py).  usage (through gpurun): bash tools/quad_counters.sh
ROOT=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$ROOT
rm -rf /tmp/qq /tmp/qq2
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/qq -o q -- python $ROOT/tools/quad_bench.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 --kernel-trace --output-format csv -d /tmp/qq2 -o q -- python $ROOT/tools/quad_bench.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
c = {}
for d in ("/tmp/qq", "/tmp/qq2"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    by = {}
    for r in csv.DictReader(open(f)):
        if "k_quad_iterate" in r["Kernel_Name"]:
            by.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            by[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for n, dd in by.items():
        c[n] = max(dd.values())
it = 4096 * 10.0
print("per trajectory-iteration: VALU %.0f  LDS %.0f  SALU %.0f  f64 FMA %.0f  f64 MUL %.0f" % (
    c["SQ_INSTS_VALU"] / it, c["SQ_INSTS_LDS"] / it, c["SQ_INSTS_SALU"] / it, c["SQ_INSTS_VALU_FMA_F64"] / it, c["SQ_INSTS_VALU_MUL_F64"] / it))
print("VALU busy %.2f  LDS busy %.2f  wait-any %.2f  wait-LDS %.2f of wave cycles" % (
    c["SQ_ACTIVE_INST_VALU"] / c["SQ_BUSY_CU_CYCLES"], c["SQ_LDS_IDX_ACTIVE"] / c["SQ_BUSY_CU_CYCLES"],
    c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"]))
PY
