#!/bin/bash
# VALU / LDS pipe utilisation of the fixed-20 launch: usage sq_quick.sh <batch> [kind] [dtype]
ROOT=${GRAFT_REPO_ROOT:-$PWD}
B=${1:-16384}; KIND=${2:-free}; DT=${3:-f32}
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$ROOT
rm -rf /tmp/sqq
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/sqq/p$i -o q -- python $ROOT/tools/prof_one.py $KIND $DT $B 100 20 > /tmp/sqq_$i.log 2>&1
  grep "kernel ms" /tmp/sqq_$i.log
done
python - <<'PY'
import csv, glob
c = {}
for f in sorted(glob.glob("/tmp/sqq/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "k_iterate" in r["Kernel_Name"]]
    last = max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k in sorted(c):
    print("%-26s %.4g" % (k, c[k]))
g = c.get
if g("SQ_BUSY_CU_CYCLES"):
    print("VALU busy per SIMD  %.3f" % (g("SQ_ACTIVE_INST_VALU", 0) * 4 / (g("SQ_BUSY_CU_CYCLES") * 4)))
    print("LDS array busy      %.3f   (bank-conflict cycles %.3f of them)" % (g("SQ_LDS_IDX_ACTIVE", 0) / g("SQ_BUSY_CU_CYCLES"), g("SQ_LDS_BANK_CONFLICT", 0) / max(g("SQ_LDS_IDX_ACTIVE", 1), 1)))
    print("LDS issue busy      %.3f" % (g("SQ_ACTIVE_INST_LDS", 0) * 4 / (g("SQ_BUSY_CU_CYCLES") * 4)))
if g("SQ_WAVE_CYCLES"):
    print("wave: parked (s_waitcnt) %.3f  issue-stalled %.3f (LDS %.3f)  issuing %.3f" % (
        g("SQ_WAIT_ANY", 0) / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY", 0) / g("SQ_WAVE_CYCLES"),
        g("SQ_WAIT_INST_LDS", 0) / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_ANY", 0) / g("SQ_WAVE_CYCLES")))
PY
