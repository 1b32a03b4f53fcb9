#!/bin/bash
# per-library SQ counters of the B = 16384 launch: usage tools/ab_counters.sh a.so b.so
cd ${GRAFT_REPO_ROOT:-$PWD}; R=$PWD; export PYTHONPATH=$PWD
cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/abc
  DIRECT_DDP_LIB=$R/$lib rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d /tmp/abc -o q -- python $R/tools/prof_one.py free f32 ${B:-16384} 100 20 > /dev/null 2>&1
  python - $lib <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/abc/**/*counter_collection.csv", recursive=True)[0]
by = {}
for r in csv.DictReader(open(f)):
    if "k_iterate" in r["Kernel_Name"]:
        by.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        by[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
print(sys.argv[1], " ".join("%s=%.4g" % (n.replace("SQ_", ""), max(d.values())) for n, d in sorted(by.items())))
PY
done
