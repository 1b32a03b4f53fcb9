#!/bin/bash
# One-minute performance check of the hot kernel on the GPU box: fixed-20 phase-1 launch of config 2 at B = 4096 (x3)
# and B = 16384, then VALU / LDS instructions per DDP iteration from one rocprofv3 --pmc pass.
# usage (through gpurun): bash tools/quick_perf.sh [parity]
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT; export PYTHONPATH=$ROOT
if [ "$1" = "parity" ]; then python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "passed|failed" | tail -2; fi
for i in 1 2 3; do python tools/prof_one.py free f32 4096 100 20 | tail -1; done
python tools/prof_one.py free f32 16384 100 20 | tail -1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/qp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/qp -o q -- python $ROOT/tools/prof_one.py free f32 4096 100 20 > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/qp/**/*counter_collection.csv", recursive=True)[0]
by = {}
for r in csv.DictReader(open(f)):
    if "k_iterate" in r["Kernel_Name"]:
        by.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        by[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for n, d in sorted(by.items()):
    print(n, "per DDP iteration: %.0f" % (max(d.values()) / 81920.0))
PY
rm -rf /tmp/qp2
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/qp2 -o q -- python $ROOT/tools/pass_counts.py > /dev/null 2>&1
python $ROOT/tools/pass_counts.py report /tmp/qp2
