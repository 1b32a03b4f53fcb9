#!/bin/bash
# bench lines of the other single-GPU configurations + the cluster bench of a round (run through gpurun from the repo root):
#   gpurun_out/rNN_configs.json, rNN_cluster_bench.json, rNN_cluster_kernel_stats.csv   usage: tools/configs_round.sh r03
R=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT; export PYTHONPATH=$ROOT
python - > $OUT/${R}_configs.json 2> $OUT/${R}_configs.err <<PY
import json, subprocess, sys
out = {}
for c in (3, 4, 5, 6):
    r = subprocess.run([sys.executable, "bench.py", "--config", str(c), "--steps", "5", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    out["config%d" % c] = json.loads(lines[-1]) if lines else {"error": r.stderr[-500:]}
print(json.dumps(out, indent=1))
PY
python tests/soak/cluster_bench.py 64 2>/dev/null | tail -1 > $OUT/${R}_cluster_bench.json
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/cst
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cst -o c -- python $ROOT/tests/soak/cluster_bench.py 64 > /dev/null 2>&1
cp $(find /tmp/cst -name "*kernel_stats.csv" | head -1) $OUT/${R}_cluster_kernel_stats.csv
