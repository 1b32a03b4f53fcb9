#!/bin/bash
# Refreshes the judged artifacts of a round on the GPU box (run through gpurun from the repo root):
#   gpurun_out/rNN_bench.json               python bench.py
#   gpurun_out/rNN_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/rNN_{fetch,write}_size_k_iterate.csv + rNN_hbm_traffic.json   separate --pmc passes
# Copy the files into profiles/ afterwards.  usage: tools/profile_round.sh r01
R=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/${R}_bench.json 2> $OUT/${R}_bench.err
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
rm -rf /tmp/prof_stats /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o st -- python $ROOT/bench.py --no-cpu-baseline > $OUT/${R}_stats_run.log 2>&1
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/${R}_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_f -o f -- python $ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 1 > $OUT/${R}_fetch_run.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_w -o w -- python $ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 1 > $OUT/${R}_write_run.log 2>&1
python - $R $OUT <<'PY'
import csv, glob, json, sys
R, OUT = sys.argv[1], sys.argv[2]
res = {}
for tag, d, name in (("FETCH_SIZE", "/tmp/prof_f", "fetch"), ("WRITE_SIZE", "/tmp/prof_w", "write")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_iterate" in r["Kernel_Name"] and r["Counter_Name"] == tag]
    keep = ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "VGPR_Count", "LDS_Block_Size", "Scratch_Size"]
    keep = [k for k in keep if k in rows[0]]
    with open("%s/%s_%s_size_k_iterate.csv" % (OUT, R, name), "w") as g:
        w = csv.writer(g); w.writerow(keep)
        for r in rows: w.writerow([r[k] for k in keep])
    vals = [float(r["Counter_Value"]) for r in rows]
    big = [v for v in vals if v > 0.5 * max(vals)]  # the timed workload launches (the phase-0 launch is much smaller)
    res[tag] = {"per_dispatch_kb": vals, "workload_mean_kb": sum(big) / len(big)}
json.dump(res, open("%s/%s_hbm_traffic.json" % (OUT, R), "w"), indent=1)
print(json.dumps({k: v["workload_mean_kb"] for k, v in res.items()}))
PY
