#!/bin/bash
# L2 hit rate and HBM bytes of the cluster path's kernels (separate --pmc passes; run through gpurun from the repo root):
#   writes gpurun_out/rNN_cluster_pmc.json.  usage: tools/cluster_pmc.sh r03
R=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cpmc
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/cpmc/p$i -o pmc -- python $ROOT/tests/soak/cluster_bench.py 64 > /dev/null 2>&1
done
python - <<PY
import csv, glob, json, collections
out = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob('/tmp/cpmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        for n in ('k_convex', 'k_resolve_fast', 'k_compact', 'k_inflate', 'k_hull_edges'):
            if n in k:
                out[n][r['Counter_Name']] += float(r['Counter_Value'])
res = {}
for n, c in out.items():
    d = dict(c)
    if 'TCC_HIT_sum' in d: d['l2_hit_rate'] = d['TCC_HIT_sum'] / max(d['TCC_HIT_sum'] + d.get('TCC_MISS_sum', 0), 1)
    res[n] = d
res['note'] = 'sums over the 7 polygon_generation / hull calls of tests/soak/cluster_bench.py 64 (1 warm-up of 2 seeds + 6 of 64 seeds); FETCH_SIZE / WRITE_SIZE in kB as reported (see the newest profiles/r*_hbm_calib.json for the factors)'
json.dump(res, open('$ROOT/gpurun_out/${R}_cluster_pmc.json', 'w'), indent=1)
print(json.dumps(res)[:1500])
PY
