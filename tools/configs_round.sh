#!/bin/bash
# bench.py on BASELINE configs 3, 4, 5 (per-GPU shard) and config 2 at B = 16384 -> gpurun_out/rNN_configs.json
R=${1:-r02}
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
{
  echo "["
  python bench.py --config 3 --no-cpu-baseline --no-secondary --steps 10; echo ","
  python bench.py --config 4 --no-cpu-baseline --no-secondary --steps 4 --warmup 1; echo ","
  python bench.py --config 5 --no-cpu-baseline --no-secondary --steps 6 --warmup 1; echo ","
  python bench.py --config 2 --batch 16384 --no-cpu-baseline --no-secondary --steps 6 --warmup 1
  echo "]"
} > gpurun_out/${R}_configs.json 2> gpurun_out/${R}_configs.err
python -c "
import json; d=json.load(open('gpurun_out/${R}_configs.json'))
for x in d: print(x['config']['workload'][:60], '%.3f M iter/s' % (x['value']/1e6), 'kernel %.1f ms' % x['roofline']['kernel_ms'], 'hbm frac %.3f' % x['roofline']['frac'])
"
