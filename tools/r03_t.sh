cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
for r in 0 2 3 4 6 8; do echo "lpt $r"; DIRECT_DDP_LPT=$r timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']), d['natural_exit']['ms'], d['natural_exit']['kernel_ms'], round(d['natural_exit']['iter_per_s']), d['natural_exit']['iterations_max'])"; done
