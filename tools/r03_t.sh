cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final_bench.log 2>&1; grep '^{' gpurun_out/final_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d.get('corridor_clusters'))"
