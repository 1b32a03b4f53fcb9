cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
timeout 300 python -m pytest tests/test_gpu_cluster.py -x -q 2>&1 | grep -E "^E  |^FAILED|passed|failed|Timeout" | head -5
timeout 300 python tests/soak/cluster_bench.py 64 2>&1 | tail -1 | cut -c1-330
for cfg in "1 1" "1 0" "0 1" "0 0"; do set -- $cfg; echo -n "HELP=$1 PAIR=$2: "; DIRECT_DDP_HELP=$1 DIRECT_DDP_PAIR=$2 python tools/ab_time.py free f32 5 4096 | tail -1; done
for s in 2560 2816; do echo -n "SLOTS=$s: "; DIRECT_DDP_SLOTS=$s python tools/ab_time.py free f32 5 4096 | tail -1; done
echo -n "corridor: "; python tools/ab_time.py corridor f32 5 4096 | tail -1
