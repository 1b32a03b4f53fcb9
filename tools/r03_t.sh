cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_cluster.py tests/test_gpu_hull.py -x -q 2>&1 | tail -2
timeout 900 python tests/soak/cluster_bench.py 2>/dev/null | tail -1 | cut -c1-330
DIRECT_DDP_LIB=$PWD/build_variants/cluster_counts.so timeout 600 python tools/cluster_counts.py 2>&1 | tail -1
