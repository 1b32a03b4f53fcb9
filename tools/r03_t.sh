cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_hull.py tests/test_gpu_cluster.py -x -q 2>&1 | tail -15
