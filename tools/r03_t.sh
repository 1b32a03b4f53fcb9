cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_hull.py -x -q 2>&1 | tail -30
