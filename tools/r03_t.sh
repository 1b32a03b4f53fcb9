cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_hull.py -x -q 2>&1 | tail -8
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>&1 | grep '^{' | cut -c1-260
