cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_cluster.py tests/test_gpu_hull.py -x -q > gpurun_out/gpu_suite.log 2>&1; grep -E "passed|failed|Error" gpurun_out/gpu_suite.log | tail -3
timeout 900 python tests/soak/hull_soak.py 20 2>/dev/null | cut -c1-400
