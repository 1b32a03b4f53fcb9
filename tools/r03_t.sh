cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_cluster.py -x -q > gpurun_out/gpu_suite.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/gpu_suite.log | tail -5
