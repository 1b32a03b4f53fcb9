cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/gpu_suite.log 2>&1; grep -E "passed|failed|Error" gpurun_out/gpu_suite.log | tail -3
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>&1 | grep '^{' | cut -c1-120
