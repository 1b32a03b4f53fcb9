cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q 2>&1 | grep -E "^FAILED|passed|failed" | head -12
python tools/ab_time.py free f32 5 4096 | tail -1
