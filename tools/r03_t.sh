cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -q 2>&1 | grep -E "^FAILED|passed|failed" | head -12
echo -n "free f64 N=100 B=4096: "; python tools/prof_one.py free f64 4096 100 20 | tail -1
echo -n "corridor f64 N=300 B=4096: "; python tools/prof_one.py corridor f64 4096 300 20 | tail -1
python tools/ab_time.py free f32 5 4096 | tail -1
