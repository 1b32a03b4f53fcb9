cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
for i in 1 2 3 4; do
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "^E  |^FAILED|passed|failed" | head -12
done
