cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "^E  |^FAILED|passed|failed" | head -12
for i in 1 2; do python tools/prof_one.py free f32 4096 100 20 | tail -1; done
python tools/prof_one.py free f32 16384 100 20 | tail -1
export DIRECT_DDP_LIB=$PWD/direct_amd/lib/libdirect_ddp_timing.so
echo "== 1 wave per SIMD"; DIRECT_DDP_SLOTS=1024 DIRECT_DDP_HELP=0 DIRECT_DDP_PAIR=0 python tools/phase_timing.py free 16384 | grep -E "^B_|^F_|kernel"
