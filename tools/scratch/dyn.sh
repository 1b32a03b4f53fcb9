mkdir -p gpurun_out
rm -f gpurun_out/dyn.log
timeout 100 python tools/ab_time.py free f32 5 >> gpurun_out/dyn.log 2>&1; echo "rc $?" >> gpurun_out/dyn.log
timeout 100 python tools/ab_time.py corridor f32 5 >> gpurun_out/dyn.log 2>&1; echo "rc $?" >> gpurun_out/dyn.log
DIRECT_DDP_LIB=$PWD/build_variants/stats.so timeout 100 python tools/ab_time.py free f32 2 >> gpurun_out/dyn.log 2>&1
DIRECT_DDP_SCHED=static timeout 100 python tools/ab_time.py free f32 3 >> gpurun_out/dyn.log 2>&1
timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 >> gpurun_out/dyn.log
