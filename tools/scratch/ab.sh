mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > gpurun_out/ab_tests.log
for rep in 1 2; do
for v in base new; do
  if [ $v = base ]; then export DIRECT_DDP_LIB=$PWD/build_variants/base.so; else unset DIRECT_DDP_LIB; fi
  echo "== $v rep $rep" >> gpurun_out/ab.log
  timeout 120 python tools/ab_time.py free f32 7 >> gpurun_out/ab.log 2>&1
  timeout 120 python tools/ab_time.py corridor f32 7 >> gpurun_out/ab.log 2>&1
done; done
