mkdir -p gpurun_out; rm -f gpurun_out/ab.log
for rep in 1 2; do
for v in base w2; do
  export DIRECT_DDP_LIB=$PWD/build_variants/$v.so
  echo "== $v rep $rep" >> gpurun_out/ab.log
  timeout 120 python tools/ab_time.py free f32 5 >> gpurun_out/ab.log 2>&1
  timeout 120 python tools/ab_time.py corridor f32 5 >> gpurun_out/ab.log 2>&1
  timeout 120 python tools/ab_time.py free f32 3 16384 >> gpurun_out/ab.log 2>&1
done; done
