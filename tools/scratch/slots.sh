mkdir -p gpurun_out; rm -f gpurun_out/slots.log
for sl in 3072 2816 2560 2304 2048; do
  echo "== slots $sl" >> gpurun_out/slots.log
  DIRECT_DDP_SLOTS=$sl timeout 120 python tools/ab_time.py free f32 5 >> gpurun_out/slots.log 2>&1
  DIRECT_DDP_SLOTS=$sl timeout 120 python tools/ab_time.py corridor f32 5 >> gpurun_out/slots.log 2>&1
done
