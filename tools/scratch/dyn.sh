mkdir -p gpurun_out
rm -f gpurun_out/dyn.log
timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 >> gpurun_out/dyn.log
for c in 1; do
  echo "== chunk $c" >> gpurun_out/dyn.log
  DIRECT_DDP_CHUNK=$c timeout 60 python tools/ab_time.py free f32 5 >> gpurun_out/dyn.log 2>&1
  DIRECT_DDP_CHUNK=$c timeout 60 python tools/ab_time.py corridor f32 5 >> gpurun_out/dyn.log 2>&1
  DIRECT_DDP_CHUNK=$c timeout 60 python tools/ab_time.py free f64 5 >> gpurun_out/dyn.log 2>&1
done
echo "== static" >> gpurun_out/dyn.log
DIRECT_DDP_SCHED=static timeout 60 python tools/ab_time.py corridor f32 5 >> gpurun_out/dyn.log 2>&1
timeout 120 python -c "
import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import numpy as np, gpu_check as g
g.timing('free',4096,100,np.float32)
g.timing('corridor',4096,100,np.float32)
" >> gpurun_out/dyn.log 2>&1
