mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > gpurun_out/ab_tests.log
for rep in 1 2; do
for v in base new; do
  if [ $v = base ]; then export DIRECT_DDP_LIB=$PWD/build_variants/base.so; else unset DIRECT_DDP_LIB; fi
  echo "== $v rep $rep" >> gpurun_out/ab.log
  python -c "
import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import numpy as np, gpu_check as g
g.timing('corridor',4096,100,np.float32)
g.timing('corridor',4096,100,np.float64)
" 2>&1 | grep -v "^per-pass\|^plan" >> gpurun_out/ab.log
done; done
