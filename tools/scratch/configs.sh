mkdir -p gpurun_out/cfg
python bench.py --no-cpu-baseline --kind corridor > gpurun_out/cfg/c3.json 2> gpurun_out/cfg/c3.err
python bench.py --no-cpu-baseline --dtype f64 > gpurun_out/cfg/c2_f64.json 2> gpurun_out/cfg/c2_f64.err
python bench.py --no-cpu-baseline --batch 16384 --nseg 300 --dtype f64 --steps 3 --warmup 1 > gpurun_out/cfg/c4.json 2> gpurun_out/cfg/c4.err
python bench.py --no-cpu-baseline --batch 16384 > gpurun_out/cfg/c2_b16k.json 2> gpurun_out/cfg/c2_b16k.err
python -c "
import sys; sys.path.insert(0,'tests/soak'); sys.path.insert(0,'.')
import numpy as np, gpu_check as g
g.timing('free',4096,100,np.float32)
" > gpurun_out/cfg/pcie.log 2>&1
