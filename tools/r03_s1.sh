#!/bin/bash
# round-3 session 1: parity suites after the boundary fixes, the N = 100 parity report, baseline perf
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT; export PYTHONPATH=$ROOT
OUT=$ROOT/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q > $OUT/r03_s1_pytest.log 2>&1
tail -3 $OUT/r03_s1_pytest.log
timeout 1500 python tests/soak/n100_report.py 64 > $OUT/r03_n100_report.log 2>&1
tail -2 $OUT/r03_n100_report.log | cut -c1-600
bash tools/quick_perf.sh > $OUT/r03_s1_perf.log 2>&1
cat $OUT/r03_s1_perf.log
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
from direct_amd import abi, problems, solver
b = problems.make_batch("free", 4096, 100, seed=1000)
s = solver.DdpSolver(4096, 100, b.p_max, np.float32)
g0 = s.solve(abi.phase0_params(), b)
b1 = b.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, b.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
g1 = s.solve(abi.phase1_params(iter_max=20, fixed_iters=1), b1)
print("launch info", s.launch_info(), s.last_kernel_ms())
PY
