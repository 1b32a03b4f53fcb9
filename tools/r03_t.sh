cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q 2>&1 | grep -E "^FAILED|passed|failed" | head -12
python - <<'PY'
import numpy as np, sys, time
sys.path.insert(0, ".")
from direct_amd import abi, problems, solver
for kind in ("free", "corridor"):
    b = problems.make_batch(kind, 4096, 100, seed=1000)
    s = solver.DdpSolver(4096, 100, b.p_max, np.float32)
    g0 = s.solve(abi.phase0_params(), b)
    b1 = b.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, b.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
    for rep in range(3):
        g1 = s.solve(abi.phase1_params(), b1)
        ms, _ = s.last_kernel_ms()
    print(kind, "natural exits: kernel %.2f ms, %.3f M iter/s, iters max %d" % (ms, g1.fwd_passes.sum() / ms / 1e3, g1.fwd_passes.max()), s.launch_info())
PY
