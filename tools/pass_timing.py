"""Times one backward sweep and one forward pass separately (stepwise C-ABI) on a full-size batch."""
import sys
import numpy as np
sys.path.insert(0, ".")
from direct_amd import abi, problems, solver
kind = sys.argv[1] if len(sys.argv) > 1 else "free"
dt = np.float32 if (len(sys.argv) < 3 or sys.argv[2] == "f32") else np.float64
B, N = 4096, 100
b = problems.make_batch(kind, B, N, seed=1000)
s = solver.DdpSolver(B, N, b.p_max, dt)
g0 = s.solve(abi.phase0_params(), b)
b1 = b.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, b.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
pf = abi.phase1_params(iter_max=20, fixed_iters=1)
s.begin(pf, b1)
for it in range(6):
    s.backward(); tb, _ = s.last_kernel_ms()
    s.forward(); tf, _ = s.last_kernel_ms()
    sc = s.scalars()
    steps = sc["step"].astype(int)
    print("iter %d: backward %.2f ms, forward %.2f ms (mean accepted step index %.2f, max %d, failed %d)"
          % (it, tb, tf, steps.mean(), steps.max(), int(sc["fp_failed"].sum())), flush=True)
