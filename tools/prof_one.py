"""One timed workload for rocprofv3: config 2 (free space, B x N, fp32|fp64), fixed-iteration phase 1."""
import sys
import numpy as np
sys.path.insert(0, ".")
from direct_amd import abi, problems, solver
kind = sys.argv[1] if len(sys.argv) > 1 else "free"
dt = np.float32 if (len(sys.argv) < 3 or sys.argv[2] == "f32") else np.float64
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
N = int(sys.argv[4]) if len(sys.argv) > 4 else 100
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
b = problems.make_batch(kind, B, N, seed=1000)
s = solver.DdpSolver(B, N, b.p_max, dt)
g0 = s.solve(abi.phase0_params(), b)
b1 = b.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, b.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
pf = abi.phase1_params(iter_max=iters, fixed_iters=1)
g1 = s.solve(pf, b1)
ms, _ = s.last_kernel_ms()
print("kernel ms", ms, "iters", int(g1.fwd_passes.sum()), "M iter/s", g1.fwd_passes.sum() / ms / 1e3)
