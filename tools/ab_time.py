"""Repeated timing of the bench workload for A/B comparisons: prints min / median kernel ms over R launches.
usage: ab_time.py [kind] [f32|f64] [R] [B]   (library chosen by DIRECT_DDP_LIB)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from direct_amd import abi, problems, solver
kind = sys.argv[1] if len(sys.argv) > 1 else "free"
dt = np.float32 if (len(sys.argv) < 3 or sys.argv[2] == "f32") else np.float64
R = int(sys.argv[3]) if len(sys.argv) > 3 else 7
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
b = problems.make_batch(kind, B, 100, seed=1000)
s = solver.DdpSolver(B, 100, b.p_max, dt)
g0 = s.solve(abi.phase0_params(), b)
ms0, _ = s.last_kernel_ms()
b1 = b.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, b.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
pf = abi.phase1_params(iter_max=20, fixed_iters=1)
ts = []
for _ in range(R):
    g1 = s.solve(pf, b1)
    ts.append(s.last_kernel_ms()[0])
its = int(g1.fwd_passes.sum())
ts = np.array(ts)
print("%s %s B=%d: phase0 %.1f ms | fixed-20 min %.2f med %.2f max %.2f ms -> %.3f M iter/s (min) %.3f (med) | cost sum %.9e"
      % (kind, np.dtype(dt).name, B, ms0, ts.min(), np.median(ts), ts.max(), its / ts.min() / 1e3, its / np.median(ts) / 1e3, float(g1.cost.sum())))
