"""One timed workload: config 2 (free space, B x N, fp32|fp64), phase 1 with NATURAL exits (bench.py's natural_exit block).
usage: prof_natural.py [kind] [f32|f64] [B] [N]; prints kernel ms, iterations, a sha1 over every output array."""
import hashlib, sys
import numpy as np
sys.path.insert(0, ".")
from direct_amd import abi, problems, solver
kind = sys.argv[1] if len(sys.argv) > 1 else "free"
dt = np.float32 if (len(sys.argv) < 3 or sys.argv[2] == "f32") else np.float64
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
N = int(sys.argv[4]) if len(sys.argv) > 4 else 100
b = problems.make_batch(kind, B, N, seed=1000)
s = solver.DdpSolver(B, N, b.p_max, dt)
g0 = s.solve(abi.phase0_params(), b)
b1 = b.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, b.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
for _ in range(2):
    g1 = s.solve(abi.phase1_params(), b1)
ms, _ = s.last_kernel_ms()
h = hashlib.sha1()
for f in ("rtn", "iter_used", "fwd_passes", "infeas_out", "cost", "costq", "opterr", "mu", "T", "poly", "bez"):
    h.update(np.ascontiguousarray(getattr(g1, f)).tobytes())
li = s.launch_info()
print("kernel ms %.3f iters %d M iter/s %.4f sweep %d helper_knots %d err %d sha1 %s" % (
    ms, int(g1.fwd_passes.sum()), g1.fwd_passes.sum() / ms / 1e3, li["shared_sweep"], li.get("helper_front_knots", 0), s.sched_error(), h.hexdigest()[:12]))
