"""Label model timing: B trajectories, N = 100, fixed 10 iterations (float storage).  usage: quad_bench.py [B]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from direct_amd import quad
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x0, xg = quad.label_problems(B, seed=1000)
p = quad.default_params(iter_max=10, fixed_iters=1)
q = quad.QuadSolver(B, 100, np.float32)
for _ in range(3):
    r = q.solve(p, x0, xg)
    ms = q.last_kernel_ms()
print("B", B, "kernel ms %.2f" % ms, "M iter/s %.3f" % (r["iters"].sum() / ms / 1e3), "cost sum %.6e" % r["cost"].sum())
