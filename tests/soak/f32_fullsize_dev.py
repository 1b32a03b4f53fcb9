import sys, numpy as np
sys.path.insert(0, ".")
from direct_amd import abi, problems, solver
from oracle import refapi
B, N = 4096, 100
fb = problems.make_batch("free", B, N, seed=1000)
s = solver.DdpSolver(B, N, fb.p_max, np.float32)
p0, p1 = abi.phase0_params(), abi.phase1_params()
g0, g1 = s.plan(p0, p1, fb)
idx = np.arange(0, 4096, 128)
r0, r1 = refapi.plan_batch(p0, p1, fb.select(idx))
d = np.abs(g1.cost[idx] / r1.cost - 1)
print("free N=100 f32 vs oracle (32 problems): rtn same", (g1.rtn[idx] == r1.rtn).mean(), "cost dev quantiles 50/90/max", np.quantile(d, [.5, .9, 1.0]), "iter diff max", np.abs(g1.iter_used[idx] - r1.iter_used).max())
cb = problems.make_batch("corridor", B, N, seed=1000)
s2 = solver.DdpSolver(B, N, cb.p_max, np.float32)
h0, h1 = s2.plan(p0, p1, cb)
q0, q1 = refapi.plan_batch(p0, p1, cb.select(idx))
ok = (q1.rtn >= 0) & (h1.rtn[idx] >= 0)
d = np.abs(h1.cost[idx] / q1.cost - 1)[ok]
print("corridor N=100 f32 vs oracle: rtn same", (h1.rtn[idx] == q1.rtn).mean(), "phase0 rtn same", (h0.rtn[idx] == q0.rtn).mean(), "cost dev quantiles 50/90/max", np.quantile(d, [.5, .9, 1.0]))
