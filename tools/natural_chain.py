"""Natural-exit phase 1 of config 2 for nested subsets of the batch that all contain the slowest trajectory: how long the
SAME chain takes alone, among 64, 1024, 4096 (kernel ms by HIP events; iterations of the slowest).  usage (gpurun):
python tools/natural_chain.py"""
import json, sys
import numpy as np
sys.path.insert(0, ".")
from direct_amd import abi, problems, solver
B, N = 4096, 100
sb = problems.make_batch("free", B, N, seed=1000).astype(np.float32)
s = solver.DdpSolver(B, N, int(sb.p_max), np.float32)
g0 = s.solve(abi.phase0_params(), sb)
b1 = sb.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, sb.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
p1 = abi.phase1_params()
g = s.solve(p1, b1)
s.close()
lag = int(np.argmax(g.fwd_passes))
order = np.argsort(-g.fwd_passes)
out = {"slowest": lag, "its_iterations": int(g.iter_used[lag]), "its_fwd_passes": int(g.fwd_passes[lag])}
for n in (1, 8, 64, 512, 1024, 2048, 3072, 4096):
    idx = np.unique(np.r_[lag, np.arange(n - 1) if n > 1 else []].astype(int))[:n] if n < B else np.arange(B)
    if lag not in idx: idx[-1] = lag
    sub = b1.select(idx)
    s2 = solver.DdpSolver(len(idx), N, int(sb.p_max), np.float32)
    s2.solve(p1, sub)
    ms = []
    for _ in range(3):
        r = s2.solve(p1, sub)
        ms.append(s2.last_kernel_ms()[0])
    out["B=%d" % n] = {"kernel_ms": min(ms), "iterations_max": int(r.iter_used.max()), "ms_per_iteration_of_the_slowest": min(ms) / int(r.iter_used.max())}
    s2.close()
print(json.dumps(out))
