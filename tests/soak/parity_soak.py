"""Randomised parity soak (GPU): many small batches, fp64, two-phase plan against the oracle.
Reports how often every discrete decision (return codes, iteration counts) matches and the worst
relative deviation of cost / durations where they do.  usage: parity_soak.py [n_batches]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from direct_amd import abi, problems, solver
from oracle import refapi
from tests import helpers

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(2024)
tot = same = 0
worst_cost = worst_T = 0.0
mism = []
for t in range(nb):
    kind = "corridor" if t % 3 else "free"
    N = int(rng.integers(3, 26))
    B = 32
    batch = problems.make_batch(kind, B, N, seed=int(rng.integers(1, 10 ** 6)))
    if t % 4 == 3:
        batch = helpers.with_extra_planes(batch, int(rng.integers(13, 33)), seed=t)
    p0, p1 = abi.phase0_params(), abi.phase1_params()
    s = solver.DdpSolver(B, N, batch.p_max, np.float64)
    g0, g1 = s.plan(p0, p1, batch)
    s.close()
    r0, r1 = refapi.plan_batch(p0, p1, batch)
    ok = (g0.rtn == r0.rtn) & (g0.iter_used == r0.iter_used) & (g1.rtn == r1.rtn) & (g1.iter_used == r1.iter_used)
    tot += B
    same += int(ok.sum())
    if ok.any():
        worst_cost = max(worst_cost, float(np.abs(g1.cost[ok] / r1.cost[ok] - 1).max()))
        worst_T = max(worst_T, float(np.abs(g1.T[ok] / np.where(r1.T[ok] == 0, 1, r1.T[ok]) - (r1.T[ok] != 0)).max()))
    for i in np.nonzero(~ok)[0]:
        mism.append((t, kind, N, int(batch.p_max), int(i), int(g0.iter_used[i]), int(r0.iter_used[i]), int(g1.iter_used[i]),
                     int(r1.iter_used[i]), float(abs(g1.cost[i] / r1.cost[i] - 1))))
print("problems %d, identical discrete decisions in both phases %d (%.2f%%); where identical: max rel cost dev %.2e, max rel T dev %.2e"
      % (tot, same, 100.0 * same / tot, worst_cost, worst_T))
for m in mism[:20]:
    print("  mismatch batch %d %s N=%d Pmax=%d problem %d: it0 %d/%d it1 %d/%d, final cost rel dev %.2e" % m)
