import sys, numpy as np
sys.path.insert(0, ".")
from direct_amd import abi, solver
from tests import helpers
for name in ["free_n5", "corridor_n8", "corridor_n8_minvo", "free_n6_tp1", "corridor_n20", "config1_n50"]:
    g, batch = helpers.load_case(name)
    p0, p1 = helpers.case_params(name)
    s = solver.DdpSolver(batch.batch, batch.n_seg_max, batch.p_max, np.float32)
    f0 = s.solve(p0, batch)
    b1 = helpers.phase1_batch(g, batch)
    f1 = s.solve(p1, b1.with_init(None, T0=b1.T0, infeas_in=b1.infeas_in, init_poly=g["p0_poly"]))
    print(name, "p0 cost dev %.2e" % np.abs(f0.cost / g["p0_cost"] - 1).max(), "p1 rtn same", (f1.rtn == g["p1_rtn"].astype(int)).all(),
          "p1 cost dev", np.abs(f1.cost / g["p1_cost"] - 1), "T dev %.2e" % helpers.rel(f1.T, g["p1_T"]), "iters", f1.iter_used, g["p1_iter_used"].astype(int))
    s.close()
