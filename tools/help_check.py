"""Shared line search on/off: bitwise comparison of every output and kernel time (DIRECT_DDP_HELP is read at create).
usage: help_check.py [B] [kind] [modes, e.g. 0,1] [f32|f64] [N]"""
import os, sys
import numpy as np
sys.path.insert(0, ".")
from direct_amd import abi, problems, solver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kind = sys.argv[2] if len(sys.argv) > 2 else "free"
DT = np.float64 if (len(sys.argv) > 4 and sys.argv[4] == "f64") else np.float32
N = int(sys.argv[5]) if len(sys.argv) > 5 else 100
b = problems.make_batch(kind, B, N, seed=1000)
res = {}
MODES = sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "1"]
for mode in MODES:
    os.environ["DIRECT_DDP_HELP"] = mode
    s = solver.DdpSolver(B, N, b.p_max, DT)
    g0 = s.solve(abi.phase0_params(), b)
    b1 = b.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, b.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
    pf = abi.phase1_params(iter_max=20, fixed_iters=1)
    ts = []
    for _ in range(5):
        g1 = s.solve(pf, b1)
        ts.append(s.last_kernel_ms()[0])
    g2 = s.solve(abi.phase1_params(), b1)   # natural exits
    res[mode] = (g0, g1, g2)
    print("help=%s: fixed-20 kernel ms min %.2f med %.2f | sched_error %d | fwd passes %d" % (
        mode, min(ts), float(np.median(ts)), s.sched_error(), int(g1.fwd_passes.sum())), flush=True)
    del s
bad = 0
for i, nm in enumerate(("phase0", "fixed20", "natural")):
    a, c = res[MODES[0]][i], res[MODES[-1]][i]
    for f in ("rtn", "iter_used", "fwd_passes", "cost", "costq", "T", "poly", "bez", "opterr", "mu"):
        x, y = getattr(a, f), getattr(c, f)
        if x is None: continue
        if not np.array_equal(np.asarray(x).view(np.uint8), np.asarray(y).view(np.uint8)):
            bad += 1
            print("DIFF", nm, f, int((np.asarray(x) != np.asarray(y)).sum()))
print("bitwise identical" if not bad else "MISMATCH %d" % bad)
