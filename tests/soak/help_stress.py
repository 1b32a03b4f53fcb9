"""Stress of the shared line search and of the shared backward sweep (not a pytest test: minutes of GPU time).  For several
batch sizes the fixed-20 launch is repeated many times with helpers - once with the library's own choice of what is shared,
once with the backward sweeps shared at EVERY batch size (DIRECT_DDP_BSHARE=1) -; every launch must reproduce the owner-only
result bit for bit and leave the scheduler's error flag clear.  usage: python tests/soak/help_stress.py [launches per size]"""
import os, sys
import numpy as np
sys.path.insert(0, ".")
from direct_amd import abi, problems, solver
R = int(sys.argv[1]) if len(sys.argv) > 1 else 40
FIELDS = ("rtn", "iter_used", "fwd_passes", "cost", "costq", "T", "poly", "bez", "opterr", "mu")
total = 0
for B, kind in ((1, "corridor"), (33, "free"), (700, "corridor"), (3100, "free"), (4096, "corridor"), (4600, "free")):
    b = problems.make_batch(kind, B, 100, seed=4000 + B).astype(np.float32)
    ref = None
    for mode, sweep in (("0", "0"), ("1", None), ("1", "1")):
        os.environ["DIRECT_DDP_HELP"] = mode
        if sweep is None:
            os.environ.pop("DIRECT_DDP_BSHARE", None)
        else:
            os.environ["DIRECT_DDP_BSHARE"] = sweep
        s = solver.DdpSolver(B, 100, b.p_max, np.float32)
        g0 = s.solve(abi.phase0_params(), b)
        b1 = b.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, b.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
        pf = abi.phase1_params(iter_max=20, fixed_iters=1)
        n = 1 if mode == "0" else R
        for i in range(n):
            g = s.solve(pf, b1) if i % 4 else s.solve(abi.phase1_params(iter_max=60), b1)   # every fourth: natural exits
            key = "nat" if i % 4 == 0 else "fix"
            assert s.sched_error() == 0, (B, i)
            if mode == "0":
                ref = {"nat": g, "fix": s.solve(pf, b1)}
            else:
                for f in FIELDS:
                    assert np.array_equal(getattr(ref[key], f), getattr(g, f)), (B, i, key, f)
                total += 1
        s.close()
    print("B = %d (%s): 2 x %d launches with helpers (line search; + backward sweep) reproduce the owner-only launch" % (B, kind, R), flush=True)
print("ok: %d launches" % total)
