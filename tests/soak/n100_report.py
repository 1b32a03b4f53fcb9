"""N = 100 parity report (tests/n100_lib.py) -> gpurun_out/<ROUND>_n100_parity.json (environment ROUND, default r06).  Run on the GPU box:
python tests/soak/n100_report.py [n_sample].  The bounded form of the same comparisons is tests/test_gpu_n100.py."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from direct_amd import abi, solver
from tests import n100_lib, soak_lib
from tests.test_gpu_soak import DeviceStepper

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def dev(dtype):
    def solve(params, batch):
        s = solver.DdpSolver(batch.batch, batch.n_seg_max, batch.p_max, dtype)
        r = s.solve(params, batch)
        s.close()
        return r
    return solve


rep = {}
t0 = time.time()
for name, kind, B, first in (("config2_free", "free", 4096, 0), ("config3_corridor", "corridor", 4096, 0),
                             ("config5_shard3", "corridor", 16384, 3 * 16384)):
    idx = np.arange(0, B, B // n)
    rep[name] = n100_lib.sample_report(kind, B, 100, idx, dev(np.float64), dev(np.float32), first=first)
    print(name, "%.0f s" % (time.time() - t0), json.dumps(rep[name]), flush=True)

# the timed launch of bench.py, stepped next to the oracle
for name, kind in (("timed_launch_config2", "free"), ("timed_launch_config3", "corridor")):
    idx = np.arange(0, 4096, 4096 // 32)
    b1, pf = n100_lib.timed_launch_inputs(kind, 4096, 100, idx, dev(np.float32))
    b64 = b1.astype(np.float64)
    o = soak_lib.OracleStepper(pf, b64)
    ref = n100_lib.stepped(o, 20)
    o.close()
    rec = {}
    for tag, dt, bb in (("f64", np.float64, b64), ("f32", np.float32, b1)):
        d = DeviceStepper(pf, bb, dt)
        first, devn = n100_lib.compare_stepped(n100_lib.stepped(d, 20), ref)
        d.close()
        rec[tag] = dict(identical_decisions=int((first < 0).sum()), n=int(len(first)),
                        first_flip=[int(x) for x in first],
                        cost_dev_final_q50_q90_max=[float("%.3g" % x) for x in np.quantile(devn[-1], [0.5, 0.9, 1.0])],
                        cost_dev_final_identical_max=float(devn[-1][first < 0].max()) if (first < 0).any() else None,
                        cost_dev_iter1_max=float(devn[1].max()))
    c = soak_lib.OracleStepper(pf, soak_lib.perturb_ulp(b64, 77))
    first, devn = n100_lib.compare_stepped(n100_lib.stepped(c, 20), ref)
    c.close()
    rec["control_double_ulp"] = dict(identical_decisions=int((first < 0).sum()), first_flip=[int(x) for x in first],
                                     cost_dev_final_q50_q90_max=[float("%.3g" % x) for x in np.quantile(devn[-1], [0.5, 0.9, 1.0])])
    c = soak_lib.OracleStepper(pf, n100_lib.perturb_float_ulp(b64, 78))
    first, devn = n100_lib.compare_stepped(n100_lib.stepped(c, 20), ref)
    c.close()
    rec["control_float_ulp"] = dict(identical_decisions=int((first < 0).sum()), first_flip=[int(x) for x in first],
                                    cost_dev_final_q50_q90_max=[float("%.3g" % x) for x in np.quantile(devn[-1], [0.5, 0.9, 1.0])])
    rep[name] = rec
    print(name, "%.0f s" % (time.time() - t0), json.dumps(rec), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rep, open("gpurun_out/%s_n100_parity.json" % os.environ.get("ROUND", "r06"), "w"), indent=1)
