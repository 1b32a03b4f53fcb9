"""CPU soak of the two witnesses against each other: the C oracle (oracle/direct_ref.c) and the independent NumPy restatement
(oracle/ddp_numpy.py) on random two-phase plans - return codes, iteration counts, objective, durations; both time powers, MINVO on and off.  The oracle is
PARITY UNPINNED (no reference-held vector exists): agreement of two restatements written from the reference's text
is the strongest pin available.  Test infrastructure.  usage: PYTHONPATH=. python tests/soak/oracle_cross_soak.py [seeds] [out.json]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from direct_amd import abi, problems  # noqa: E402
from oracle import ddp_numpy, refapi  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/oracle_cross_soak.json"
rep = dict(solves=0, rtn_same=0, iters_same=0, cost_max=0.0, T_max=0.0, rtn_hist={})
t0 = time.time()
for kind in ("free", "corridor"):
    for N in (3, 5, 8, 12):
        for s in range(S):
            batch = problems.make_batch(kind, 2, N, seed=7000 + 31 * N + s)
            for tp, minvo in ((2, 0), (1, 0), (2, 1)):
                p0, p1 = abi.phase0_params(time_power=tp, minvo=minvo), abi.phase1_params(time_power=tp, minvo=minvo)
                r0 = refapi.solve_batch(p0, batch)[0]
                b1 = batch.phase1_inputs(r0, monomial=False)
                r1 = refapi.solve_batch(p1, b1)[0]
                for p, bb, r in ((p0, batch, r0), (p1, b1, r1)):
                    for b in range(bb.batch):
                        d, res = ddp_numpy.solve_problem(bb, b, p)
                        rep["solves"] += 1
                        same_r = int(res["rtn"]) == int(r.rtn[b])
                        same_i = same_r and int(res["iter_used"]) == int(r.iter_used[b])
                        rep["rtn_same"] += same_r
                        rep["iters_same"] += same_i
                        k = str(int(r.rtn[b]))
                        rep["rtn_hist"][k] = rep["rtn_hist"].get(k, 0) + 1
                        if same_i:
                            rep["cost_max"] = max(rep["cost_max"], abs(res["cost"] / r.cost[b] - 1))
                            rep["T_max"] = max(rep["T_max"], float(np.abs(res["T"] / r.T[b, :N] - 1).max()))
        print(kind, N, rep, round(time.time() - t0, 1), flush=True)
rep["seconds"] = round(time.time() - t0, 1)
json.dump(rep, open(out, "w"), indent=1)
print(json.dumps(rep))
