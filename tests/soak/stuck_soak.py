"""Soak of the rtn = -4 path (not a pytest test): tests/stuck_lib.py's forced scenario over many batches, injection depths and
both storage types on the device.  Writes gpurun_out/<ROUND>_stuck_soak.json.  usage: python tests/soak/stuck_soak.py [n_seeds]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from direct_amd import abi, problems, solver  # noqa: E402
from tests import stuck_lib  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rows = []
tot = dict(problems=0, usable=0, accepted=0, barrier_moved=0, decisions_wrong=0)
worst = {"float64": 0.0, "float32": 0.0}
for seed in range(500, 500 + n_seeds):
    for kind, N in (("corridor", 10), ("free", 8), ("corridor", 16)):
        for K in (2, 3, 5, 8):
            batch = problems.make_batch(kind, 8, N, seed=seed)
            p = abi.phase0_params(fixed_iters=1)
            try:
                sc = stuck_lib.Scenario(p, batch, K, -1e-8)
            except AssertionError:
                continue   # no row with a positive stored dual gain at the chosen knot
            if not all(s["rtn"] == -4 for s in sc.sc):
                sc.close()
                continue
            rec = dict(seed=seed, kind=kind, N=N, K=K, usable=int(sc.usable.sum()), accepted=int(sc.accepted().sum()))
            for dtype in (np.float64, np.float32):
                s = solver.DdpSolver(8, N, batch.p_max, dtype)
                s.begin(p, batch)
                out = sc.run(s)
                s.close()
                u = sc.usable
                wrong = int(sum(out["step"][i] != sc.sc[i]["step"] or out["fp_failed"][i] != sc.sc[i]["fp_failed"] for i in range(8) if u[i]))
                dev = max(float(out["dev"][n][u].max()) if u.any() else 0.0 for n in ("X", "U", "S", "Y"))
                name = np.dtype(dtype).name
                rec[name] = dict(decisions_wrong=wrong, max_dev=dev, rtn_all_minus4=bool((out["rtn"] == -4).all()))
                if wrong == 0:
                    worst[name] = max(worst[name], dev)
                if dtype == np.float64:
                    tot["decisions_wrong"] += wrong
            tot["problems"] += 8
            tot["usable"] += rec["usable"]
            tot["accepted"] += rec["accepted"]
            rows.append(rec)
            sc.close()
rep = dict(protocol="tests/stuck_lib.py: 8 problems per batch, K iterations, one dual entry of the middle knot made negative, the stuck trip; "
                    "device (double / float storage) against the oracle", totals=tot, worst_iterate_deviation_where_decisions_agree=worst,
           scenarios=len(rows), rows=rows)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rep, open("gpurun_out/%s_stuck_soak.json" % os.environ.get("ROUND", "r06"), "w"), indent=1)
print(json.dumps(dict(totals=tot, worst=worst, scenarios=len(rows))))
