"""CPU soak of the kernel SOURCE (ddp_wave.h through the lane-loop emulator, tests/emu) against the oracle: random two-phase
plans, both corridor kinds, ragged lengths - return codes, iteration counts, objective, durations - and the forced-stuck
scenario (tests/stuck_lib.py) over further seeds.  Test infrastructure, no GPU, not a product path.
usage: PYTHONPATH=. python tests/soak/emu_soak.py [seeds per shape] [out.json]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from direct_amd import abi, problems  # noqa: E402
from oracle import refapi  # noqa: E402
from tests import helpers, soak_lib, stuck_lib  # noqa: E402
from tests.emu import emuapi  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/emu_soak.json"
rep = dict(plans=0, rtn_same=0, iters_same=0, iters_differ_where_the_oracle_flips_under_one_ulp=0, cost_max=0.0, T_max=0.0, bez_max=0.0, rtn_hist={}, stuck=dict(problems=0, accepted=0, worst=0.0), shapes=[])
t0 = time.time()
for kind in ("free", "corridor"):
    for N in (3, 5, 8, 12, 17, 24):
        for s in range(S):
            B = 4
            batch = problems.make_batch(kind, B, N, seed=9000 + 97 * N + s)
            n_seg = np.array([N, max(2, N - 1), max(2, N // 2), N], np.int32)  # ragged
            xd = batch.xd.copy()
            for i in range(B):
                xd[i, :3] = batch.seeds[i, n_seg[i] - 1] if batch.seeds is not None and n_seg[i] < N else xd[i, :3]
            rb = abi.HostBatch(n_seg, batch.x0, xd, batch.T0, batch.n_planes, batch.planes, seeds=batch.seeds)
            p0, p1 = abi.phase0_params(), abi.phase1_params()
            r0, _ = refapi.solve_batch(p0, rb)
            e0 = emuapi.solve_batch(p0, rb)
            b1 = rb.phase1_inputs(r0, monomial=False)
            r1, _ = refapi.solve_batch(p1, b1)
            e1 = emuapi.solve_batch(p1, rb.phase1_inputs(e0, monomial=False))
            for r, e, pp, bb in ((r0, e0, p0, rb), (r1, e1, p1, b1)):
                same = (r.rtn == e.rtn) & (r.iter_used == e.iter_used)
                if not same.all():  # control: does the ORACLE keep its own count when its inputs move by one ulp?
                    flips = np.zeros(B, bool)
                    for cs in (11, 12, 13, 14):
                        rc = refapi.solve_batch(pp, soak_lib.perturb_ulp(bb, cs))[0]
                        flips |= (rc.iter_used != r.iter_used) | (rc.rtn != r.rtn)
                    rep["iters_differ_where_the_oracle_flips_under_one_ulp"] += int((flips & ~same).sum())
                rep["plans"] += B
                rep["rtn_same"] += int((r.rtn == e.rtn).sum())
                rep["iters_same"] += int(same.sum())
                for v in r.rtn:
                    rep["rtn_hist"][str(int(v))] = rep["rtn_hist"].get(str(int(v)), 0) + 1
                if same.any():
                    rep["cost_max"] = max(rep["cost_max"], float(np.abs(e.cost[same] / r.cost[same] - 1).max()))
                    for i in np.nonzero(same)[0]:
                        n = int(n_seg[i])
                        rep["T_max"] = max(rep["T_max"], helpers.rel(e.T[i, :n], r.T[i, :n]))
                        rep["bez_max"] = max(rep["bez_max"], helpers.rel(e.bez[i, :n], r.bez[i, :n]))
        rep["shapes"].append([kind, N, round(time.time() - t0, 1)])
        print(kind, N, rep["plans"], rep["rtn_same"], rep["iters_same"], "%.2e" % rep["cost_max"], flush=True)
for name, params, kind, K, y_inj, zero in stuck_lib.scenarios():
    for s in range(S):
        b = problems.make_batch(kind, 4, 10, seed=500 + 13 * s)
        if zero:
            b = b.with_init(np.zeros((b.batch, b.n_seg_max, 18)))
        try:
            sc = stuck_lib.Scenario(params, b, K, y_inj)
        except AssertionError:
            continue  # a problem without a usable dual gain at the knot: not a scenario
        e = emuapi.EmuSolver(params, b)
        o = sc.run(e)
        sc.check(o, 1e-9)
        rep["stuck"]["problems"] += b.batch
        rep["stuck"]["accepted"] += int(sc.accepted().sum())
        u = sc.usable
        if u.any():
            rep["stuck"]["worst"] = max(rep["stuck"]["worst"], float(max(o["dev"][n][u].max() for n in "XUSY")))
        sc.close()
    print("stuck", name, rep["stuck"], flush=True)
rep["seconds"] = round(time.time() - t0, 1)
json.dump(rep, open(out, "w"), indent=1)
print(json.dumps(rep))
