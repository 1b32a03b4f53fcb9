"""Generates tests/golden/*.npz from the NumPy restatement (oracle/ddp_numpy.py).

Run in the build container:  PYTHONPATH=. python tests/golden/make_golden.py
The reference itself cannot be executed here (no Eigen/ROS), so these vectors pin the
INDEPENDENT NumPy restatement; the C oracle, the lane-loop emulator and the HIP kernels are all tested
against them.  Each case stores the inputs in the flat layout of include/direct_ddp.h, the
per-iteration trace (cost, costq, logcost, err, mu, reg, step, opterr, stepsize, fp_failed) and the
getter outputs of both phases of fastTrajPlanning's protocol (teach_repeat_planner.cpp:886-921).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from direct_amd import abi, problems  # noqa: E402
from oracle import ddp_numpy  # noqa: E402

OUT_KEYS = ("rtn", "iter_used", "fwd_passes", "cost", "costq", "jerk_cost", "terminal_norm2", "opterr", "mu",
            "infeas_out")


def run_phase(batch, params):
    B, nm = batch.batch, batch.n_seg_max
    out = {k: np.zeros(B) for k in OUT_KEYS}
    out["bez"] = np.zeros((B, nm, 18)); out["poly"] = np.zeros((B, nm, 18)); out["T"] = np.zeros((B, nm))
    trace = np.full((B, params.iter_max, 10), np.nan)
    for b in range(B):
        d, res = ddp_numpy.solve_problem(batch, b, params)
        N = int(batch.n_seg[b])
        for k in OUT_KEYS:
            out[k][b] = res[k]
        out["bez"][b, :N], out["poly"][b, :N], out["T"][b, :N] = res["bez"], res["poly"], res["T"]
        tr = np.array(d.trace)
        trace[b, :len(tr)] = tr
    out["trace"] = trace
    return out


def make_case(name, batch, p0, p1):
    print("case", name, "B", batch.batch, "N", batch.n_seg_max, flush=True)
    r0 = run_phase(batch, p0)
    T1 = np.where((r0["rtn"] == 2)[:, None], r0["T"], batch.T0)   # UpdateTime, TRP:911-915
    b1 = batch.with_init(r0["bez"], T0=T1, infeas_in=r0["infeas_out"].astype(np.uint8))
    r1 = run_phase(b1, p1)
    d = dict(n_seg=batch.n_seg, x0=batch.x0, xd=batch.xd, T0=batch.T0, n_planes=batch.n_planes,
             planes=batch.planes, seeds=batch.seeds)
    d.update({"p0_" + k: v for k, v in r0.items()})
    d.update({"p1_" + k: v for k, v in r1.items()})
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz"), **d)
    print("  phase0 rtn", r0["rtn"], "iters", r0["iter_used"], "phase1 rtn", r1["rtn"], "iters", r1["iter_used"])


if __name__ == "__main__":
    p0, p1 = abi.phase0_params(), abi.phase1_params()
    make_case("free_n5", problems.make_batch("free", 3, 5, seed=7), p0, p1)
    make_case("corridor_n8", problems.make_batch("corridor", 3, 8, seed=7), p0, p1)
    make_case("corridor_n20", problems.make_batch("corridor", 2, 20, seed=7), p0, p1)
    make_case("config1_n50", problems.make_config1(), p0, p1)
    # non-default switches: MINVO basis, time_power 1
    make_case("corridor_n8_minvo", problems.make_batch("corridor", 2, 8, seed=9),
              abi.phase0_params(minvo=1), abi.phase1_params(minvo=1))
    make_case("free_n6_tp1", problems.make_batch("free", 2, 6, seed=9),
              abi.phase0_params(time_power=1), abi.phase1_params(time_power=1))
