"""Generates tests/golden/exit_*.npz: one fixture per way OUT of the outer loop (ddp_optimizer.cpp:295-412) and per
failure branch inside it, from the NumPy restatement (oracle/ddp_numpy.py) - the second witness of an oracle that nothing
reference-held pins.  The committed DDP fixtures of make_golden.py all leave through rtn 2 (phase 0) and
rtn 1 (phase 1); these cover the rest:

  exit_iter_max          the for loop runs out (DDP:295; rtn 0, iter_used = iter_max)
  exit_neg_time          a negative duration after forwardpass() (DDP:317-326; rtn -3) - reachable through a negative
                         INPUT duration only: the fraction-to-boundary rule on the T_min row keeps every accepted T above 0.3
  exit_stuck_first       backward pass stuck in the first iteration (DDP:297-310, 392-396; rtn -4), zero gains (DDP:154-159)
  exit_llt_retry         LLT failures inside a sweep with recovery at a larger regulariser (DDP:546-551 / 595-600, 452-474)
  exit_line_ok           line_init_flag: feasible and stagnating (DDP:381-388; line_failed = false)
  exit_line_no_update    line_init_flag: 101 iterations without a step (DDP:398-409)
  exit_forced_stuck      backward pass stuck MID-SOLVE with an accepted step of the stale-gain forward pass (tests/stuck_lib.py:
                         natural mid-solve rtn = -4 solves are chaotic, so the situation is built through the stepwise interface)

NOT reachable by any input found (tools search of round 6 over ~40 k solves): the optimality exit DDP:335-338.  It needs
max(opterr, mu) <= 1e-7, i.e. ten barrier updates of a converging solve - and every converging solve is feasible and
stagnating long before, which is rtn 2 (zero init), rtn 1 (DDP:374) or the line-init exit (DDP:381), all tested first.

Run in the build container:  PYTHONPATH=. python tests/golden/make_exit_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from direct_amd import abi, problems  # noqa: E402
from oracle import ddp_numpy  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
PARAM_FIELDS = ("max_vel", "max_acc", "w_snap", "w_terminal", "w_time", "iter_max", "time_power", "zero_init", "line_init",
                "minvo", "infeas", "fixed_iters", "exact_dt")
OUT_KEYS = ("rtn", "iter_used", "fwd_passes", "cost", "costq", "jerk_cost", "terminal_norm2", "opterr", "mu", "infeas_out",
            "line_failed_out")


def take(batch, idx, **kw):
    """rows idx of a HostBatch (optionally with replaced arrays)"""
    idx = np.asarray(idx)
    f = lambda a: None if a is None else np.ascontiguousarray(a[idx])
    d = dict(seeds=f(batch.seeds), init_bez=f(batch.init_bez), infeas_in=f(batch.infeas_in), init_poly=f(batch.init_poly))
    return abi.HostBatch(batch.n_seg[idx], batch.x0[idx], batch.xd[idx], batch.T0[idx], batch.n_planes[idx], batch.planes[idx], **d)


def inputs_dict(batch, params):
    d = dict(n_seg=batch.n_seg, x0=batch.x0, xd=batch.xd, T0=batch.T0, n_planes=batch.n_planes, planes=batch.planes,
             seeds=batch.seeds)
    if batch.init_bez is not None:
        d["init_bez"] = batch.init_bez
    if batch.infeas_in is not None:
        d["infeas_in"] = batch.infeas_in
    for f in PARAM_FIELDS:
        d["param_" + f] = np.float64(getattr(params, f))
    return d


def solve_case(name, batch, params, expect):
    B, nm = batch.batch, batch.n_seg_max
    out = {k: np.zeros(B) for k in OUT_KEYS}
    out["bez"] = np.zeros((B, nm, 18)); out["poly"] = np.zeros((B, nm, 18)); out["T"] = np.zeros((B, nm))
    trace = np.full((B, params.iter_max + 1, 12), np.nan)
    for b in range(B):
        d, res = ddp_numpy.solve_problem(batch, b, params)
        N = int(batch.n_seg[b])
        for k in OUT_KEYS:
            out[k][b] = res[k]
        out["bez"][b, :N], out["poly"][b, :N], out["T"][b, :N] = res["bez"], res["poly"], res["T"]
        tr = np.array(d.trace)
        trace[b, :len(tr)] = tr
    out["trace"] = trace
    expect(out)
    d = inputs_dict(batch, params)
    d.update({"out_" + k: v for k, v in out.items()})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, "rtn", out["rtn"], "iter_used", out["iter_used"], "line_failed", out["line_failed_out"], flush=True)


def forced_stuck_case(name, batch, params, K, y_inject):
    """tests/stuck_lib.py's scenario on the NumPy restatement"""
    B, N = batch.batch, int(batch.n_seg[0])
    knot = N // 2
    ncm = 6 * batch.p_max + 55
    post = dict(X=np.zeros((B, N + 1, 9)), U=np.zeros((B, N, 10)), S=np.zeros((B, N, ncm)), Y=np.zeros((B, N, ncm)))
    sc = {k: np.zeros(B) for k in ("rtn", "step", "fp_failed", "reg", "cost", "stepsize", "n_sweeps", "usable", "mu")}
    rows = np.zeros(B, np.int32)
    for b in range(B):
        d = ddp_numpy.make_problem(batch, b, params)
        for _ in range(K):
            assert not d.iterate_once()
            d.iter += 1
        sc["usable"][b] = float(not d.fp_failed)
        r = int(np.argmax(d.ky[knot]))
        assert d.ky[knot][r] > 1e-6
        rows[b] = r
        d.y[knot][r] = y_inject
        done = d.iterate_once()
        assert done and d.rtn == -4 and d.iter == K
        for k in range(N):
            nc = d.c[k].size
            post["U"][b, k] = d.u[k]
            post["S"][b, k, :nc] = d.s[k]
            post["Y"][b, k, :nc] = d.y[k]
        for k in range(N + 1):
            post["X"][b, k] = d.x[k]
        t = d.trace[-1]
        sc["rtn"][b], sc["step"][b], sc["fp_failed"][b], sc["reg"][b] = d.rtn, d.step, float(d.fp_failed), d.reg
        sc["cost"][b], sc["stepsize"][b], sc["n_sweeps"][b], sc["mu"][b] = d.cost, d.stepsize, t[10], d.mu
    assert (sc["reg"] == 24).all() and (sc["n_sweeps"] >= 21 + 18).all()   # up to reg 24, then 21 sweeps there (DDP:297-310)
    assert ((sc["fp_failed"] == 0) & (sc["usable"] == 1)).sum() >= 3
    out = inputs_dict(batch, params)
    out.update(K=np.int32(K), knot=np.int32(knot), rows=rows, y_inject=np.float64(y_inject))
    out.update({"post_" + k: v for k, v in post.items()})
    out.update({"sc_" + k: v for k, v in sc.items()})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "accepted", (sc["fp_failed"] == 0).astype(int), "step", sc["step"], flush=True)


if __name__ == "__main__":
    z = lambda b: np.zeros((b.batch, b.n_seg_max, 18))

    b = problems.make_batch("free", 3, 3, seed=300)
    solve_case("exit_iter_max", b.with_init(z(b)), abi.phase1_params(iter_max=6),
               lambda o: (o["rtn"] == 0).all() and (o["iter_used"] == 6).all() or sys.exit("iter_max"))

    b = problems.make_batch("free", 4, 5, seed=5)
    T0 = b.T0.copy(); T0[1, 2] = -0.7; T0[3, 0] = -1e-3
    solve_case("exit_neg_time", b.with_init(None, T0=T0), abi.phase0_params(),
               lambda o: (o["rtn"][1] == -3 and o["rtn"][3] == 2) or sys.exit("neg_time"))

    b = problems.make_batch("corridor", 32, 8, seed=1)
    bb = take(b.with_init(z(b), T0=b.T0 * 3.0, infeas_in=np.zeros(32, np.uint8)), [17, 18, 19])
    solve_case("exit_stuck_first", bb, abi.phase1_params(iter_max=60),
               lambda o: (o["rtn"][1] == -4 and o["iter_used"][1] == 0 and o["trace"][1, 0, 10] == 45) or sys.exit("stuck_first"))

    b = problems.make_batch("free", 16, 3, seed=300)
    bb = take(b.with_init(z(b), T0=b.T0 * 0.4), [1, 2, 3])
    solve_case("exit_llt_retry", bb, abi.phase1_params(),
               lambda o: ((np.nan_to_num(o["trace"][:, :, 10]) > 1).any(axis=1).all() and (o["rtn"] == 1).all()) or sys.exit("llt_retry"))

    b = problems.make_batch("free", 16, 2, seed=301)
    bb = take(b.with_init(z(b)), [5, 6, 7])
    solve_case("exit_line_ok", bb, abi.phase1_params(line_init=1, infeas=1, iter_max=140),
               lambda o: ((o["line_failed_out"] == 0).all() and o["iter_used"][1] == 3) or sys.exit("line_ok"))

    b = problems.make_batch("corridor", 16, 6, seed=300)
    bb = take(b.with_init(z(b), T0=b.T0 * 0.03), [0, 1])
    solve_case("exit_line_no_update", bb, abi.phase1_params(line_init=1, infeas=1, iter_max=140),
               lambda o: ((o["line_failed_out"] == 1).all() and (o["iter_used"] == 100).all() and (o["rtn"] == 0).all()) or sys.exit("line_no_update"))

    forced_stuck_case("exit_forced_stuck", problems.make_batch("corridor", 8, 10, seed=77), abi.phase0_params(), 3, -1e-8)
