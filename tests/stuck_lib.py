"""The backward-pass-stuck exit (rtn = -4, DDP:297-311, 392-396) on WELL-CONDITIONED states.  TEST INFRASTRUCTURE, shared by
tests/test_emu_parity.py (lane-loop emulator, CPU) and tests/test_gpu_parity.py (device).

Why forced.  Natural mid-solve rtn = -4 solves exist (phase 0 with w_time = 20: a few per hundred corridors, after 27 .. 50
iterations) but every one of them is chaotic by the time it gets stuck: the ORACLE ITSELF flips a decision some iterations
before the end when its inputs move by one ulp (tools run of round 6: 70 of 70).  They cannot pin the reference's forward
pass after an aborted retry sequence to 1e-9.  This module builds the same situation a few iterations into a solve,
through the public stepwise interface and identically for every implementation:

  1. K outer iterations (oracle and implementation agree to 1e-12 there);
  2. one dual entry y[k*][r] is set to a tiny NEGATIVE number (direct_ddp_set_field / direct_ref_set): the condensed
     matrix  Quu_reg + cu' diag(s / y) cu  (DDP:540-543) of knot k* is indefinite for every regulariser, so each sweep of
     the next iteration's retry loop reaches knots N-1 .. k*+1 and aborts at k* - the loop gives up after 24 + 21 sweeps
     (DDP:297-310) with knots 0 .. k* still holding the gains of iteration K-1's sweep: another iterate, possibly
     another barrier parameter;
  3. row r is the row of knot k* whose stored dual gain ky is largest: the reference's trial value
     y+ = y + alpha ky + Ky dx (DDP:681) is positive again, the fraction-to-boundary rule passes (DDP:684), and the stale
     forward pass ACCEPTS a step on most problems;
  4. the iterate after that last trip, the step index and the return code are compared.

The round-5 kernels (slack / dual gains regenerated from the current iterate only) leave the oracle by 3 - 8 % in cost and
choose other step sizes on exactly the problems that accept a step."""
import numpy as np

from direct_amd import abi
from oracle import refapi
from tests import helpers


class Scenario:
    def __init__(self, params, batch, K, y_inject, knot=None):
        self.params, self.batch, self.K, self.y_inject = params, batch, K, y_inject
        self.B, self.N = batch.batch, int(batch.n_seg[0])
        self.knot = self.N // 2 if knot is None else knot
        self.oracle = [refapi.Stepper(params, batch, i) for i in range(self.B)]
        self.mu_before = []   # barrier parameter of the LAST COMPLETED backward sweep (what the stale gains were formed with)
        for q in self.oracle:
            if K > 1:
                q.iterate(K - 1)
            self.mu_before.append(q.scalars()["mu"])
            q.iterate(1)
        self.mu_now = [q.scalars()["mu"] for q in self.oracle]
        # A line search that failed in iteration K-1 leaves the iterate where the gains were formed: the injection below
        # would then change the very iterate the implementation regenerates the stale gains from, while the oracle's stored
        # gains keep the old value - an artefact of injecting (no solver rewrites an iterate between two passes), so
        # those problems only check the return code.
        self.usable = np.array([q.scalars()["fp_failed"] == 0 for q in self.oracle])
        self.rows = []
        for i, q in enumerate(self.oracle):
            nc = 6 * int(batch.n_planes[i, self.knot]) + 55
            ky = q.get(abi.FIELD_KY)[self.knot][:nc]
            r = int(np.argmax(ky))
            assert ky[r] > 1e-6, (i, ky[r])
            self.rows.append(r)
            Y = q.get(abi.FIELD_Y)
            Y[self.knot, r] = y_inject
            q.set(abi.FIELD_Y, Y)
        self.pre = [dict(X=q.get(abi.FIELD_X), U=q.get(abi.FIELD_U), S=q.get(abi.FIELD_S), Y=q.get(abi.FIELD_Y)) for q in self.oracle]
        for q in self.oracle:
            q.iterate(1)
        self.sc = [q.scalars() for q in self.oracle]

    def accepted(self):
        """problems whose stale forward pass accepted a step (the others leave the iterate where it was)"""
        return np.array([s["fp_failed"] == 0 and s["stepsize"] > 0 for s in self.sc]) & self.usable

    def barrier_moved(self):
        """problems whose barrier parameter changed between the sweep that formed the stale gains and the stuck trip"""
        return np.array([a != b for a, b in zip(self.mu_before, self.mu_now)]) & self.usable

    def run(self, impl):
        """impl: begun on (params, batch); anything with iterate / get / set / scalars.  Returns per-problem deviations."""
        impl.iterate(self.K)
        pre = max(helpers.rel(impl.get(f)[i][:self.N + (f == abi.FIELD_X)], self.pre[i][n])
                  for i in range(self.B) for f, n in ((abi.FIELD_X, "X"), (abi.FIELD_U, "U")))
        Y = impl.get(abi.FIELD_Y)
        for i, r in enumerate(self.rows):
            Y[i, self.knot, r] = self.y_inject
        impl.set(abi.FIELD_Y, Y)
        impl.iterate(1)
        se = impl.scalars()
        out = dict(pre=pre, rtn=se["rtn"].astype(int), done=se["done"].astype(int), step=se["step"].astype(int),
                   fp_failed=se["fp_failed"].astype(int), reg=se["reg"].astype(int), dev={}, cost=[])
        for f, n in ((abi.FIELD_X, "X"), (abi.FIELD_U, "U"), (abi.FIELD_S, "S"), (abi.FIELD_Y, "Y")):
            got = impl.get(f)
            out["dev"][n] = np.array([helpers.rel(got[i][:self.N + (f == abi.FIELD_X)], self.oracle[i].get(f)) for i in range(self.B)])
        out["cost"] = np.array([abs(se["cost"][i] / self.sc[i]["cost"] - 1) for i in range(self.B)])
        return out

    def check(self, out, tol):
        """the oracle's last trip, decision by decision, and the iterate it leaves"""
        assert out["pre"] < tol, out["pre"]
        u = self.usable
        for i, s in enumerate(self.sc):
            assert s["rtn"] == -4 and out["rtn"][i] == -4 and out["done"][i] == 1, (i, s["rtn"], out["rtn"][i])
            assert out["reg"][i] == s["reg"] == 24
            if u[i]:
                assert out["fp_failed"][i] == s["fp_failed"] and out["step"][i] == s["step"], (i, out["step"][i], s["step"])
        for n in ("X", "U", "S", "Y"):
            assert out["dev"][n][u].max() < tol, (n, out["dev"][n])
        assert out["cost"][u].max() < tol, out["cost"]

    def close(self):
        for q in self.oracle:
            q.close()


def scenarios():
    """(name, params, kind, K, y_inject, zero init_bez?): phase 0 as the caller runs it; the caller's phase-1 weights in
    infeasible mode (base 4 regulariser: 4^24 = 2.8e14 to beat) deep enough into the solve that barrier updates happen."""
    return [
        ("phase0", abi.phase0_params(), "corridor", 3, -1e-8, False),
        ("phase0_late", abi.phase0_params(fixed_iters=1), "corridor", 9, -1e-8, False),
        ("infeas_w100_k16", abi.phase1_params(infeas=1, fixed_iters=1, iter_max=40), "free", 16, -1e-18, True),
        ("infeas_w100_k17", abi.phase1_params(infeas=1, fixed_iters=1, iter_max=40), "free", 17, -1e-18, True),
    ]
