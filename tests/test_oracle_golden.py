"""The C oracle (oracle/direct_ref.c) against the golden vectors of the independent NumPy restatement.

PARITY UNPINNED w.r.t. the original Eigen code (it cannot be built here and ships no vectors,
SURVEY.md 8c): what is pinned is that two separately written restatements agree to ~1e-9 on every
per-iteration quantity, including all discrete decisions (step index, regulariser, exits)."""
import numpy as np
import pytest

from direct_amd import abi
from oracle import refapi
from tests import helpers


@pytest.mark.parametrize("name", helpers.CASES)
def test_oracle_matches_golden(name):
    g, batch = helpers.load_case(name)
    p0, p1 = helpers.case_params(name)
    r0, tr0 = refapi.solve_batch(p0, batch, trace_cap=p0.iter_max)
    helpers.check_result(r0, g, "p0_", 1e-9, T_tol=1e-8)
    r1, tr1 = refapi.solve_batch(p1, helpers.phase1_batch(g, batch), trace_cap=p1.iter_max)
    helpers.check_result(r1, g, "p1_", 1e-8, T_tol=1e-6)
    for tr, key, res in ((tr0, "p0_trace", r0), (tr1, "p1_trace", r1)):
        gt = g[key]
        for b in range(batch.batch):
            n = int(res.fwd_passes[b])
            a, e = tr[b, :n], gt[b, :n]
            assert np.array_equal(a[:, 6], e[:, 6]), "step indices differ"      # accepted line-search index
            assert np.array_equal(a[:, 5], e[:, 5]), "regulariser differs"
            assert np.array_equal(a[:, 9], e[:, 9]), "fp_failed differs"
            for col in (0, 1, 2, 4):                                            # cost, costq, logcost, mu
                assert np.abs(a[:, col] - e[:, col]).max() <= 1e-7 * np.abs(e[:, col]).max() + 1e-12


@pytest.mark.parametrize("name", helpers.EXIT_CASES)
def test_oracle_matches_exit_goldens(name):
    """Every way out of the outer loop and every failure branch inside it (DDP:295-412; tests/golden/make_exit_golden.py):
    the loop running out, rtn -3, rtn -4 in the first iteration, LLT failures with recovery, both line-init exits - the C
    oracle against the NumPy restatement's vectors, decision by decision: regulariser, step index, line-search failure,
    the number of backward sweeps of every retry loop and whether it gave up."""
    g, batch, p = helpers.load_exit_case(name)
    r, tr = refapi.solve_batch(p, batch, trace_cap=p.iter_max + 1)
    helpers.check_exit_result(r, g, 1e-9)
    gt = g["out_trace"]
    for b in range(batch.batch):
        n = int(r.fwd_passes[b])
        a, e = tr[b, :n], gt[b, :n]
        for col in (5, 6, 9, 10, 11):   # reg, step, fp_failed, n_sweeps, bp_failed
            assert np.array_equal(a[:, col], e[:, col]), (b, refapi.TRACE_COLS[col])
        for col in (0, 1, 4):           # cost, costq, mu (the log-cost is NaN where the start is infeasible in feasible mode)
            assert np.abs(a[:, col] - e[:, col]).max() <= 1e-7 * np.abs(e[:, col]).max() + 1e-12
    if name == "exit_llt_retry":
        assert (np.nan_to_num(gt[:, :, 10]) > 1).any(axis=1).all() and (np.nan_to_num(gt[:, :, 11]) == 0).all()
    if name == "exit_stuck_first":
        assert gt[1, 0, 10] == 45 and gt[1, 0, 11] == 1   # reg 1 .. 24, 21 sweeps at 24, given up


def test_oracle_matches_forced_stuck_golden():
    """The backward pass stuck mid-solve and the stale-gain forward pass that follows (tests/stuck_lib.py), as the NumPy
    restatement ran it (its stored ks, Ks, ky, Ky are Eigen-literal): the C oracle's iterate after the last trip."""
    g, batch, p = helpers.load_exit_case("exit_forced_stuck")

    class Impl:
        def __init__(self):
            self.st = [refapi.Stepper(p, batch, i) for i in range(batch.batch)]

        def iterate(self, n):
            for s in self.st:
                s.iterate(n)

        def get(self, f):
            return np.stack([s.get(f) for s in self.st])

        def set(self, f, a):
            for i, s in enumerate(self.st):
                s.set(f, a[i])

        def scalars(self):
            rows = [s.scalars() for s in self.st]
            return {n: np.array([r[n] for r in rows]) for n in abi.SCALAR_NAMES}
    helpers.run_forced_stuck(Impl(), g, 1e-9)


def test_plan_batch_equals_two_calls():
    g, batch = helpers.load_case("corridor_n8")
    p0, p1 = helpers.case_params("corridor_n8")
    r0, r1 = refapi.plan_batch(p0, p1, batch)
    helpers.check_result(r0, g, "p0_", 1e-9, T_tol=1e-8)
    helpers.check_result(r1, g, "p1_", 1e-8, T_tol=1e-6)


def test_openmp_batch_is_deterministic():
    g, batch = helpers.load_case("corridor_n20")
    p0, _ = helpers.case_params("corridor_n20")
    a, _ = refapi.solve_batch(p0, batch, n_threads=1)
    b, _ = refapi.solve_batch(p0, batch, n_threads=2)
    assert np.array_equal(a.bez, b.bez) and np.array_equal(a.cost, b.cost)


def test_fixed_iters_runs_exactly_iter_max():
    g, batch = helpers.load_case("free_n5")
    p = abi.phase1_params(iter_max=7, fixed_iters=1)
    r, _ = refapi.solve_batch(p, helpers.phase1_batch(g, batch))
    assert (r.fwd_passes == 7).all() and (r.iter_used == 7).all() and (r.rtn == 0).all()


def test_invalid_time_power_rejected():
    g, batch = helpers.load_case("free_n5")
    with pytest.raises(RuntimeError):
        refapi.solve_batch(abi.phase0_params(time_power=3), batch)
