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
