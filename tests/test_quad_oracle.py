"""The label-model oracle (oracle/quad_ref.c) checked against things that do not depend on it: finite-difference
Jacobians of the Euler-discretised quadrotor dynamics, hover as a fixed point, monotone cost, and reaching the goal.
(The BASELINE-label model has no reference counterpart: include/direct_quad.h.)"""
import ctypes as C

import numpy as np

from direct_amd import quad
from oracle import quadapi


def params(**kw):
    # default_params lives in the product library; the struct is filled here by hand so that this file needs no build
    p = quad.Params(0.98, 9.81, (C.c_double * 3)(2.64e-3, 2.64e-3, 4.96e-3), 0.05, 1.0, 0.1, 1.0, 0.05, 0.05, 50.0,
                    1000.0, 500.0, 500.0, 100.0, 4.0, 1e-6, 50, 0)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def test_jacobians_match_finite_differences():
    p = params()
    rng = np.random.default_rng(0)
    for _ in range(5):
        x = rng.normal(0, 0.4, 12)
        u = np.array([9.0, 0.01, -0.02, 0.005]) + rng.normal(0, 0.01, 4)
        A, B = quadapi.jacobians(p, x, u)
        h = 1e-6
        for j in range(12):
            e = np.zeros(12); e[j] = h
            fd = (quadapi.step(p, x + e, u) - quadapi.step(p, x - e, u)) / (2 * h)
            assert np.abs(fd - A[:, j]).max() < 1e-8
        for j in range(4):
            e = np.zeros(4); e[j] = h
            fd = (quadapi.step(p, x, u + e) - quadapi.step(p, x, u - e)) / (2 * h)
            assert np.abs(fd - B[:, j]).max() < 1e-7


def test_hover_is_a_fixed_point_and_costs_nothing():
    p = params()
    x = np.zeros(12); x[:3] = (1.0, -2.0, 1.5)
    assert np.array_equal(quadapi.step(p, x, np.array([0.98 * 9.81, 0, 0, 0])), x)
    s = quadapi.Stepper(p, 20, x, x)
    assert s.get()["cost"] == 0.0
    s.close()


def test_ilqr_reaches_the_goal_with_monotone_cost():
    p = params()
    x0, xg = quad.label_problems(4, seed=1000)
    for b in range(4):
        s = quadapi.Stepper(p, 100, x0[b], xg[b])
        costs = [s.get()["cost"]]
        while not s.iterate(1):
            costs.append(s.get()["cost"])
        g = s.get()
        assert all(b2 <= a for a, b2 in zip(costs, costs[1:]))
        assert costs[-1] < 0.02 * costs[0]
        assert np.linalg.norm(g["x"][-1, :3] - xg[b, :3]) < 0.15 and np.abs(g["x"][-1, 3:6]).max() < 0.2
        assert g["iter"] < 50 and np.abs(g["x"][:, 6:8]).max() < 1.2   # tilts, never near the Euler singularity
        s.close()
