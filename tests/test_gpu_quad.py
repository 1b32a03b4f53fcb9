"""BASELINE-label model on the device (12-state / 4-control quadrotor iLQR, include/direct_quad.h) against its CPU
checker oracle/quad_ref.c.  No reference counterpart exists; tolerances: fp64 per-pass gains 1e-9, whole solve
identical iteration counts and cost 1e-9; float storage: cost 1e-3 (SURVEY.md 8c's fp32 whole-solve tolerance)."""
import numpy as np
import pytest

from direct_amd import quad
from oracle import quadapi

pytestmark = pytest.mark.gpu


def test_per_pass_gains_and_iterates_fp64(built):
    p = quad.default_params()
    x0, xg = quad.label_problems(6, seed=1000)
    s = quad.QuadSolver(6, 40, np.float64)
    s.begin(p, x0, xg)
    ref = [quadapi.Stepper(p, 40, x0[b], xg[b]) for b in range(6)]
    g = s.get()
    for b in range(6):
        r = ref[b].get()
        assert abs(g["cost"][b] / r["cost"] - 1) < 1e-13 and np.abs(g["x"][b] - r["x"]).max() < 1e-12
    for it in range(4):
        s.iterate(1)
        g = s.get()
        for b in range(6):
            ref[b].iterate(1)
            r = ref[b].get()
            scale = np.abs(r["K"]).max()
            assert np.abs(g["K"][b] - r["K"]).max() < 1e-9 * scale and np.abs(g["kf"][b] - r["kf"]).max() < 1e-9 * np.abs(r["kf"]).max()
            assert g["step"][b] == r["step"] and g["reg"][b] == r["reg"] and g["fp_failed"][b] == r["fp_failed"]
            assert abs(g["cost"][b] / r["cost"] - 1) < 1e-10
            assert np.abs(g["x"][b] - r["x"]).max() < 1e-9 and np.abs(g["u"][b] - r["u"]).max() < 1e-9
    s.close()


def test_whole_solve_batch(built):
    p = quad.default_params()
    B, N = 512, 100
    x0, xg = quad.label_problems(B, seed=1000)
    s = quad.QuadSolver(B, N, np.float64)
    g = s.solve(p, x0, xg)
    idx = np.array([0, 17, 255, 511])
    r = quadapi.solve_batch(p, N, x0[idx], xg[idx])
    assert np.array_equal(g["iters"][idx], r["iters"])
    assert np.abs(g["cost"][idx] / r["cost"] - 1).max() < 1e-9
    assert np.abs(g["x"][idx] - r["x"]).max() < 1e-7 and np.abs(g["u"][idx] - r["u"]).max() < 1e-6
    # size-independent properties of the whole batch
    assert (g["iters"] > 2).all() and (g["iters"] <= p.iter_max).all()
    assert np.linalg.norm(g["x"][:, -1, :3] - xg[:, :3], axis=1).max() < 0.2      # reaches the goal
    step = g["x"][:, 1:, :3] - g["x"][:, :-1, :3] - p.dt * g["x"][:, :-1, 3:6]    # pdot = v, explicit Euler
    assert np.abs(step).max() < 1e-12
    assert np.array_equal(s.solve(p, x0, xg)["x"], g["x"])                        # deterministic
    sub = quad.QuadSolver(64, N, np.float64)
    assert np.array_equal(sub.solve(p, x0[128:192], xg[128:192])["x"], g["x"][128:192])   # sharding invariant
    sub.close()
    s.close()
    # float storage, double arithmetic
    f = quad.QuadSolver(B, N, np.float32)
    h = f.solve(p, x0, xg)
    assert np.abs(h["cost"][idx] / r["cost"] - 1).max() < 1e-3
    assert np.linalg.norm(h["x"][:, -1, :3] - xg[:, :3], axis=1).max() < 0.2
    pf = quad.default_params(iter_max=10, fixed_iters=1)
    hf = f.solve(pf, x0, xg)
    assert (hf["iters"] == 10).all() and f.last_kernel_ms() > 0
    f.close()
