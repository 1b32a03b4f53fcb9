"""Checks of the oracle that depend on NO restatement: calculus identities of the model
(SURVEY.md 8c item 3).  References: ddp_optimizer.cpp:836-1015 (F, G, R tables), 1132-1285 (c),
1455-1604 (cx, cu), 782-812 (conversions)."""
import numpy as np
import pytest

from direct_amd import abi, problems
from oracle import ddp_numpy, refapi

rng = np.random.default_rng(0)


def rand_xu(T=None):
    x = rng.normal(size=9)
    u = rng.normal(size=10) * 0.3
    u[9] = rng.uniform(0.5, 2.5) if T is None else T
    return x, u


def poly_coeffs(x, u):
    C = np.zeros((6, 3))
    C[0], C[1], C[2] = x[0:3], x[3:6], 0.5 * x[6:9]
    C[3], C[4], C[5] = u[0:3], u[3:6], u[6:9]
    return C


def test_dynamics_is_polynomial_endpoint():
    """x+ = [p(T), p'(T), p''(T)] of the quintic with c0=p, c1=v, c2=a/2, c3..c5=u (DDP:862-871)."""
    for _ in range(5):
        x, u = rand_xu()
        T, C = u[9], poly_coeffs(x, u)
        p = sum(C[i] * T ** i for i in range(6))
        v = sum(i * C[i] * T ** (i - 1) for i in range(1, 6))
        a = sum(i * (i - 1) * C[i] * T ** (i - 2) for i in range(2, 6))
        assert np.allclose(refapi.eval_nextx(x, u), np.concatenate([p, v, a]), rtol=1e-12, atol=1e-12)


def test_running_cost_is_jerk_integral():
    """u'R(T)u = int_0^T |jerk|^2 dt (DDP:991-999; the weight is called w_snap, quirk Q15)."""
    p = abi.phase1_params(w_snap=1.0, w_time=0.0)
    for _ in range(5):
        x, u = rand_xu()
        T, C = u[9], poly_coeffs(x, u)
        ts = np.linspace(0, T, 20001)
        jerk = sum(i * (i - 1) * (i - 2) * C[i][None, :] * ts[:, None] ** (i - 3) for i in range(3, 6))
        integral = np.trapezoid((jerk ** 2).sum(1), ts)
        assert abs(2 * refapi.eval_q(p, u) - integral) < 1e-6 * max(1.0, integral)


def test_bezier_constraints_bound_the_curve():
    """Bezier mode: the position/velocity/acceleration rows are control points of p, p', p'' on [0,T]
    (convex-hull property), so max over samples <= max over control rows (DDP:79-95, 1181-1276)."""
    p = abi.phase1_params()
    planes = np.array([[1.0, 0, 0, -0.0], [0, 1.0, 0, 0], [0, 0, 1.0, 0], [-1.0, 0, 0, 0], [0, -1.0, 0, 0], [0, 0, -1.0, 0]])
    for _ in range(5):
        x, u = rand_xu()
        T, C = u[9], poly_coeffs(x, u)
        c = refapi.eval_c(p, x, u, planes) + 2e-4
        ts = np.linspace(0, T, 501)
        pos = sum(C[i][None, :] * ts[:, None] ** i for i in range(6))
        vel = sum(i * C[i][None, :] * ts[:, None] ** (i - 1) for i in range(1, 6))
        acc = sum(i * (i - 1) * C[i][None, :] * ts[:, None] ** (i - 2) for i in range(2, 6))
        cpos = c[:36].reshape(6, 6)                     # [ctrl point j][plane k] = n_k . B_j + d_k
        for k in range(6):
            assert (pos @ planes[k, :3] + planes[k, 3]).max() <= cpos[:, k].max() + 1e-9
        assert vel.max() - p.max_vel <= c[36:51].max() + 1e-9
        assert acc.max() - p.max_acc <= c[66:78].max() + 1e-9
        assert abs(c[-1] - (0.3 - T)) < 1e-12
        # endpoints interpolate: first / last position control point are p(0), p(T)
        assert np.allclose(cpos[0], planes[:, :3] @ pos[0] + planes[:, 3])
        assert np.allclose(cpos[5], planes[:, :3] @ pos[-1] + planes[:, 3])


@pytest.mark.parametrize("minvo", [0, 1])
def test_constraint_jacobians_vs_finite_differences(minvo):
    """cx and the u-columns of cu match central differences of c in both bases.  The dc/dT column is
    exact only with the MINVO basis: in Bezier mode it still uses the MINVO tables (quirk Q1,
    DDP:1543-1561) -- asserted here as a DIFFERENCE so the quirk cannot be 'fixed' silently."""
    batch = problems.make_batch("corridor", 1, 4, seed=5)
    params = abi.phase1_params(minvo=minvo)
    st = refapi.Stepper(abi.phase0_params(minvo=minvo), batch, 0)
    st.computeall()
    X, U = st.get(abi.FIELD_X), st.get(abi.FIELD_U)
    cx, cu = st.get(100), st.get(101)
    k = 1
    P = int(batch.n_planes[0, k])
    nc = 6 * P + 55
    planes = batch.planes[0, k, :P]
    x, u = X[k].copy(), U[k].copy()
    x += rng.normal(size=9) * 0.1
    u[:9] += rng.normal(size=9) * 0.1
    st.set(abi.FIELD_X, np.vstack([X[:k], x[None], X[k + 1:]]))
    st.set(abi.FIELD_U, np.vstack([U[:k], u[None], U[k + 1:]]))
    st.computeall()
    cx, cu = st.get(100)[k, :nc], st.get(101)[k, :nc]
    h = 1e-6
    for a in range(9):
        e = np.zeros(9); e[a] = h
        fd = (refapi.eval_c(params, x + e, u, planes) - refapi.eval_c(params, x - e, u, planes)) / (2 * h)
        assert np.abs(fd - cx[:, a]).max() < 1e-6
    for a in range(9):
        e = np.zeros(10); e[a] = h
        fd = (refapi.eval_c(params, x, u + e, planes) - refapi.eval_c(params, x, u - e, planes)) / (2 * h)
        assert np.abs(fd - cu[:, a]).max() < 1e-6
    e = np.zeros(10); e[9] = h
    fdT = (refapi.eval_c(params, x, u + e, planes) - refapi.eval_c(params, x, u - e, planes)) / (2 * h)
    if minvo:
        assert np.abs(fdT - cu[:, 9]).max() < 2e-6 * max(1.0, np.abs(fdT).max())
    else:
        assert np.abs(fdT - cu[:, 9]).max() > 1e-3, "Bezier-mode dc/dT must keep the MINVO tables (quirk Q1)"
        assert abs(cu[-1, 9] + 1.0) < 1e-15
    # the non-parity flag gives the exact derivative in Bezier mode
    if not minvo:
        st2 = refapi.Stepper(abi.phase0_params(exact_dt=1), batch, 0)
        st2.set(abi.FIELD_X, st.get(abi.FIELD_X)); st2.set(abi.FIELD_U, st.get(abi.FIELD_U))
        st2.computeall()
        assert np.abs(fdT - st2.get(101)[k, :nc, 9]).max() < 2e-6 * max(1.0, np.abs(fdT).max())


def test_dynamics_and_cost_jacobians_vs_finite_differences():
    batch = problems.make_batch("free", 1, 3, seed=2)
    st = refapi.Stepper(abi.phase1_params(), batch.with_init(np.zeros((1, 3, 18))), 0)
    X, U = st.get(abi.FIELD_X), st.get(abi.FIELD_U)
    U[:, :9] = rng.normal(size=(3, 9)) * 0.2
    X[1:] += rng.normal(size=(3, 9)) * 0.2
    st.set(abi.FIELD_X, X); st.set(abi.FIELD_U, U)
    st.computeall()
    fx, fu, qu, quu = st.get(102), st.get(103), st.get(104), st.get(105)
    p = abi.phase1_params()
    h = 1e-6
    k = 1
    for a in range(9):
        e = np.zeros(9); e[a] = h
        fd = (refapi.eval_nextx(X[k] + e, U[k]) - refapi.eval_nextx(X[k] - e, U[k])) / (2 * h)
        assert np.abs(fd - fx[k][:, a]).max() < 1e-7
    for a in range(10):
        e = np.zeros(10); e[a] = h
        fd = (refapi.eval_nextx(X[k], U[k] + e) - refapi.eval_nextx(X[k], U[k] - e)) / (2 * h)
        assert np.abs(fd - fu[k][:, a]).max() < 1e-6
        fdq = (refapi.eval_q(p, U[k] + e) - refapi.eval_q(p, U[k] - e)) / (2 * h)
        assert abs(fdq - qu[k][a]) < 1e-5 * max(1.0, abs(qu[k][a]))
    assert np.allclose(quu[k], quu[k].T)


def test_bezier_poly_round_trip():
    for _ in range(5):
        il = rng.normal(size=18)
        T = rng.uniform(0.4, 3.0)
        assert np.allclose(refapi.poly2bez(refapi.bez2poly(il, T), T), il, rtol=1e-10, atol=1e-10)
    # control points of the time-scaled Bezier: first control point * T is p(0)
    il = rng.normal(size=18)
    poly = refapi.bez2poly(il, 1.7)
    assert np.allclose(poly[:3], 1.7 * il[:3])


def test_time_allocation_matches_product_host_code():
    batch = problems.make_batch("free", 4, 7, seed=3)
    T_ref = refapi.time_allocation(batch.n_seg, batch.x0[:, :3], batch.xd[:, :3], batch.seeds)
    T_np = problems.time_allocation(batch.n_seg, batch.x0[:, :3], batch.xd[:, :3], batch.seeds)
    assert np.allclose(T_ref, T_np, rtol=1e-14)
    assert np.allclose(T_ref, batch.T0, rtol=1e-14)
    # closed forms: long segment = trapezoid, short = triangle
    assert abs(refapi.time_allocation([1], [[0, 0, 0]], [[10.0, 0, 0]], np.zeros((1, 1, 3)))[0, 0] - (1 + 4 + 1)) < 1e-12
    assert abs(refapi.time_allocation([1], [[0, 0, 0]], [[0.5, 0, 0]], np.zeros((1, 1, 3)))[0, 0] - 1.0) < 1e-12


def test_kkt_residual_small_at_convergence():
    """At the converged phase-1 iterate the barrier KKT residual opterr is far below its start."""
    from tests import helpers
    g, batch = helpers.load_case("corridor_n8")
    tr = g["p1_trace"]
    for b in range(batch.batch):
        n = int(g["p1_fwd_passes"][b])
        assert tr[b, n - 1, 7] < 50.0 < tr[b, 0, 7] * 1e3
        assert (np.diff(tr[b, :n, 4]) <= 1e-15).all(), "mu must be non-increasing"
