"""Shared helpers of the test suite: golden-fixture loading and comparison utilities."""
import os

import numpy as np

from direct_amd import abi

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ("free_n5", "corridor_n8", "corridor_n20", "config1_n50", "corridor_n8_minvo", "free_n6_tp1")
SMALL_CASES = ("free_n5", "corridor_n8", "corridor_n8_minvo", "free_n6_tp1")


def case_params(name):
    kw = {}
    if name.endswith("_minvo"):
        kw["minvo"] = 1
    if name.endswith("_tp1"):
        kw["time_power"] = 1
    return abi.phase0_params(**kw), abi.phase1_params(**kw)


def load_case(name):
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    batch = abi.HostBatch(g["n_seg"], g["x0"], g["xd"], g["T0"], g["n_planes"], g["planes"], seeds=g["seeds"])
    return g, batch


def phase1_batch(g, batch):
    """Inputs of the second polyCurveGeneration call (teach_repeat_planner.cpp:911-921)."""
    T1 = np.where((g["p0_rtn"] == 2)[:, None], g["p0_T"], batch.T0)
    return batch.with_init(g["p0_bez"], T0=T1, infeas_in=g["p0_infeas_out"].astype(np.uint8))


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def check_result(res, g, prefix, tol, exact_iters=True, T_tol=None, bez_tol=None):
    """Compare a HostResult with the golden outputs of one phase."""
    T_tol = tol if T_tol is None else T_tol
    bez_tol = T_tol if bez_tol is None else bez_tol
    assert (res.rtn == g[prefix + "rtn"].astype(int)).all(), (res.rtn, g[prefix + "rtn"])
    if exact_iters:
        assert (res.iter_used == g[prefix + "iter_used"].astype(int)).all(), (res.iter_used, g[prefix + "iter_used"])
        assert (res.fwd_passes == g[prefix + "fwd_passes"].astype(int)).all()
    assert np.abs(res.cost / g[prefix + "cost"] - 1).max() < tol, np.abs(res.cost / g[prefix + "cost"] - 1).max()
    assert rel(res.T, g[prefix + "T"]) < T_tol, rel(res.T, g[prefix + "T"])
    assert rel(res.bez, g[prefix + "bez"]) < bez_tol, rel(res.bez, g[prefix + "bez"])
    assert rel(res.poly, g[prefix + "poly"]) < bez_tol
    assert rel(res.jerk_cost, g[prefix + "jerk_cost"]) < 10 * bez_tol
    assert (res.infeas_out == g[prefix + "infeas_out"].astype(int)).all()


def with_extra_planes(batch, p_max, seed=0):
    """Pads every polytope of a HostBatch with random half-spaces through points ~1 m outside its seed
    segment (valid but mostly inactive), up to p_max planes per polytope in total with varying counts:
    exercises the RPL = 3 / 4 kernels (nc = 6P + 55 > 128 / 192)."""
    rng = np.random.default_rng(seed)
    B, nm, p0 = batch.planes.shape[:3]
    planes = np.zeros((B, nm, p_max, 4))
    planes[:, :, :p0] = batch.planes
    n_planes = batch.n_planes.copy()
    for b in range(B):
        for k in range(int(batch.n_seg[b])):
            a = batch.x0[b, :3] if k == 0 else batch.seeds[b, k]
            c = batch.xd[b, :3] if k == batch.n_seg[b] - 1 else batch.seeds[b, k + 1]
            target = int(rng.integers(max(p0, p_max - 5), p_max + 1))
            j = int(n_planes[b, k])
            while j < target:
                n = rng.normal(size=3)
                n /= np.linalg.norm(n)
                off = max(float(n @ a), float(n @ c)) + float(rng.uniform(0.8, 3.0))   # both seeds strictly inside
                planes[b, k, j] = np.r_[n, -off]
                j += 1
            n_planes[b, k] = j
    return abi.HostBatch(batch.n_seg, batch.x0, batch.xd, batch.T0, n_planes, planes, seeds=batch.seeds, dtype=batch.dtype)


# ---- fixtures of tests/golden/make_exit_golden.py: one per exit / failure branch of the outer loop (DDP:295-412) -------
EXIT_CASES = ("exit_iter_max", "exit_neg_time", "exit_stuck_first", "exit_llt_retry", "exit_line_ok", "exit_line_no_update")
_PARAM_FIELDS = ("max_vel", "max_acc", "w_snap", "w_terminal", "w_time", "iter_max", "time_power", "zero_init", "line_init",
                 "minvo", "infeas", "fixed_iters", "exact_dt")
_INT_PARAMS = ("iter_max", "time_power", "zero_init", "line_init", "minvo", "infeas", "fixed_iters", "exact_dt")


def load_exit_case(name):
    """-> (npz, HostBatch, Params) of one exit fixture"""
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    batch = abi.HostBatch(g["n_seg"], g["x0"], g["xd"], g["T0"], g["n_planes"], g["planes"], seeds=g["seeds"],
                          init_bez=g["init_bez"] if "init_bez" in g.files else None,
                          infeas_in=g["infeas_in"] if "infeas_in" in g.files else None)
    kw = {f: (int(g["param_" + f]) if f in _INT_PARAMS else float(g["param_" + f])) for f in _PARAM_FIELDS}
    return g, batch, abi.Params(**kw)


def check_exit_result(res, g, tol, bez_tol=None, discrete=True):
    """a HostResult against the getter outputs of an exit fixture (the NumPy restatement's)"""
    bez_tol = 100 * tol if bez_tol is None else bez_tol
    if discrete:
        for f in ("rtn", "iter_used", "fwd_passes", "infeas_out", "line_failed_out"):
            assert (np.asarray(getattr(res, f)).astype(int) == g["out_" + f].astype(int)).all(), (f, getattr(res, f), g["out_" + f])
    assert np.abs(res.cost / g["out_cost"] - 1).max() < tol, np.abs(res.cost / g["out_cost"] - 1).max()
    assert rel(res.T, g["out_T"]) < bez_tol and rel(res.bez, g["out_bez"]) < bez_tol and rel(res.poly, g["out_poly"]) < bez_tol
    assert rel(res.jerk_cost, g["out_jerk_cost"]) < 10 * bez_tol


def run_forced_stuck(impl, g, tol):
    """the stepwise scenario of exit_forced_stuck.npz on an implementation that has begun (iterate / get / set / scalars):
    K iterations, one dual entry per problem made negative, the stuck trip - against the NumPy restatement's iterate"""
    K, knot, rows, y_inject = int(g["K"]), int(g["knot"]), g["rows"], float(g["y_inject"])
    B, N = g["n_seg"].shape[0], int(g["n_seg"][0])
    impl.iterate(K)
    Y = impl.get(abi.FIELD_Y)
    for i, r in enumerate(rows):
        Y[i, knot, r] = y_inject
    impl.set(abi.FIELD_Y, Y)
    impl.iterate(1)
    sc = impl.scalars()
    u = g["sc_usable"] == 1
    assert (sc["rtn"].astype(int) == -4).all() and (sc["reg"].astype(int) == 24).all()
    assert (sc["step"].astype(int)[u] == g["sc_step"].astype(int)[u]).all() and (sc["fp_failed"].astype(int)[u] == g["sc_fp_failed"].astype(int)[u]).all()
    assert np.abs(np.asarray(sc["cost"], np.float64)[u] / g["sc_cost"][u] - 1).max() < tol
    for f, n in ((abi.FIELD_X, "X"), (abi.FIELD_U, "U"), (abi.FIELD_S, "S"), (abi.FIELD_Y, "Y")):
        got = np.asarray(impl.get(f), np.float64)
        want = g["post_" + n]
        assert got.shape == want.shape, (n, got.shape, want.shape)
        for i in np.where(u)[0]:
            assert rel(got[i], want[i]) < tol, (n, i, rel(got[i], want[i]))
