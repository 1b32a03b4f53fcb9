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
