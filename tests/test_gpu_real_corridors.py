"""The DDP path on what the reference's own pipeline feeds it (tests/real_corridor_lib.py): 64 grid paths on a
200 x 200 x 40 voxel map -> corridorGeneration on the device -> the replay protocol of corridorRecCallBack
(teach_repeat_planner.cpp:316-320: the first n polytopes, n = 2 ..) -> ONE ragged batch of ~700 two-phase plans with
N <= 21 segments and 6 .. 60 planes per polytope, run by the kernels of the widest polytope with the row slots a knot
does not need skipped at run time (ddp_wave.h, slot_on); the row-slot CLASSES (direct_ddp.hip, classify_batch: a plan on
the kernels of its OWN widest polytope; opt-in) are forced on for the comparison.

Checked: both phases of 96 plans spread over the batch against the oracle (identical return codes and iteration
counts, cost 1e-6), both storage types; the class dispatch against the single-class dispatch bit for bit; containment
of the feasible results."""
import numpy as np
import pytest

from direct_amd import abi, solver
from oracle import refapi
from tests import real_corridor_lib, soak_lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def real_batch(built):
    return real_corridor_lib.real_corridor_batch(64)


def test_real_corridor_plans_against_the_oracle(real_batch, monkeypatch):
    batch, meta = real_batch
    assert batch.batch >= 600 and meta["corridor_generation"]["one_by_one_equals_lock_step"]
    assert meta["corridor_generation"]["corridors_with_a_polytope_above_P_LIMIT"] == 0   # DIRECT_P_LIMIT = 128: none is refused
    assert batch.p_max > 33       # some plan needs more than the four-slot kernels ...
    widest = batch.n_planes.max(axis=1)
    assert (widest <= 22).sum() > 20    # ... and many need no more than the three-slot ones
    p0, p1 = abi.phase0_params(), abi.phase1_params()
    idx = np.arange(0, batch.batch, max(1, batch.batch // 96))[:96]
    sb = batch.select(idx)
    # both phases against the oracle from IDENTICAL inputs: phase 1 of both starts from the oracle's phase-0 result
    # through the reference's own Bezier hand-off (teach_repeat_planner.cpp:911-921)
    r0 = refapi.solve_batch(p0, sb)[0]
    b1 = soak_lib.phase1_inputs(sb, r0)
    r1 = refapi.solve_batch(p1, b1)[0]
    s = solver.DdpSolver(batch.batch, int(batch.n_seg_max), int(batch.p_max), np.float64)
    g0, g1 = s.plan(p0, p1, batch)          # the whole ragged batch, fused two-phase plan
    assert s.sched_error() == 0
    d = s.sample(batch.n_seg, g1.bez, g1.T, 0.05, 8192, derivs=0, n_planes=batch.n_planes, planes=batch.planes)
    d1 = s.solve(p1, b1)                    # the sample's phase 1 from the oracle's hand-off
    s.close()
    assert (g0.rtn[idx] == r0.rtn).all() and (g0.iter_used[idx] == r0.iter_used).all()
    assert np.abs(g0.cost[idx] / r0.cost - 1).max() < 1e-8
    # identical return codes and iteration counts - except where the algorithm itself amplifies the last bit: a solve
    # that needs ~100 iterations may end one iteration apart (measured: 1 of 96, a 97-iteration solve), exactly what the
    # oracle does against ITSELF with its inputs moved by one ulp (computed here, on these problems)
    c1 = refapi.solve_batch(p1, soak_lib.perturb_ulp(b1, 7))[0]
    differs = (d1.rtn != r1.rtn) | (d1.iter_used != r1.iter_used)
    ctl = (c1.rtn != r1.rtn) | (c1.iter_used != r1.iter_used)
    assert differs.sum() <= max(2, ctl.sum() + 1), (np.flatnonzero(differs), np.flatnonzero(ctl))
    assert (r1.iter_used[differs] >= 40).all(), r1.iter_used[differs]       # never a short, well-conditioned solve
    ok = (r1.rtn == 1) & ~differs             # converged (DDP:374): cost to 1e-6
    assert ok.sum() >= 40
    assert np.abs(d1.cost[ok] / r1.cost[ok] - 1).max() < 1e-6
    lim = (r1.rtn == 0) & ~differs & ~ctl     # stopped by the iteration limit: wherever iteration 100 leaves them - bounded by the control
    if lim.any():
        cdev = np.abs(c1.cost[lim] / r1.cost[lim] - 1).max()
        assert np.abs(d1.cost[lim] / r1.cost[lim] - 1).max() <= max(1e-6, 30 * cdev), cdev
    # the fused plan (device's own phase 0, monomial hand-off) ends like the oracle's plan wherever the oracle converges
    conv = r1.rtn == 1
    assert (g1.rtn[idx][conv] == 1).mean() >= 0.95 and np.median(np.abs(g1.cost[idx][conv] / r1.cost[conv] - 1)) < 1e-6
    feas = (g1.rtn >= 0) & (g1.infeas_out == 0)     # left the infeasible mode: every constraint row holds
    assert feas.sum() > batch.batch // 2
    assert d["cmax"][feas].max() < 1e-3             # Bezier control points are inside; the samples with them (2e-4 shift, DDP:1281)
    # the class dispatch is scheduling only: every plan on the kernels of its own widest polytope (side by side on forked
    # streams) gives the same bits as the one class this batch size gets by default
    monkeypatch.setenv("DIRECT_DDP_CLASSES", "1")
    s1 = solver.DdpSolver(batch.batch, int(batch.n_seg_max), int(batch.p_max), np.float64)
    h0, h1 = s1.plan(p0, p1, batch)
    s1.close()
    for f in ("rtn", "iter_used", "fwd_passes", "cost", "T", "bez"):
        assert np.array_equal(getattr(g0, f), getattr(h0, f)) and np.array_equal(getattr(g1, f), getattr(h1, f)), f


def test_real_corridor_plans_float_storage(real_batch):
    batch, _ = real_batch
    p0, p1 = abi.phase0_params(), abi.phase1_params()
    idx = np.arange(3, batch.batch, max(1, batch.batch // 64))[:64]
    sb = batch.select(idx).astype(np.float32).astype(np.float64)
    r0, r1 = refapi.plan_batch(p0, p1, sb)
    s = solver.DdpSolver(batch.batch, int(batch.n_seg_max), int(batch.p_max), np.float32)
    g0, g1 = s.plan(p0, p1, batch.astype(np.float32))
    assert s.sched_error() == 0
    s.close()
    assert ((g0.rtn[idx] >= 0) == (r0.rtn >= 0)).all()
    assert (g1.rtn[idx] == r1.rtn).mean() >= 0.95
    both = (g1.rtn[idx] == r1.rtn) & (r1.rtn == 1)
    assert both.sum() >= 32
    dev = np.abs(g1.cost[idx][both] / r1.cost[both] - 1)
    assert np.median(dev) < 1e-5 and (dev < 1e-3).mean() >= 0.95     # fp32 whole-solve 1e-3
