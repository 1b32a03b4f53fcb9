"""The kernel SOURCE (direct_amd/csrc/ddp_wave.h) compiled by the test-only lane-loop emulator
(tests/emu/emu.cpp: every 64-lane phase becomes a loop) against the oracle and the golden vectors.
This checks the restructured mathematics of the HIP kernels -- Kronecker-form constraint rows,
3x3 accumulators, column-per-lane LLT -- on a machine without a GPU.  It is not a product path."""
import numpy as np
import pytest

from direct_amd import abi, problems
from oracle import refapi
from tests import helpers, stuck_lib
from tests.emu import emuapi


@pytest.mark.parametrize("name", helpers.CASES)
def test_emulated_kernels_match_golden_fp64(name):
    g, batch = helpers.load_case(name)
    p0, p1 = helpers.case_params(name)
    e0 = emuapi.solve_batch(p0, batch)
    helpers.check_result(e0, g, "p0_", 1e-9, T_tol=1e-8)
    e1 = emuapi.solve_batch(p1, helpers.phase1_batch(g, batch))
    helpers.check_result(e1, g, "p1_", 1e-8, T_tol=1e-6)


@pytest.mark.parametrize("kind,N", [("free", 5), ("corridor", 8)])
def test_emulated_passes_match_oracle(kind, N):
    """One backward sweep and one forward pass from identical state: gains and iterates."""
    batch = problems.make_batch(kind, 2, N, seed=21)
    p0 = abi.phase0_params()
    e = emuapi.EmuSolver(p0, batch)
    r = [refapi.Stepper(p0, batch, i) for i in range(2)]
    e.backward()
    for q in r:
        q.backward()
    for f in (abi.FIELD_KU, abi.FIELD_KUU, abi.FIELD_KS, abi.FIELD_KY):
        for i in range(2):
            assert helpers.rel(e.get(f)[i], r[i].get(f)) < 1e-10
    e.forward()
    for q in r:
        q.forward()
    for f in (abi.FIELD_X, abi.FIELD_U, abi.FIELD_S, abi.FIELD_Y, abi.FIELD_C):
        for i in range(2):
            assert helpers.rel(e.get(f)[i], r[i].get(f)) < 1e-10
    sc = e.scalars()
    for i in range(2):
        rs = r[i].scalars()
        assert sc["step"][i] == rs["step"] and sc["fp_failed"][i] == rs["fp_failed"]
        assert abs(sc["logcost"][i] / rs["logcost"] - 1) < 1e-12


def test_emulated_float_storage_mode():
    """DIRECT_F32 = float storage in HBM with double arithmetic (DESIGN.md "Precision"): same exits as
    fp64; the objective agrees to the level the float-rounded inputs allow."""
    g, batch = helpers.load_case("corridor_n8")
    p0, p1 = helpers.case_params("corridor_n8")
    b1 = helpers.phase1_batch(g, batch)
    b1 = b1.with_init(None, T0=b1.T0, infeas_in=b1.infeas_in, init_poly=g["p0_poly"])
    e1 = emuapi.solve_batch(p1, b1, np.float32, compute64=True)
    assert (e1.rtn == g["p1_rtn"].astype(int)).all()
    assert np.abs(e1.cost / g["p1_cost"] - 1).max() < 1e-3
    assert helpers.rel(e1.T, g["p1_T"]) < 1e-2


def test_monomial_warm_start_equals_bezier_warm_start():
    """init_poly (C-ABI extension) and init_bez describe the same warm start in double."""
    g, batch = helpers.load_case("corridor_n8")
    _, p1 = helpers.case_params("corridor_n8")
    b1 = helpers.phase1_batch(g, batch)
    b1p = b1.with_init(None, T0=b1.T0, infeas_in=b1.infeas_in, init_poly=g["p0_poly"])
    ra, _ = refapi.solve_batch(p1, b1)
    rb, _ = refapi.solve_batch(p1, b1p)
    assert (ra.iter_used == rb.iter_used).all() and np.abs(ra.cost / rb.cost - 1).max() < 1e-9
    e = emuapi.solve_batch(p1, b1p)
    assert (e.iter_used == ra.iter_used).all() and np.abs(e.cost / ra.cost - 1).max() < 1e-8


def test_ragged_batch_and_mixed_plane_counts():
    """n_seg differs per problem (ragged), planes per polytope differ per knot."""
    a = problems.make_batch("corridor", 2, 9, seed=31)
    n_seg = np.array([9, 5], np.int32)
    xd = a.xd.copy()
    xd[1, :3] = a.seeds[1, 5]          # goal of the short problem = its 6th seed
    T0 = problems.time_allocation(n_seg, a.x0[:, :3], xd[:, :3], a.seeds)
    batch = abi.HostBatch(n_seg, a.x0, xd, T0, a.n_planes, a.planes, seeds=a.seeds)
    p0 = abi.phase0_params()
    r, _ = refapi.solve_batch(p0, batch)
    e = emuapi.solve_batch(p0, batch)
    assert (e.rtn == r.rtn).all() and (e.iter_used == r.iter_used).all()
    assert np.abs(e.cost / r.cost - 1).max() < 1e-9
    assert helpers.rel(e.T[1, :5], r.T[1, :5]) < 1e-8 and (e.T[1, 5:] == 0).all()


def test_fixed_iteration_mode_and_resume():
    """fixed_iters disables the early exits; iterate(a) + iterate(b) == iterate(a+b)."""
    g, batch = helpers.load_case("free_n5")
    b1 = helpers.phase1_batch(g, batch)
    p = abi.phase1_params(iter_max=6, fixed_iters=1)
    one = emuapi.EmuSolver(p, b1)
    one.iterate(6)
    two = emuapi.EmuSolver(p, b1)
    two.iterate(2)
    two.iterate(4)
    ra, rb = one.finish(), two.finish()
    assert (ra.fwd_passes == 6).all() and (rb.fwd_passes == 6).all()
    assert np.array_equal(ra.bez, rb.bez) and np.array_equal(ra.cost, rb.cost)
    r, _ = refapi.solve_batch(p, b1)
    assert np.abs(ra.cost / r.cost - 1).max() < 1e-9


@pytest.mark.parametrize("kind", ["free", "corridor"])
def test_line_initialisation_matches_oracle(kind):
    """line_init_flag = true (DDP:194-248, 255-269, 283-286, 398-409): straight quintics between the
    polytope seeds with duration doubling, reg = 10, the line-init exit rules."""
    batch = problems.make_batch(kind, 3, 6, seed=33)
    # make some first durations too short so that the doubling loop has to act
    batch.T0[1] *= 0.35
    batch.T0[2] *= 0.1
    p = abi.phase1_params(line_init=1, infeas=1, iter_max=40)
    e = emuapi.EmuSolver(p, batch)
    r = [refapi.Stepper(p, batch, i) for i in range(3)]
    for i in range(3):
        assert helpers.rel(e.get(abi.FIELD_U)[i], r[i].get(abi.FIELD_U)) < 1e-12
        assert helpers.rel(e.get(abi.FIELD_X)[i], r[i].get(abi.FIELD_X)) < 1e-12
        se, sr = e.scalars(), r[i].scalars()
        assert int(se["infeas"][i]) == int(sr["infeas"]) and int(se["reg"][i]) == int(sr["reg"]) == 10
        assert abs(se["cost"][i] / sr["cost"] - 1) < 1e-11 and abs(se["mu"][i] / sr["mu"] - 1) < 1e-11
    assert not np.allclose(e.get(abi.FIELD_U)[2][:, 9], batch.T0[2])  # durations were doubled
    g = emuapi.solve_batch(p, batch)
    o, _ = refapi.solve_batch(p, batch)
    assert (g.rtn == o.rtn).all() and (g.iter_used == o.iter_used).all()
    assert (g.line_failed_out == o.line_failed_out).all() and (g.infeas_out == o.infeas_out).all()
    assert np.abs(g.cost / o.cost - 1).max() < 1e-8
    assert helpers.rel(g.T, o.T) < 1e-8


@pytest.mark.parametrize("p_max", [20, 32, 44, 54, 76, 90, 110, 128])
def test_many_planes_per_polytope(p_max):
    """P up to 20 / 32 planes: nc = 175 / 247 rows per knot, the RPL = 3 / 4 instantiations."""
    batch = helpers.with_extra_planes(problems.make_batch("corridor", 2, 6, seed=41), p_max, seed=p_max)
    assert batch.n_planes.max() >= p_max - 2
    p0 = abi.phase0_params()
    e = emuapi.EmuSolver(p0, batch)
    r = [refapi.Stepper(p0, batch, i) for i in range(2)]
    e.backward()
    e.forward()
    for i, q in enumerate(r):
        q.backward()
        q.forward()
        for f in (abi.FIELD_KU, abi.FIELD_KUU, abi.FIELD_KS, abi.FIELD_KY, abi.FIELD_X, abi.FIELD_S, abi.FIELD_Y):
            assert helpers.rel(e.get(f)[i], q.get(f)) < 1e-9, f
    g = emuapi.solve_batch(p0, batch)
    o, _ = refapi.solve_batch(p0, batch)
    assert (g.rtn == o.rtn).all() and (g.iter_used == o.iter_used).all()
    assert np.abs(g.cost / o.cost - 1).max() < 1e-8


def test_emulated_begin_rejects_bad_sizes_row_by_row():
    """begin() validates n_seg / n_planes itself (device-resident inputs never pass the host's checks): bad
    rows finish at once with DIRECT_RTN_INVALID and zero outputs, good rows are solved as if alone."""
    batch = problems.make_batch("corridor", 6, 5, seed=9)
    p0 = abi.phase0_params()
    want = emuapi.solve_batch(p0, batch)
    bad = abi.HostBatch(batch.n_seg.copy(), batch.x0, batch.xd, batch.T0, batch.n_planes.copy(), batch.planes, seeds=batch.seeds)
    bad.n_seg[1] = 0
    bad.n_seg[2] = 9
    bad.n_planes[4, 3] = batch.p_max + 1
    got = emuapi.solve_batch(p0, bad)
    rows = np.array([1, 2, 4])
    assert (got.rtn[rows] == abi.RTN_INVALID).all() and not got.bez[rows].any() and not got.T[rows].any()
    for i in (0, 3, 5):
        assert got.rtn[i] == want.rtn[i] and np.array_equal(got.bez[i], want.bez[i])


def test_backward_pass_stuck_exit_matches_oracle():
    """rtn = -4 (DDP:392-396) in the FIRST iteration of feasible-mode solves from an infeasible start: the backward pass
    fails 21 times at the largest regulariser and the reference still runs its forward pass - with the zero gains of
    DDP:154-159 for the knots no sweep reached (Wave::stale_fwd_pass, the `nogain` rows)."""
    b = problems.make_batch("corridor", 32, 8, seed=1)
    bb = b.with_init(np.zeros((32, 8, 18)), T0=b.T0 * 3.0, infeas_in=np.zeros(32, np.uint8))
    p = abi.phase1_params(iter_max=60)
    r, _ = refapi.solve_batch(p, bb)
    e = emuapi.solve_batch(p, bb)
    assert (r.rtn == -4).sum() >= 1
    assert (e.rtn == r.rtn).all() and (e.iter_used == r.iter_used).all() and (e.fwd_passes == r.fwd_passes).all()
    ok = np.isfinite(r.cost)
    assert np.abs(e.cost[ok] / r.cost[ok] - 1).max() < 1e-8


@pytest.mark.parametrize("name", helpers.EXIT_CASES)
def test_emulated_kernels_match_exit_goldens(name):
    """every way out of the outer loop (DDP:295-412) against the NumPy restatement's vectors (make_exit_golden.py)"""
    g, batch, p = helpers.load_exit_case(name)
    helpers.check_exit_result(emuapi.solve_batch(p, batch), g, 1e-8)


def test_emulated_kernels_match_forced_stuck_golden():
    g, batch, p = helpers.load_exit_case("exit_forced_stuck")
    e = emuapi.EmuSolver(p, batch)
    helpers.run_forced_stuck(e, g, 1e-9)
    e.close()


@pytest.mark.parametrize("scenario", range(len(stuck_lib.scenarios())))
def test_forward_pass_after_a_stuck_backward_pass_uses_the_stored_gains(scenario):
    """The reference's forwardpass() after DDP:297-310 gave up runs with the gains its members hold: knots the retry
    sequence never reached keep those of the last COMPLETED sweep - another iterate, another barrier parameter
    (DDP:568-572, 611-614, 630-631 -> 680-703).  Forced a few iterations into well-conditioned solves (tests/stuck_lib.py),
    both storage types: every decision of the last trip and the iterate it leaves."""
    name, p, kind, K, y_inject, zero_bez = stuck_lib.scenarios()[scenario]
    batch = problems.make_batch(kind, 8, 10, seed=77)
    if zero_bez:
        batch = batch.with_init(np.zeros((8, 10, 18)))
    sc = stuck_lib.Scenario(p, batch, K, y_inject)
    if name == "phase0":
        assert sc.accepted().sum() >= 3   # the stale pass really moves the iterate
    if name.startswith("infeas_w100"):
        assert sc.barrier_moved().sum() >= 1   # the stale gains were formed with another barrier parameter
    for dtype, c64, tol in ((np.float64, False, 1e-9), (np.float32, True, 1e-3)):
        e = emuapi.EmuSolver(p, batch, dtype, c64)
        sc.check(sc.run(e), tol)
        e.close()
    sc.close()


@pytest.mark.parametrize("params", [abi.phase0_params(), abi.phase1_params()], ids=["infeasible", "feasible"])
def test_stored_gain_forward_pass_equals_the_regular_one_when_every_gain_is_current(params):
    """With a completed backward sweep every gain belongs to the current iterate, and the stored-gain form of the forward
    pass (Wave::stale_fwd_pass: s+, y+ from the iterate the gains were formed from) must reproduce the regular pass
    (run_round: the same iterate by construction) - both modes of DDP:678-706."""
    g, batch = helpers.load_case("corridor_n8")
    b = batch if params.zero_init else helpers.phase1_batch(g, batch)
    for warm in (0, 2):
        a, c = emuapi.EmuSolver(params, b), emuapi.EmuSolver(params, b)
        for s in (a, c):
            s.iterate(warm)
            s.backward()
        a.forward()
        c.forward_stored()
        sa, sc_ = a.scalars(), c.scalars()
        for n in ("step", "fp_failed", "filter_n"):
            assert (sa[n] == sc_[n]).all(), n
        assert (sa["fp_failed"] == 0).all()
        assert np.abs(sa["cost"] / sc_["cost"] - 1).max() < 1e-12 and np.abs(sa["logcost"] / sc_["logcost"] - 1).max() < 1e-12
        for f in (abi.FIELD_X, abi.FIELD_U, abi.FIELD_S) + ((abi.FIELD_Y,) if params.infeas else ()):
            assert helpers.rel(a.get(f), c.get(f)) < 1e-11, f
        a.close()
        c.close()


@pytest.mark.parametrize("name", ["corridor_n8", "free_n5"] if "free_n5" in helpers.CASES else list(helpers.CASES)[:2])
def test_line_search_round_layouts_are_bitwise_equal(name, monkeypatch):
    """How the step sizes of a line search are grouped into sweeps is scheduling only (ddp_wave.h, fwd_pass): one step
    per sweep, or {0} {1,2} {3,4} ... {9,10}.  Each trial's arithmetic and the order in which the trials are judged stay
    those of the sequential search, so every output of both phases must be bit-identical."""
    g, batch = helpers.load_case(name)
    p0, p1 = helpers.case_params(name)
    outs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("DIRECT_EMU_PAIR", mode)
        e0 = emuapi.solve_batch(p0, batch)
        e1 = emuapi.solve_batch(p1, helpers.phase1_batch(g, batch))
        outs.append((e0, e1))
    steps = 0
    for ph in range(2):
        for f in ("rtn", "iter_used", "fwd_passes", "cost", "costq", "T", "poly", "bez", "opterr", "mu"):
            a = np.asarray(getattr(outs[0][ph], f))
            for o in outs[1:]:
                assert np.array_equal(a.view(np.uint8), np.asarray(getattr(o[ph], f)).view(np.uint8)), (ph, f)
        steps += int((outs[0][ph].fwd_passes > 1).sum())
    assert steps > 0


@pytest.mark.parametrize("kind,N,B,seed", [("free", 5, 3, 21), ("corridor", 12, 4, 3), ("free", 30, 2, 5)])
def test_emulated_split_backward_sweep_is_bitwise_the_fused_one(kind, N, B, seed, monkeypatch):
    """bwd_knot is written once and instantiated fused / front / back (ddp_wave.h): the front half of every knot below the
    owner's first claim computed on its own, handed over through a BRec record, and finished by the back half must give
    the very bits of the fused knot - both phases, both storage types, natural exits."""
    batch = problems.make_batch(kind, B, N, seed=seed)
    for params in (abi.phase0_params(), abi.phase1_params()):
        for dtype, c64 in ((np.float64, False), (np.float32, True)):
            res, knots = [], []
            for split in ("0", "1"):
                monkeypatch.setenv("DIRECT_EMU_BSPLIT", split)
                s = emuapi.EmuSolver(params, batch, dtype, c64)
                s.iterate(params.iter_max)
                knots.append(s.split_knots())
                res.append(s.finish())
                s.close()
            assert knots[0] == 0 and knots[1] == B * max(N - 2, 0), knots
            for f in ("rtn", "iter_used", "fwd_passes", "cost", "costq", "opterr", "mu", "T", "poly", "bez"):
                assert np.array_equal(getattr(res[0], f), getattr(res[1], f)), (f, dtype, params.zero_init)


def test_phase1_inputs_reproduce_the_callers_hand_off_where_phase_0_fails():
    """HostBatch.phase1_inputs (the host twin of k_chain): where phase 0 does not return 2 the reference converts its control
    points back with the CALLER's durations (TRP:911-921; DDP:799-812 then 167-193), i.e. coefficient c_i scaled by
    (T_0 / T_1)^(i-1).  Phase 0 cut short so that no problem returns 2; emulated phase 1 from the monomial and from the
    Bezier form of the hand-off against the oracle's fused plan; the unscaled monomials (what round 5 handed over) are far off."""
    batch = problems.make_batch("corridor", 8, 12, seed=123)
    p0, p1 = abi.phase0_params(iter_max=4), abi.phase1_params(infeas=1)
    r0, r1 = refapi.plan_batch(p0, p1, batch)
    assert (r0.rtn != 2).all()
    e0 = emuapi.solve_batch(p0, batch)
    for mono in (True, False):
        e1 = emuapi.solve_batch(p1, batch.phase1_inputs(e0, monomial=mono))
        assert (e1.rtn == r1.rtn).all() and (e1.iter_used == r1.iter_used).all()
        assert np.abs(e1.cost / r1.cost - 1).max() < 1e-8, mono
    old = emuapi.solve_batch(p1, batch.with_init(None, T0=batch.T0, infeas_in=e0.infeas_out.astype(np.uint8), init_poly=e0.poly))
    assert np.abs(old.cost / r1.cost - 1).max() > 1e-2
    # where phase 0 succeeds the factor is exactly one
    q0 = emuapi.solve_batch(abi.phase0_params(), batch)
    ok = q0.rtn == 2
    assert ok.any() and np.array_equal(batch.phase1_inputs(q0).init_poly[ok], q0.poly[ok])
