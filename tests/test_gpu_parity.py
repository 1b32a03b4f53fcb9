"""Parity of the HIP path (through the C-ABI of include/direct_ddp.h) against the oracle and the
golden vectors.  Tolerances: DIRECT_F64: per-pass <= 1e-10 rel, whole solve identical rtn / iteration
counts and cost <= 1e-8 rel (1e-6 on a 64-problem batch).  DIRECT_F32 = float STORAGE with double
arithmetic (DESIGN.md "Precision"): the oracle is given the same float-rounded inputs; per-pass
<= 2e-5 rel (storage rounding of s, y, gains), whole solve cost <= 2e-2 rel at natural exits, which
are stagnation tests (ddp_optimizer.cpp:374) and may fire an iteration apart."""
import numpy as np
import pytest

from direct_amd import abi, problems, solver
from oracle import refapi
from tests import helpers, soak_lib, stuck_lib

pytestmark = pytest.mark.gpu


def make_solver(batch, dtype):
    return solver.DdpSolver(batch.batch, batch.n_seg_max, batch.p_max, dtype)


@pytest.mark.parametrize("kind,N", [("free", 5), ("corridor", 8), ("corridor", 20)])
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-10), (np.float32, 2e-5)])
def test_one_backward_and_forward_pass(built, kind, N, dtype, tol):
    batch = problems.make_batch(kind, 3, N, seed=7).astype(dtype).astype(np.float64)  # same inputs for both
    p0 = abi.phase0_params()
    s = make_solver(batch, dtype)
    s.begin(p0, batch)
    r = [refapi.Stepper(p0, batch, i) for i in range(3)]
    ncm = r[0].ncmax
    assert max(helpers.rel(s.get(abi.FIELD_X)[i], r[i].get(abi.FIELD_X)) for i in range(3)) < max(tol * 1e-2, 1e-7 * (dtype == np.float32))
    assert max(helpers.rel(s.get(abi.FIELD_C)[i][:, :ncm], r[i].get(abi.FIELD_C)) for i in range(3)) < max(tol * 1e-2, 1e-6 * (dtype == np.float32))
    s.backward()
    for q in r:
        q.backward()
    for f in (abi.FIELD_KU, abi.FIELD_KUU, abi.FIELD_KS, abi.FIELD_KY):
        assert max(helpers.rel(s.get(f)[i], r[i].get(f)) for i in range(3)) < tol, f
    sc = s.scalars()
    for i in range(3):
        assert abs(sc["opterr"][i] / r[i].scalars()["opterr"] - 1) < tol
        assert sc["bp_failed"][i] == 0
    s.forward()
    for q in r:
        q.forward()
    sc = s.scalars()
    for i in range(3):
        rs = r[i].scalars()
        assert sc["step"][i] == rs["step"] and sc["fp_failed"][i] == rs["fp_failed"]
        assert abs(sc["cost"][i] / rs["cost"] - 1) < tol
    for f in (abi.FIELD_X, abi.FIELD_U, abi.FIELD_S, abi.FIELD_Y):
        assert max(helpers.rel(s.get(f)[i], r[i].get(f)) for i in range(3)) < tol, f
    s.close()


@pytest.mark.parametrize("name", helpers.CASES)
def test_whole_solve_matches_golden_fp64(built, name):
    g, batch = helpers.load_case(name)
    p0, p1 = helpers.case_params(name)
    s = make_solver(batch, np.float64)
    helpers.check_result(s.solve(p0, batch), g, "p0_", 1e-9, T_tol=1e-8)
    helpers.check_result(s.solve(p1, helpers.phase1_batch(g, batch)), g, "p1_", 1e-8, T_tol=1e-6)
    # the fused two-phase entry point (teach_repeat_planner.cpp:886-921) gives the same answer
    g0, g1 = s.plan(p0, p1, batch)
    helpers.check_result(g0, g, "p0_", 1e-9, T_tol=1e-8)
    helpers.check_result(g1, g, "p1_", 1e-8, T_tol=1e-6)
    s.close()


@pytest.mark.parametrize("name", helpers.CASES)
def test_whole_solve_fp32_tolerance(built, name):
    """DIRECT_F32 (float storage, double arithmetic) on the golden cases: SURVEY.md 8(c)'s whole-solve tolerances for
    fp32 (cost 1e-3, durations 1e-3) with a decade to spare on the cost; measured 1.4e-5 / 3e-4 at worst
    (tests/soak/f32_golden_dev.py).  The distribution over random problems, with its ill-conditioned tail, is bounded in
    tests/test_gpu_soak.py::test_float_storage_deviation_distribution."""
    g, batch = helpers.load_case(name)
    p0, p1 = helpers.case_params(name)
    s = make_solver(batch, np.float32)
    f0 = s.solve(p0, batch)
    assert (f0.rtn == g["p0_rtn"].astype(int)).all()
    assert (f0.iter_used == g["p0_iter_used"].astype(int)).all()
    assert np.abs(f0.cost / g["p0_cost"] - 1).max() < 1e-4
    b1 = helpers.phase1_batch(g, batch)
    f1 = s.solve(p1, b1.with_init(None, T0=b1.T0, infeas_in=b1.infeas_in, init_poly=g["p0_poly"]))
    assert (f1.rtn == g["p1_rtn"].astype(int)).all()
    assert np.abs(f1.cost / g["p1_cost"] - 1).max() < 1e-4, np.abs(f1.cost / g["p1_cost"] - 1).max()
    assert helpers.rel(f1.T, g["p1_T"]) < 1e-3
    # iteration counts: held to what merely STORING the inputs in float does to the oracle itself (every real input moved
    # by -1 / 0 / +1 ulp of a float, four seeds) - not to a constant
    from tests import n100_lib
    c1 = b1.with_init(None, T0=b1.T0, infeas_in=b1.infeas_in, init_poly=g["p0_poly"])
    ctl = max(int(np.abs(refapi.solve_batch(p1, n100_lib.perturb_float_ulp(c1, 7000 + q))[0].iter_used - g["p1_iter_used"].astype(int)).max())
              for q in range(4))
    assert ctl <= 3, ctl   # (the control itself is bounded: a change of the oracle that made it drift would show here, not loosen the line below)
    assert np.abs(f1.iter_used - g["p1_iter_used"].astype(int)).max() <= ctl, (f1.iter_used, g["p1_iter_used"], ctl)
    s.close()


def test_random_batch_against_oracle_fp64(built):
    """64 corridors, N = 12, both phases: identical exits and iteration counts, cost 1e-8."""
    batch = problems.make_batch("corridor", 64, 12, seed=123)
    p0, p1 = abi.phase0_params(), abi.phase1_params()
    r0, r1 = refapi.plan_batch(p0, p1, batch)
    s = make_solver(batch, np.float64)
    g0, g1 = s.plan(p0, p1, batch)
    assert (g0.rtn == r0.rtn).all() and (g0.iter_used == r0.iter_used).all()
    assert (g1.rtn == r1.rtn).all() and (g1.iter_used == r1.iter_used).all()
    # Phase 1 alone, from the ORACLE's phase-0 result through the reference's own hand-off (Bezier control points,
    # TRP:918-921): SURVEY's whole-solve bound, cost 1e-8, control points and durations 1e-6
    q1 = s.solve(p1, batch.phase1_inputs(r0, monomial=False))
    assert (q1.rtn == r1.rtn).all() and (q1.iter_used == r1.iter_used).all()
    assert np.abs(q1.cost / r1.cost - 1).max() < 1e-8
    assert helpers.rel(q1.bez, r1.bez) < 1e-6 and helpers.rel(q1.T, r1.T) < 1e-6
    # ... and the same warm start as monomial coefficients (what the fused plan hands over, k_chain): another rounding of the
    # same numbers, i.e. an input perturbation of one ulp (bounded below with the fused plan)
    m1 = s.solve(p1, batch.phase1_inputs(r0, monomial=True))
    assert (m1.rtn == r1.rtn).all() and (m1.iter_used == r1.iter_used).all()
    # The FUSED plan starts phase 1 from the DEVICE's phase-0 result, 1e-13 from the oracle's, and phase 1 amplifies that
    # like any other perturbation of its inputs (measured on the emulator: 3e-8 with either hand-off form, 3e-9 from the
    # oracle's own control points): bounded by what the oracle shows against itself with inputs moved by one ulp
    c0, c1 = refapi.plan_batch(p0, p1, soak_lib.perturb_ulp(batch, 11))
    ok = (c1.rtn == r1.rtn) & (c1.iter_used == r1.iter_used)
    ctl_cost = np.abs(c1.cost[ok] / r1.cost[ok] - 1).max()
    ctl_bez = max(helpers.rel(c1.bez[ok], r1.bez[ok]), helpers.rel(c1.T[ok], r1.T[ok]))
    assert np.abs(m1.cost / r1.cost - 1).max() < max(1e-8, 30 * ctl_cost), (np.abs(m1.cost / r1.cost - 1).max(), ctl_cost)
    assert np.abs(g1.cost / r1.cost - 1).max() < max(1e-8, 30 * ctl_cost), (np.abs(g1.cost / r1.cost - 1).max(), ctl_cost)
    assert max(helpers.rel(g1.bez, r1.bez), helpers.rel(g1.T, r1.T)) < max(1e-6, 30 * ctl_bez), ctl_bez
    assert helpers.rel(g1.terminal_norm2, r1.terminal_norm2) < max(1e-6, 100 * ctl_bez)
    s.close()


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-8), (np.float32, 2e-3)])
def test_plan_where_phase_0_does_not_find_a_feasible_trajectory(built, dtype, tol):
    """Where phase 0 does not return 2 UpdateTime is skipped and phase 1 converts phase 0's control points back with the
    CALLER's durations (TRP:911-921; ddp_optimizer.cpp:799-812 then 167-193): the warm start is the phase-0 curve in normalised time,
    coefficient c_i scaled by (T_0 / T_1)^(i-1).  direct_ddp_plan_batch applies that factor to the monomial coefficients it
    hands over (k_chain); until round 6 it handed them over unscaled - 80 x off in cost on these problems.  Phase 0 is cut
    short (4 iterations) so that NO problem returns 2."""
    batch = problems.make_batch("corridor", 32, 12, seed=123).astype(dtype).astype(np.float64)
    p0, p1 = abi.phase0_params(iter_max=4), abi.phase1_params(infeas=1)
    r0, r1 = refapi.plan_batch(p0, p1, batch)
    assert (r0.rtn != 2).all()
    s = make_solver(batch, dtype)
    g0, g1 = s.plan(p0, p1, batch)
    s.close()
    assert (g0.rtn == r0.rtn).all() and (g0.iter_used == r0.iter_used).all()
    assert (g1.rtn == r1.rtn).all()
    if dtype == np.float64:
        assert (g1.iter_used == r1.iter_used).all()
    same = g1.iter_used == r1.iter_used
    assert same.mean() > 0.8
    assert np.abs(g1.cost[same] / r1.cost[same] - 1).max() < tol
    assert helpers.rel(g1.T[same], r1.T[same]) < 100 * tol


def test_ragged_batch_and_handle_reuse(built):
    a = problems.make_batch("corridor", 2, 9, seed=31)
    n_seg = np.array([9, 5], np.int32)
    xd = a.xd.copy()
    xd[1, :3] = a.seeds[1, 5]
    T0 = problems.time_allocation(n_seg, a.x0[:, :3], xd[:, :3], a.seeds)
    batch = abi.HostBatch(n_seg, a.x0, xd, T0, a.n_planes, a.planes, seeds=a.seeds)
    p0 = abi.phase0_params()
    r, _ = refapi.solve_batch(p0, batch)
    s = make_solver(batch, np.float64)
    for _ in range(2):  # the handle is reusable (the reference object is single-use, quirk Q9)
        e = s.solve(p0, batch)
        assert (e.rtn == r.rtn).all() and (e.iter_used == r.iter_used).all()
        assert np.abs(e.cost / r.cost - 1).max() < 1e-9
        assert helpers.rel(e.T[1, :5], r.T[1, :5]) < 1e-8
    s.close()


def test_many_planes_rpl3_and_rpl4(built):
    """P up to 20 (nc = 175, 3 rows per lane) and 30 (nc = 235, 4 rows per lane)."""
    for pmax in (20, 30):
        batch = problems.make_batch("corridor", 4, 6, seed=5, p_max=pmax)
        assert batch.n_planes.max() > 12
        p0 = abi.phase0_params()
        r, _ = refapi.solve_batch(p0, batch)
        s = make_solver(batch, np.float64)
        e = s.solve(p0, batch)
        assert (e.rtn == r.rtn).all() and (e.iter_used == r.iter_used).all()
        assert np.abs(e.cost / r.cost - 1).max() < 1e-8
        s.close()


def test_error_paths(built):
    batch = problems.make_batch("free", 4, 5, seed=1)
    s = make_solver(batch, np.float32)
    with pytest.raises(solver.DirectError) as e:
        s.solve(abi.phase0_params(time_power=3), batch)
    assert e.value.status == abi.DIRECT_ERR_INVALID
    noseeds = abi.HostBatch(batch.n_seg, batch.x0, batch.xd, batch.T0, batch.n_planes, batch.planes, dtype=np.float32)
    with pytest.raises(solver.DirectError) as e:
        s.solve(abi.phase1_params(line_init=1), noseeds)   # line-init needs the polytope seeds
    assert e.value.status == abi.DIRECT_ERR_INVALID
    with pytest.raises(solver.DirectError) as e:
        s.solve(abi.phase1_params(), batch)          # warm start without init_bez / init_poly
    assert e.value.status == abi.DIRECT_ERR_INVALID
    big = problems.make_batch("free", 8, 5, seed=1)
    with pytest.raises(solver.DirectError):
        s.solve(abi.phase0_params(), big)            # batch > max_batch
    bad = problems.make_batch("free", 4, 5, seed=1)
    bad.n_planes[0, 0] = 99
    with pytest.raises(solver.DirectError):
        s.solve(abi.phase0_params(), bad)
    s.close()


def test_best_cost_reduction(built):
    rng = np.random.default_rng(0)
    cost = rng.uniform(1, 100, 1000).astype(np.float32)
    rtn = rng.integers(-4, 3, 1000).astype(np.int32)
    s = solver.DdpSolver(1000, 4, 6, np.float32)
    i, c = s.best_cost(cost, rtn)
    want = np.where(rtn >= 0, cost, np.inf)
    assert i == int(np.argmin(want)) and abs(c - float(want.min())) < 1e-6
    s.close()


def test_cpp_host_shim_two_phase_plan(built, tmp_path):
    """The reference-shaped C++ class (direct_amd/host/ddp_optimizer.hpp) driven like fastTrajPlanning."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "test_shim")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(root, "tests/cpp/test_shim.cpp"), "-o", exe,
                           "-L" + os.path.join(root, "direct_amd/lib"), "-ldirect_ddp",
                           "-L" + os.path.join(root, "oracle"), "-ldirect_ref",
                           "-Wl,-rpath," + os.path.join(root, "direct_amd/lib") + ":" + os.path.join(root, "oracle") + ":/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "PASS" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("kind", ["free", "corridor"])
def test_line_initialisation_matches_oracle(built, kind):
    """line_init_flag = true (DDP:194-248, 255-269, 283-286, 398-409) through the C-ABI, fp64."""
    batch = problems.make_batch(kind, 6, 7, seed=33)
    batch.T0[1] *= 0.35
    batch.T0[2] *= 0.1
    p = abi.phase1_params(line_init=1, infeas=1, iter_max=40)
    s = make_solver(batch, np.float64)
    s.begin(p, batch)
    for i in range(batch.batch):
        r = refapi.Stepper(p, batch, i)
        assert helpers.rel(s.get(abi.FIELD_U)[i], r.get(abi.FIELD_U)) < 1e-12
        assert helpers.rel(s.get(abi.FIELD_X)[i], r.get(abi.FIELD_X)) < 1e-12
        sc, sr = s.scalars(), r.scalars()
        assert int(sc["infeas"][i]) == int(sr["infeas"]) and int(sc["reg"][i]) == 10
        assert abs(sc["cost"][i] / sr["cost"] - 1) < 1e-11
    g = s.solve(p, batch)
    o, _ = refapi.solve_batch(p, batch)
    assert (g.rtn == o.rtn).all() and (g.iter_used == o.iter_used).all()
    assert (g.line_failed_out == o.line_failed_out).all() and (g.infeas_out == o.infeas_out).all()
    assert np.abs(g.cost / o.cost - 1).max() < 1e-8
    s.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_output_sampling_matches_oracle(built, dtype):
    """direct_traj_sample_batch (k_sample) against the oracle's literal sampling loop and the golden vectors."""
    import os
    for case in ("corridor_n8", "free_n5", "config1_n50"):
        g = np.load(os.path.join(helpers.GOLDEN_DIR, "sample_" + case + ".npz"))
        B, nm = g["T"].shape
        s = solver.DdpSolver(B, nm, 6, dtype)
        bez, T = g["bez"].astype(dtype), g["T"].astype(dtype)
        d = s.sample(g["n_seg"], bez, T, float(g["dt"]), int(g["capacity"]))
        o = refapi.sample_batch(g["n_seg"], bez, T, float(g["dt"]), int(g["capacity"]))   # same (rounded) inputs
        tol = 1e-12 if dtype == np.float64 else 2e-6                                      # float: output rounding
        assert (d["count"] == o["count"]).all() and (d["seg_first"] == o["seg_first"]).all()
        for k in ("pos", "vel", "acc", "length", "vmax", "amax"):
            assert helpers.rel(d[k], o[k]) < tol, (case, k)
        if dtype == np.float64:
            assert (d["count"] == g["count"]).all()
            assert helpers.rel(d["pos"], g["pos"]) < 1e-12 and helpers.rel(d["length"], g["length"]) < 1e-12
        # more than 64 samples in one segment (chunking), capacity clamp, negative duration
        T2 = T.copy()
        T2[0, 0] = 9.0
        d2 = s.sample(g["n_seg"], bez, T2, 0.05, 100)
        o2 = refapi.sample_batch(g["n_seg"], bez, T2, 0.05, 100)
        assert (d2["count"] == o2["count"]).all() and d2["count"][0] > 180
        assert helpers.rel(d2["pos"], o2["pos"]) < tol and helpers.rel(d2["length"], o2["length"]) < tol
        T2[0, 1] = -1.0
        assert s.sample(g["n_seg"], bez, T2, 0.05, 100)["count"][0] == -1
        s.close()


def test_recorded_corridor_replay_in_one_batch(built):
    """The benchmark replay of corridorRecCallBack (TRP:316-320: the first n polytopes, n = 2..64, one solver
    run each) as ONE ragged batch through direct_ddp_plan_batch, against the oracle problem by problem."""
    import os
    from direct_amd import corridor_io
    cor, _ = corridor_io.unpack(open(os.path.join(helpers.GOLDEN_DIR, "corridor_msg.bin"), "rb").read(), 64, 12)
    batch = corridor_io.replay_batch(cor, n_first=2)
    p0, p1 = abi.phase0_params(), abi.phase1_params()
    s = make_solver(batch, np.float64)
    g0, g1 = s.plan(p0, p1, batch)
    s.close()
    r0, r1 = refapi.plan_batch(p0, p1, batch)
    assert (g0.rtn == r0.rtn).all() and (g0.iter_used == r0.iter_used).all()
    assert (g1.rtn == r1.rtn).all() and (g1.iter_used == r1.iter_used).all()
    assert np.abs(g1.cost / r1.cost - 1).max() < 1e-6 and helpers.rel(g1.T, r1.T) < 1e-6
    assert helpers.rel(g1.bez, r1.bez) < 1e-5


def check_against_controls(r, n):
    """Assertions over a tests/n100_lib.py report: the device is bounded BY the oracle's own reaction to a one-ulp
    perturbation of its inputs (double ulp for double storage, float ulp for float storage), computed in this very test
    on these very problems - not by a constant chosen to let the run pass."""
    for ph in ("phase0", "phase1"):
        d64, d32 = r["device_f64"][ph], r["device_f32"][ph]
        c64 = [c[ph] for c in r["control_double_ulp"]]
        c32 = [c[ph] for c in r["control_float_ulp"]]
        cmax = lambda cs: max([c["cost_dev_q50_q90_max"][2] for c in cs if c["cost_dev_q50_q90_max"]] or [0.0])
        # double storage
        assert d64["same_feasibility"] >= min(c["same_feasibility"] for c in c64), (ph, d64, c64)
        assert d64["same_outcome"] >= min(c["same_outcome"] for c in c64), (ph, d64, c64)
        assert d64["n_cost_dev_below_1e_8"] >= min(c["n_cost_dev_below_1e_8"] for c in c64), (ph, d64, c64)
        assert d64["cost_dev_q50_q90_max"][0] < 1e-9, (ph, d64)                      # the bulk: SURVEY 8(c)'s 1e-8 with a decade to spare
        assert d64["cost_dev_q50_q90_max"][2] <= max(1e-8, 15 * cmax(c64)), (ph, d64, c64)   # the tail: what the algorithm does to one ulp (the maximum over five seeds
                                                                                            # of eight problems is itself a noisy statistic: measured ratios 0.3 .. 11)
        if "per_problem" in d64:
            # Problem by problem: where the oracle CONVERGES (rtn 1 / 2) and every one-ulp control reproduces its return
            # code and iteration count, nothing is being amplified - the device is held to SURVEY 8(c) exactly there:
            # identical rtn and iterations, cost / durations / control points within 1e-6.
            pp = d64["per_problem"]
            stable = [pp["ref_rtn"][i] in (1, 2) and all(c["per_problem"]["same"][i] for c in c64) for i in range(n)]
            assert sum(stable) >= 1 or not any(r in (1, 2) for r in pp["ref_rtn"]), (ph, pp["ref_rtn"], c64)
            for i in range(n):
                if stable[i]:
                    assert pp["same"][i], (ph, i, pp)
                    assert pp["cost_dev"][i] < 1e-6 and pp["T_dev"][i] < 1e-6 and pp["bez_dev"][i] < 1e-5, (ph, i, pp)
        # float storage: SURVEY 8(c)'s fp32 tolerances wherever the float-ulp control keeps them
        assert d32["same_feasibility"] >= min(c["same_feasibility"] for c in c32), (ph, d32, c32)
        assert d32["same_rtn"] >= min(c["same_rtn"] for c in c32), (ph, d32, c32)
        assert d32["same_outcome"] >= min(c["same_outcome"] for c in c32), (ph, d32, c32)
        assert d32["n_cost_dev_below_1e_3"] >= min(c["n_cost_dev_below_1e_3"] for c in c32), (ph, d32, c32)
        assert d32["cost_dev_q50_q90_max"][0] < 1e-5, (ph, d32)
        assert d32["cost_dev_q50_q90_max"][2] <= max(1e-3, 3 * cmax(c32)), (ph, d32, c32)


@pytest.mark.parametrize("p_max", [20, 32, 44, 54, 65, 76, 100, 128])
def test_many_planes_per_polytope(built, p_max):
    """P up to 20 / 32 / 44 / 54 / 65 / 76 planes per polytope (nc = 175 .. 511): the kernels with three to eight row
    slots per lane, both storage types, both phases (real voxel clusters reach 69 planes: profiles/r03_hull_soak.json).
    Eight problems; phase 1 of every implementation starts from the ORACLE's phase-0 result, so that both phases compare
    identical inputs.  Problems that run into the iteration limit end wherever iteration 100 leaves them - how far apart
    that may be is MEASURED here, by the oracle against itself with its inputs moved by one ulp (three seeds), and the
    device is held to that."""
    from tests import n100_lib
    batch = helpers.with_extra_planes(problems.make_batch("corridor", 8, 9, seed=41), p_max, seed=p_max)
    assert batch.n_planes.max() >= p_max - 2

    def dev(dtype):
        def solve(params, b):
            s = solver.DdpSolver(b.batch, b.n_seg_max, b.p_max, dtype)
            r = s.solve(params, b)
            s.close()
            return r
        return solve
    r = n100_lib.batch_report(batch, dev(np.float64), dev(np.float32), control_seeds=(11, 12, 13, 14, 15), detail=True)
    check_against_controls(r, 8)


def test_containment_audit_of_the_samples(built):
    """cmax of direct_traj_sample_batch: the worst plane value over the samples of each trajectory, against a
    numpy evaluation over the oracle's samples; solved corridors are inside (convex-hull property of the
    Bezier control points, which is what the constraints act on)."""
    batch = problems.make_batch("corridor", 12, 9, seed=91)
    s = make_solver(batch, np.float64)
    g0, g1 = s.plan(abi.phase0_params(), abi.phase1_params(), batch)
    d = s.sample(batch.n_seg, g1.bez, g1.T, 0.05, 2048, derivs=0, n_planes=batch.n_planes, planes=batch.planes)
    s.close()
    o = refapi.sample_batch(batch.n_seg, g1.bez, g1.T, 0.05, 2048, derivs=0, n_planes=batch.n_planes, planes=batch.planes)
    assert (d["count"] == o["count"]).all()
    assert np.abs(d["cmax"] - o["cmax"]).max() < 1e-10
    ok = g1.rtn >= 0
    assert ok.any() and (d["cmax"][ok] < 1e-6).all()
    # a trajectory pushed out of its corridor is flagged
    shifted = g1.bez.copy()
    shifted[0, 2, 0:6] += 5.0 / g1.T[0, 2]          # x control points of segment 2 moved by 5 m
    s = make_solver(batch, np.float64)
    bad = s.sample(batch.n_seg, shifted, g1.T, 0.05, 2048, derivs=0, n_planes=batch.n_planes, planes=batch.planes)
    s.close()
    assert bad["cmax"][0] > 1.0 and np.array_equal(bad["cmax"][1:], d["cmax"][1:])


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-9), (np.float32, 5e-5)])
def test_forward_pass_after_a_backward_pass_that_failed_part_way(built, dtype, tol):
    """Stepwise entry points on a stale row cache: with float storage the feasible-mode forward trials read
    c and s / c of the nominal iterate from scratch arrays that only a COMPLETED backward sweep fills.  Here the sweep
    is made to fail at knot 2 (negative slacks there: Quu - cu' (s/c) cu loses definiteness, LLT reports it,
    ddp_optimizer.cpp:595-600) right after an accepted iteration has moved the iterate to another buffer; the forward
    pass that follows must still evaluate its trials against the CURRENT iterate, as the reference's does
    (ddp_optimizer.cpp:696) - compared with the oracle driven through the same sequence."""
    batch = problems.make_batch("corridor", 3, 8, seed=11).astype(dtype).astype(np.float64)
    p0, p1 = abi.phase0_params(), abi.phase1_params()
    r0, _ = refapi.solve_batch(p0, batch)
    b1 = batch.with_init(None, T0=np.where((r0.rtn == 2)[:, None], r0.T, batch.T0), infeas_in=np.zeros(3, np.uint8),
                         init_poly=r0.poly).astype(dtype).astype(np.float64)
    s = make_solver(b1, dtype)
    s.begin(p1, b1)
    r = [refapi.Stepper(p1, b1, i) for i in range(3)]
    s.iterate(1)
    for q in r:
        q.iterate(1)
    assert (s.scalars()["infeas"] == 0).all() and (s.scalars()["fp_failed"] == 0).all()
    S = s.get(abi.FIELD_S).astype(np.float64)
    for i in range(3):
        S[i, 2, :6 * int(b1.n_planes[i, 2]) + 55] = -1.0e3   # the rows that exist at knot 2
    s.set(abi.FIELD_S, S)
    for i, q in enumerate(r):
        q.set(abi.FIELD_S, S[i][:, :q.ncmax])
    s.backward()
    for q in r:
        q.backward()
    assert (s.scalars()["bp_failed"] == 1).all() and all(q.scalars()["bp_failed"] == 1 for q in r)
    s.forward()
    for q in r:
        q.forward()
    sc = s.scalars()
    for i, q in enumerate(r):
        rs = q.scalars()
        assert sc["step"][i] == rs["step"] and sc["fp_failed"][i] == rs["fp_failed"], (i, sc["step"][i], rs["step"])
        assert abs(sc["cost"][i] / rs["cost"] - 1) < tol
    ncm = r[0].ncmax
    for f in (abi.FIELD_X, abi.FIELD_U):
        assert max(helpers.rel(s.get(f)[i], r[i].get(f)) for i in range(3)) < tol, f
    assert max(helpers.rel(s.get(abi.FIELD_S)[i][:, :ncm], r[i].get(abi.FIELD_S)) for i in range(3)) < 20 * tol
    s.close()


@pytest.mark.parametrize("name", helpers.EXIT_CASES)
def test_whole_solve_matches_exit_goldens_fp64(built, name):
    """Every way out of the outer loop and every failure branch inside it (DDP:295-412) through the C-ABI, against the
    vectors of the NumPy restatement (tests/golden/make_exit_golden.py): the loop running out, rtn -3 (negative input
    duration), rtn -4 in the first iteration, LLT failures with recovery at a larger regulariser, both line-init exits."""
    g, batch, p = helpers.load_exit_case(name)
    s = make_solver(batch, np.float64)
    helpers.check_exit_result(s.solve(p, batch), g, 1e-8)
    s.close()


@pytest.mark.parametrize("name", helpers.EXIT_CASES)
def test_whole_solve_matches_exit_goldens_float_storage(built, name):
    """the same with float storage on float-rounded inputs: identical return codes, iteration counts within what the
    oracle itself shows between double and float-rounded inputs, cost 1e-3"""
    g, batch, p = helpers.load_exit_case(name)
    b32 = batch.astype(np.float32).astype(np.float64)
    ctl, _ = refapi.solve_batch(p, b32)
    s = make_solver(batch, np.float32)
    r = s.solve(p, b32)
    s.close()
    assert (r.rtn == ctl.rtn).all() and (r.line_failed_out == ctl.line_failed_out).all()
    slack = np.abs(ctl.iter_used - g["out_iter_used"]).max()
    assert np.abs(r.iter_used - ctl.iter_used).max() <= slack
    same = r.iter_used == ctl.iter_used
    assert np.abs(r.cost[same] / ctl.cost[same] - 1).max() < 1e-3


def test_forced_stuck_golden(built):
    """exit_forced_stuck.npz: the backward pass stuck mid-solve and the stale-gain forward pass that follows, as the
    NumPy restatement ran it - the second witness of the rtn = -4 path next to the C oracle (tests/stuck_lib.py)."""
    g, batch, p = helpers.load_exit_case("exit_forced_stuck")
    s = make_solver(batch, np.float64)
    s.begin(p, batch)
    helpers.run_forced_stuck(s, g, 1e-9)
    s.close()


@pytest.mark.parametrize("scenario", range(len(stuck_lib.scenarios())))
def test_forward_pass_after_a_stuck_backward_pass_uses_the_stored_gains(built, scenario):
    """rtn = -4 with the reference's own last forward pass (DDP:297-311 -> 647-778 -> 392-396): the knots the retry sequence
    never reached keep the gains of the last COMPLETED sweep (another iterate, another barrier parameter), the hot kernel
    parks the trajectory and k_stuck / Wave::stale_fwd_pass finishes it.  Forced a few iterations into well-conditioned
    solves (tests/stuck_lib.py) through the C-ABI's stepwise calls, both storage types: every decision of the last
    trip and the iterate it leaves; the round-5 kernels left the oracle by 3 - 8 % in cost on the problems that accept a
    step.  The twin of tests/test_emu_parity.py's test of the same name."""
    name, p, kind, K, y_inject, zero_bez = stuck_lib.scenarios()[scenario]
    batch = problems.make_batch(kind, 8, 10, seed=77)
    if zero_bez:
        batch = batch.with_init(np.zeros((8, 10, 18)))
    sc = stuck_lib.Scenario(p, batch, K, y_inject)
    if name == "phase0":
        assert sc.accepted().sum() >= 3
    for dtype, tol in ((np.float64, 1e-9), (np.float32, 1e-3)):
        s = make_solver(batch, dtype)
        s.begin(p, batch)
        out = sc.run(s)
        sc.check(out, tol)
        # ... and what the caller sees: the getters after finish (cost 1e-8, control points 1e-6 for double storage)
        res = s.finish()
        assert (res.rtn == -4).all() and (res.iter_used == K).all() and (res.fwd_passes == K + 1).all()
        u = sc.usable
        ctol, btol = (1e-8, 1e-6) if dtype == np.float64 else (1e-3, 1e-2)
        assert np.abs(res.cost[u] / np.array([q["cost"] for q in sc.sc])[u] - 1).max() < ctol
        for i in np.where(u)[0]:
            U = sc.oracle[i].get(abi.FIELD_U)
            assert helpers.rel(res.T[i], U[:, 9]) < btol and helpers.rel(res.poly[i][:, 9:], U[:, :9]) < btol
        s.close()
    sc.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_backward_pass_stuck_in_the_first_iteration(built, dtype):
    """rtn = -4 at iteration 0 of feasible-mode solves from an infeasible start (natural, no forcing): no sweep ever
    completed, the stale gains are the zeros of DDP:154-159, every trial dies at the fraction-to-boundary rule and the
    iterate is the initial roll.  Whole solves against the oracle: return codes, iteration counts, cost, control points."""
    b = problems.make_batch("corridor", 32, 8, seed=1)
    bb = b.with_init(np.zeros((32, 8, 18)), T0=b.T0 * 3.0, infeas_in=np.zeros(32, np.uint8)).astype(dtype).astype(np.float64)
    p = abi.phase1_params(iter_max=60)
    r, _ = refapi.solve_batch(p, bb)
    stuck = r.rtn == -4
    assert stuck.sum() >= 1
    s = make_solver(bb, dtype)
    g = s.solve(p, bb)
    s.close()
    assert (g.rtn[stuck] == -4).all() and (g.iter_used[stuck] == r.iter_used[stuck]).all() and (g.fwd_passes[stuck] == r.fwd_passes[stuck]).all()
    tol = 1e-8 if dtype == np.float64 else 1e-5
    assert np.abs(g.cost[stuck] / r.cost[stuck] - 1).max() < tol
    assert helpers.rel(g.bez[stuck], r.bez[stuck]) < 100 * tol and helpers.rel(g.T[stuck], r.T[stuck]) < 100 * tol


@pytest.mark.parametrize("mode", ["infeasible", "feasible"])
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-11), (np.float32, 1e-5)])
def test_stored_gain_forward_pass_equals_the_regular_one_when_every_gain_is_current(built, mode, dtype, tol):
    """direct_ddp_forward_pass_stored after a COMPLETED backward pass: every gain belongs to the current iterate, the
    stored-gain form (Wave::stale_fwd_pass) must reproduce the regular forward pass (run_round) - both modes."""
    g, batch = helpers.load_case("corridor_n8")
    params = abi.phase0_params() if mode == "infeasible" else abi.phase1_params()
    b = batch if params.zero_init else helpers.phase1_batch(g, batch)
    for warm in (0, 2):
        a, c = make_solver(b, dtype), make_solver(b, dtype)
        for s in (a, c):
            s.begin(params, b)
            s.iterate(warm)
            s.backward()
        a.forward()
        c.forward_stored()
        sa, sc_ = a.scalars(), c.scalars()
        for n in ("step", "fp_failed", "filter_n"):
            assert (sa[n] == sc_[n]).all(), n
        assert (sa["fp_failed"] == 0).all()
        assert np.abs(sa["cost"] / sc_["cost"] - 1).max() < max(tol, 1e-7 * (dtype == np.float32))
        for f in (abi.FIELD_X, abi.FIELD_U, abi.FIELD_S) + ((abi.FIELD_Y,) if params.infeas else ()):
            assert helpers.rel(a.get(f), c.get(f)) < tol, f
        a.close()
        c.close()
