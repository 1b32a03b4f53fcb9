"""BASELINE-size batches (B = 4096, N = 100): the oracle cannot be run on all of them in seconds,
so parity is established (a) exactly against the oracle on a random SAMPLE of the batch and (b)
through size-independent properties of the whole batch: determinism, sharding invariance,
dynamics / continuity of the returned polynomials, feasibility and duration bounds at exit,
Bezier <-> monomial consistency of the two output formats, monotone objective."""
import numpy as np
import pytest

from direct_amd import abi, problems, solver
from oracle import refapi
from tests import helpers

pytestmark = pytest.mark.gpu
B, N = 4096, 100


@pytest.fixture(scope="module")
def free_batch():
    return problems.make_batch("free", B, N, seed=1000)


@pytest.fixture(scope="module")
def corridor_batch():
    return problems.make_batch("corridor", B, N, seed=1000)


def poly_eval(poly, T, order):
    """value of the order-th derivative of every segment's quintic at t = T; poly [.., 18] = c0xyz..c5xyz"""
    C = poly.reshape(poly.shape[:-1] + (6, 3))
    out = np.zeros(poly.shape[:-1] + (3,))
    for i in range(order, 6):
        f = 1.0
        for q in range(order):
            f *= (i - q)
        out += f * C[..., i, :] * T[..., None] ** (i - order)
    return out


def check_properties(batch, res, fp_tol):
    n = batch.n_seg_max
    B = batch.batch
    ok = res.rtn >= 0
    assert ok.mean() > 0.85
    # continuity of position / velocity / acceleration across segments = the dynamics roll-out
    for order in range(3):
        end = poly_eval(res.poly[:, :-1], res.T[:, :-1], order)
        coef = {0: 1.0, 1: 1.0, 2: 2.0}[order]
        start = coef * res.poly[:, 1:].reshape(B, n - 1, 6, 3)[:, :, order, :]
        scale = np.abs(start).max() + 1.0
        assert np.abs(end - start)[ok].max() < fp_tol * scale
    # the trajectory starts at x0
    assert np.abs(res.poly[:, 0, :3] - batch.x0[:, :3]).max() < fp_tol * 20
    # both output formats describe the same curve: Bezier control point 0 / 5 (time-scaled) = p(0) / p(T)
    bez = res.bez.reshape(B, n, 3, 6)
    p0 = res.poly[..., :3]
    pT = poly_eval(res.poly, res.T, 0)
    assert np.abs(bez[..., 0] * res.T[..., None] - p0)[ok].max() < fp_tol * 50
    assert np.abs(bez[..., 5] * res.T[..., None] - pT)[ok].max() < fp_tol * 50
    # durations respect T >= 0.3 (up to the fraction-to-boundary margin) and costs are finite
    assert res.T[ok].min() > 0.29
    assert np.isfinite(res.cost[ok]).all() and (res.cost[ok] > 0).all()


def test_free_space_full_batch_fp32(built, free_batch):
    s = solver.DdpSolver(B, N, free_batch.p_max, np.float32)
    p0, p1 = abi.phase0_params(), abi.phase1_params()
    g0, g1 = s.plan(p0, p1, free_batch)
    assert (g0.rtn == 2).all()            # zero init is feasible in free space: phase 0 exits at once
    check_properties(free_batch, g1, 2e-4)
    # determinism: a second run is bit-identical
    h0, h1 = s.plan(p0, p1, free_batch)
    assert np.array_equal(g1.bez, h1.bez) and np.array_equal(g1.rtn, h1.rtn) and np.array_equal(g1.cost, h1.cost)
    # sharding invariance: solving a slice alone gives bit-identical results to solving it in the batch
    sub = free_batch.select(np.arange(512, 768))
    s0, s1 = s.plan(p0, p1, sub)
    assert np.array_equal(s1.bez, g1.bez[512:768]) and np.array_equal(s1.iter_used, g1.iter_used[512:768])
    # sample parity against the fp64 oracle: 32 problems spread over the batch, phase 1 from IDENTICAL inputs (the
    # device's own phase-0 result, which the fused plan hands over on the device: durations where rtn == 2, monomial
    # coefficients, feasibility flags).  With the iterate stored as hi + lo float pairs float storage follows the fp64
    # iterates through the stagnation exits (ddp_optimizer.cpp:374) too: SURVEY.md 8(c)'s fp32 tolerances hold for EVERY
    # problem of the sample.  (r02 needed a 15 % allowance here: single-float iterates left the oracle at the exits of
    # 11 % of these problems.  One problem of this sample is a knife edge of the ALGORITHM - at the barrier update that
    # precedes its exit opterr lands at 25 or at 76 around the exit rule's threshold of 50, ddp_optimizer.cpp:340-378,
    # depending on a 6e-8 perturbation of the INPUTS, and the cost at exit differs by 2.8x -, which is why both sides
    # must be given bit-identical inputs; tests/test_gpu_n100.py has the controls.)
    idx = np.arange(0, B, 128)
    b1 = free_batch.astype(np.float32).with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, free_batch.T0.astype(np.float32)),
                                                 infeas_in=g0.infeas_out, init_poly=g0.poly).select(idx)
    r1, _ = refapi.solve_batch(p1, b1.astype(np.float64))
    assert (g1.rtn[idx] == r1.rtn).all()
    dev = np.abs(g1.cost[idx] / r1.cost - 1)
    # The bulk within 1e-6; the tail - knife-edge problems that flip with a 1e-7 change anywhere in the float path - is
    # held to the CONTROL: the oracle against itself on these very problems with every input moved by one ulp of a float
    # (two seeds): as many problems within 1e-3, and iteration counts no further apart, than the controls themselves show
    from tests import n100_lib
    ctl = [refapi.solve_batch(p1, n100_lib.perturb_float_ulp(b1.astype(np.float64), 500 + q))[0] for q in range(2)]
    c_ok = min(int((np.abs(c.cost / r1.cost - 1) < 1e-3).sum()) for c in ctl)
    c_it = max(int(np.sort(np.abs(c.iter_used - r1.iter_used))[-2]) for c in ctl)
    assert np.median(dev) < 1e-6 and (dev < 1e-3).sum() >= c_ok, (np.sort(dev)[::-1][:5], c_ok)
    assert np.sort(np.abs(g1.iter_used[idx] - r1.iter_used))[-2] <= max(1, c_it), (c_it,)
    idx = np.array([0, 777, 2048, 4095])
    r0, r1 = refapi.plan_batch(p0, p1, free_batch.select(idx))
    # at a FIXED iteration count the two precisions follow the same path much more closely
    pf = abi.phase1_params(iter_max=6, fixed_iters=1)
    b1 = free_batch.select(idx).with_init(None, T0=np.where((r0.rtn == 2)[:, None], r0.T, free_batch.T0[idx]),
                                          infeas_in=r0.infeas_out, init_poly=r0.poly)
    rf, _ = refapi.solve_batch(pf, b1)
    s4 = solver.DdpSolver(len(idx), N, b1.p_max, np.float32)
    gf = s4.solve(pf, b1)
    s4.close()
    assert np.abs(gf.cost / rf.cost - 1).max() < 1e-3, np.abs(gf.cost / rf.cost - 1)
    s.close()


def test_corridor_full_batch_fp32(built, corridor_batch):
    s = solver.DdpSolver(B, N, corridor_batch.p_max, np.float32)
    p0, p1 = abi.phase0_params(), abi.phase1_params()
    g0, g1 = s.plan(p0, p1, corridor_batch)
    found = g0.rtn == 2
    assert found.mean() > 0.25   # the fp64 oracle also needs > 50 iterations on most N = 100 corridors
    # phase 0 found a feasible trajectory: every (shifted) constraint is below 2e-4 for those problems
    s.begin(p0, corridor_batch)
    s.iterate(p0.iter_max)
    c = s.get(abi.FIELD_C)
    nc = 6 * corridor_batch.n_planes + 55
    mask = np.arange(c.shape[2])[None, None, :] < nc[:, :, None]
    cmax = np.where(mask, c, -np.inf).max(axis=(1, 2))
    assert (cmax[found] < 2e-4).all()
    check_properties(corridor_batch, g1, 5e-4)
    s.close()


def test_fixed_iteration_benchmark_mode_counts(built, free_batch):
    """The benchmark workload: every problem executes exactly iter_max forward passes."""
    s = solver.DdpSolver(B, N, free_batch.p_max, np.float32)
    g0 = s.solve(abi.phase0_params(), free_batch)
    b1 = free_batch.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, free_batch.T0), infeas_in=g0.infeas_out,
                              init_poly=g0.poly)
    g1 = s.solve(abi.phase1_params(iter_max=20, fixed_iters=1), b1)
    assert (g1.fwd_passes == 20).all() and (g1.iter_used == 20).all()
    ms, n = s.last_kernel_ms()
    assert ms > 0 and n == 1
    s.close()


def test_fp64_sample_of_full_size_problems(built, corridor_batch):
    """N = 100 corridors in fp64: exact agreement with the oracle on a sample of the config-3 batch."""
    idx = np.array([3, 1500, 4000])
    sub = corridor_batch.select(idx)
    p0, p1 = abi.phase0_params(), abi.phase1_params()
    r0, r1 = refapi.plan_batch(p0, p1, sub)
    s = solver.DdpSolver(3, N, sub.p_max, np.float64)
    g0, g1 = s.plan(p0, p1, sub)
    assert (g0.rtn == r0.rtn).all() and (g0.iter_used == r0.iter_used).all()
    assert (g1.rtn == r1.rtn).all()
    same = g1.iter_used == r1.iter_used
    assert same.all()
    # phase 1 alone from the ORACLE's phase-0 result, through the reference's own hand-off (Bezier, TRP:918-921) and as the
    # monomial coefficients the fused plan hands over: cost 1e-8, durations 1e-6
    q1 = s.solve(p1, sub.phase1_inputs(r0, monomial=False))
    assert (q1.rtn == r1.rtn).all() and (q1.iter_used == r1.iter_used).all()
    assert np.abs(q1.cost / r1.cost - 1).max() < 1e-8 and helpers.rel(q1.T, r1.T) < 1e-6
    m1 = s.solve(p1, sub.phase1_inputs(r0, monomial=True))
    assert (m1.rtn == r1.rtn).all() and (m1.iter_used == r1.iter_used).all()
    # the fused plan starts phase 1 from the device's phase-0 result (1e-13 from the oracle's), which 100 knots x 20 .. 100
    # iterations amplify like any input perturbation: bounded by the oracle against itself, inputs moved by one ulp
    from tests import soak_lib
    devs = []
    for cs in (3, 4, 5):
        c0, c1 = refapi.plan_batch(p0, p1, soak_lib.perturb_ulp(sub, cs))
        ok = (c1.rtn == r1.rtn) & (c1.iter_used == r1.iter_used)
        devs.append((np.abs(c1.cost[ok] / r1.cost[ok] - 1).max() if ok.any() else 1.0, helpers.rel(c1.T[ok], r1.T[ok]) if ok.any() else 1.0))
    ctl_cost, ctl_T = max(d[0] for d in devs), max(d[1] for d in devs)
    assert np.abs(m1.cost / r1.cost - 1).max() < max(1e-8, 30 * ctl_cost), (np.abs(m1.cost / r1.cost - 1).max(), ctl_cost)
    assert np.abs(g1.cost / r1.cost - 1).max() < max(1e-8, 30 * ctl_cost), (np.abs(g1.cost / r1.cost - 1).max(), ctl_cost)
    assert helpers.rel(g1.T, r1.T) < max(1e-6, 30 * ctl_T), (helpers.rel(g1.T, r1.T), ctl_T)
    s.close()


def test_device_memory_interface_matches_host_interface(built, free_batch):
    """Inputs and outputs resident in HBM (torch tensors as plain device pointers)."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    import ctypes as C
    sub = free_batch.select(np.arange(256)).astype(np.float32)
    p0 = abi.phase0_params()
    s = solver.DdpSolver(256, N, sub.p_max, np.float32)
    want = s.solve(p0, sub)
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(getattr(sub, k)).to(dev) for k in ("n_seg", "x0", "xd", "T0", "n_planes", "planes")}
    cin = abi.BatchIn()
    cin.batch, cin.n_seg_max, cin.p_max, cin.mem = 256, N, sub.p_max, abi.MEM_DEVICE
    for k, v in t.items():
        setattr(cin, k, v.data_ptr())
    o = dict(rtn=torch.zeros(256, dtype=torch.int32, device=dev), cost=torch.zeros(256, device=dev),
             bez=torch.zeros(256, N, 18, device=dev), T=torch.zeros(256, N, device=dev),
             fwd_passes=torch.zeros(256, dtype=torch.int32, device=dev))
    cout = abi.BatchOut()
    cout.mem = abi.MEM_DEVICE
    for k, v in o.items():
        setattr(cout, k, v.data_ptr())
    s.set_stream(torch.cuda.current_stream().cuda_stream)
    s.solve_device(p0, cin, cout)
    torch.cuda.synchronize()
    assert np.array_equal(o["rtn"].cpu().numpy(), want.rtn)
    assert np.array_equal(o["bez"].cpu().numpy(), want.bez)
    assert np.array_equal(o["T"].cpu().numpy(), want.T)
    s.close()


def test_ticket_scheduler_is_bitwise_equal_to_one_workgroup_per_trajectory(built, corridor_batch, monkeypatch):
    """k_iterate_dyn (persistent waves drawing (trajectory, iteration) tickets; the default when the batch
    exceeds the resident waves) against k_iterate (one workgroup per trajectory): a trajectory's
    iterations run on different CUs / XCDs, the arithmetic and every discrete decision must not change."""
    res = {}
    for mode in ("static", "dynamic"):
        monkeypatch.setenv("DIRECT_DDP_SCHED", mode)
        s = solver.DdpSolver(B, N, corridor_batch.p_max, np.float32)
        g0, g1 = s.plan(abi.phase0_params(), abi.phase1_params(iter_max=30), corridor_batch.astype(np.float32))
        res[mode] = (g0, g1)
        s.close()
    for a, b in zip(res["static"], res["dynamic"]):
        for f in ("rtn", "iter_used", "fwd_passes", "infeas_out", "cost", "costq", "opterr", "mu", "T", "poly", "bez"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert res["dynamic"][1].iter_used.max() > 1


def test_paired_line_search_trials_are_bitwise_equal_to_the_sequential_search(built, corridor_batch, monkeypatch):
    """From the second attempt on two step sizes share a forward sweep (run_round<2>): each trial's arithmetic and the
    order of acceptance are those of the sequential search, so every output must be bit-identical - natural exits,
    both phases, 4096 polyhedron corridors (mean 2.4 trials per iteration, up to 11)."""
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("DIRECT_DDP_PAIR", mode)
        s = solver.DdpSolver(B, N, corridor_batch.p_max, np.float32)
        res[mode] = s.plan(abi.phase0_params(), abi.phase1_params(iter_max=40), corridor_batch.astype(np.float32))
        s.close()
    for a, b in zip(res["0"], res["1"]):
        for f in ("rtn", "iter_used", "fwd_passes", "infeas_out", "cost", "costq", "opterr", "mu", "T", "poly", "bez"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert res["1"][1].iter_used.max() > 5


def test_shared_line_search_is_bitwise_equal_to_the_owner_only_search(built, corridor_batch, free_batch, monkeypatch):
    """Waves that wait for a trajectory's chunk run later rounds of its line search into their own trial buffers and
    report TrialRes records (ddp_wave.h HelpSlot): which wave evaluated a step must not show anywhere - natural exits
    and the fixed-20 bench launch, every output bit-identical, no scheduler error - and the feature must really have
    been exercised (the launch is not slower than the owner-only one by more than noise is a bench matter, not tested)."""
    res, fixed = {}, {}
    for mode in ("0", "1"):
        monkeypatch.setenv("DIRECT_DDP_HELP", mode)
        s = solver.DdpSolver(B, N, corridor_batch.p_max, np.float32)
        res[mode] = s.plan(abi.phase0_params(), abi.phase1_params(iter_max=40), corridor_batch.astype(np.float32))
        assert s.sched_error() == 0
        s.close()
        fb = free_batch.astype(np.float32)
        s = solver.DdpSolver(B, N, fb.p_max, np.float32)
        g0 = s.solve(abi.phase0_params(), fb)
        b1 = fb.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, fb.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
        for _ in range(3):  # the interleaving differs from launch to launch
            g1 = s.solve(abi.phase1_params(iter_max=20, fixed_iters=1), b1)
            assert s.sched_error() == 0
            fixed.setdefault(mode, []).append(g1)
        s.close()
    fields = ("rtn", "iter_used", "fwd_passes", "infeas_out", "cost", "costq", "opterr", "mu", "T", "poly", "bez")
    for a, b in zip(res["0"], res["1"]):
        for f in fields:
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
    for g in fixed["0"][1:] + fixed["1"]:
        for f in fields:
            assert np.array_equal(getattr(fixed["0"][0], f), getattr(g, f)), f
    assert int(fixed["1"][0].fwd_passes.sum()) == B * 20


@pytest.mark.parametrize("dt,nb,kind", [(np.float32, B, "free"), (np.float32, B, "corridor"), (np.float64, 600, "corridor"),
                                        (np.float32, 1, "free"), (np.float64, 40, "free")])
def test_shared_backward_sweep_is_bitwise_equal_to_the_owner_only_sweep(built, dt, nb, kind, monkeypatch):
    """Waves that wait for a trajectory's next ticket compute the value-independent half of knots of its backward sweep
    (phases T2, rows, S, S2, the constraint half of H) and hand it over through records in HBM (ddp_wave.h, bwd_knot
    MODE 1 / 2, BwdShare): who computed which knot must not show anywhere.  DIRECT_DDP_BSHARE = 0 (every sweep with its
    owner), 1 (helpers), 2 (the forced split: every knot below the owner's first claim goes through a record, no helper
    needed): natural exits of both phases and the fixed-20 launch, every output bit-identical, no scheduler error, and the
    hand-over path really taken."""
    batch = problems.make_batch(kind, nb, N, seed=1000 if nb == B else 500 + nb).astype(dt)
    fields = ("rtn", "iter_used", "fwd_passes", "infeas_out", "cost", "costq", "opterr", "mu", "T", "poly", "bez")
    res, fixed, knots = {}, {}, {}
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("DIRECT_DDP_BSHARE", mode)
        s = solver.DdpSolver(nb, N, batch.p_max, dt)
        res[mode] = s.plan(abi.phase0_params(), abi.phase1_params(iter_max=30), batch)
        assert s.sched_error() == 0
        g0 = res[mode][0]
        b1 = batch.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, batch.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
        for _ in range(2):  # the interleaving differs from launch to launch
            fixed.setdefault(mode, []).append(s.solve(abi.phase1_params(iter_max=20, fixed_iters=1), b1))
            assert s.sched_error() == 0
        li = s.launch_info()
        knots[mode] = (li["shared_sweep"], li["helper_front_knots"], li["bwd_knot_visits"])
        s.close()
    for mode in ("1", "2"):
        for a, b in zip(res["0"], res[mode]):
            for f in fields:
                assert np.array_equal(getattr(a, f), getattr(b, f)), (mode, f)
        for g in fixed[mode] + fixed["0"][1:]:
            for f in fields:
                assert np.array_equal(getattr(fixed["0"][0], f), getattr(g, f)), (mode, f)
    assert knots["0"][:2] == (0, 0), knots
    assert knots["2"][0] == 2 and knots["2"][1] > 0.9 * knots["2"][2], knots   # every knot below the owner's first claim (the top two)
    assert knots["1"][0] == 1, knots
    if nb == 1:   # helpers exist: the idle waves next to a lone trajectory
        assert knots["1"][1] > 0, knots


@pytest.mark.parametrize("dt,nb,kind", [(np.float32, B, "free"), (np.float64, 1500, "corridor"), (np.float32, 9000, "corridor")])
def test_pipelined_batches_on_two_yielding_handles_are_bit_identical_to_serial_launches(built, dt, nb, kind):
    """DIRECT_FLAG_YIELD: two handles with a stream each, device-memory calls alternating between them - the waves a launch
    no longer needs leave the hot kernel early so that the other handle's launch fills the CUs (include/direct_ddp.h).  Which
    wave runs which ticket, and next to which other launch, must not show: natural exits and the fixed-20 launch of six
    pipelined batches against one serial launch of an ordinary handle, every output bit for bit, no scheduler error."""
    import torch
    from direct_amd import devmem
    dev = torch.device("cuda:0")
    batch = problems.make_batch(kind, nb, N, seed=700 + nb).astype(dt)
    one = solver.DdpSolver(nb, N, batch.p_max, dt)
    g0 = one.solve(abi.phase0_params(), batch)
    b1 = batch.phase1_inputs(g0)
    din = devmem.DeviceBatch(b1, dev)
    ref = devmem.DeviceResult(nb, N, dt, dev)
    hs = [solver.DdpSolver(nb, N, batch.p_max, dt, flags=abi.FLAG_YIELD) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    for h, st in zip(hs, streams):
        h.set_stream(st.cuda_stream)
    outs = [devmem.DeviceResult(nb, N, dt, dev) for _ in range(6)]
    for p in (abi.phase1_params(iter_max=20, fixed_iters=1), abi.phase1_params(iter_max=40)):
        one.solve_device(p, din.cin, ref.cout)
        torch.cuda.synchronize()
        for i in range(6):
            hs[i % 2].solve_device(p, din.cin, outs[i].cout)
        torch.cuda.synchronize()
        assert one.sched_error() == 0 and all(h.sched_error() == 0 for h in hs)
        for i in range(6):
            for k in ref.t:
                assert torch.equal(outs[i].t[k], ref.t[k]), (i, k, p.fixed_iters)
        assert int(ref.t["fwd_passes"].sum().item()) > 10 * nb
    for h in hs + [one]:
        h.close()


def test_a_lone_trajectory_with_hundreds_of_waiters_on_its_shared_sweep(built, monkeypatch):
    """B = 1, N = 100, 320 fixed iterations: the ticket scheduler hands out n_epochs + tail tickets for the ONE trajectory and
    every holder polls its open sweep - far more than the 255 an 8-bit helper count could hold (the carry went
    into the sweep's tag, the owner's flag wait ran into its spin limit and every row came back -101).  The count field now
    holds every resident wave and kBsMaxHelpers are let in: no scheduler error, helpers really used, every output bit
    identical to the owner-only launch."""
    batch = problems.make_batch("free", 1, N, seed=4242).astype(np.float32)
    p = abi.phase1_params(iter_max=320, fixed_iters=1)
    fields = ("rtn", "iter_used", "fwd_passes", "cost", "costq", "opterr", "mu", "T", "poly", "bez")
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("DIRECT_DDP_BSHARE", mode)
        s = solver.DdpSolver(1, N, batch.p_max, np.float32)
        g0 = s.solve(abi.phase0_params(), batch)
        b1 = batch.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, batch.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
        res[mode] = s.solve(p, b1)
        assert s.sched_error() == 0
        li = s.launch_info()
        if mode == "1":
            assert li["shared_sweep"] == 1 and li["helper_front_knots"] > 0, li
        s.close()
    assert res["1"].rtn[0] != -101 and res["1"].fwd_passes[0] == 320
    for f in fields:
        assert np.array_equal(getattr(res["0"], f), getattr(res["1"], f)), f


@pytest.mark.parametrize("dt,nb,nseg,chunk", [(np.float64, 3300, 40, "1"), (np.float32, 3500, 60, "3")])
def test_shared_line_search_other_storage_types_and_chunk_sizes(built, dt, nb, nseg, chunk, monkeypatch):
    """The same bitwise equality for double storage, an awkward batch size, shorter trajectories and tickets of three
    outer-loop trips (a helper then holds the ticket of a chunk while the chunk before it runs three line searches)."""
    batch = problems.make_batch("corridor", nb, nseg, seed=77)
    monkeypatch.setenv("DIRECT_DDP_CHUNK", chunk)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("DIRECT_DDP_HELP", mode)
        s = solver.DdpSolver(nb, nseg, batch.p_max, dt)
        res[mode] = s.plan(abi.phase0_params(), abi.phase1_params(iter_max=30), batch.astype(dt))
        assert s.sched_error() == 0
        s.close()
    for a, b in zip(res["0"], res["1"]):
        for f in ("rtn", "iter_used", "fwd_passes", "infeas_out", "cost", "costq", "opterr", "mu", "T", "poly", "bez"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert res["1"][1].iter_used.max() > 5


@pytest.mark.parametrize("dt,nb,kind,chunk", [(np.float32, B, "free", "1"), (np.float64, 3300, "corridor", "1"), (np.float32, 5000, "corridor", "3"),
                                              (np.float32, 2, "free", "1")])
def test_help_only_tickets_behind_the_last_epoch_change_no_bit(built, dt, nb, kind, chunk, monkeypatch):
    """A fixed-length launch hands out DIRECT_DDP_TAIL more rounds of tickets behind its last epoch that never run a
    chunk: their holders wait for the trajectory's last chunk and join its open line searches meanwhile (direct_ddp.hip,
    next_work), instead of leaving the kernel while the slowest chains still search.  0, 1 and the default 8 rounds: the
    fixed-20 launch (twice each: the interleaving differs) and the natural exits are bit-identical, the error flag stays clear
    (a help-only ticket that waited for an epoch that never comes would raise it)."""
    batch = problems.make_batch(kind, nb, 60 if chunk == "3" else N, seed=900 + nb).astype(dt)
    nseg = int(batch.n_seg.max())
    fields = ("rtn", "iter_used", "fwd_passes", "infeas_out", "cost", "costq", "opterr", "mu", "T", "poly", "bez")
    monkeypatch.setenv("DIRECT_DDP_CHUNK", chunk)
    fixed, nat = {}, {}
    for mode in ("0", "1", None):
        if mode is None:
            monkeypatch.delenv("DIRECT_DDP_TAIL", raising=False)
        else:
            monkeypatch.setenv("DIRECT_DDP_TAIL", mode)
        s = solver.DdpSolver(nb, nseg, batch.p_max, dt)
        nat[mode] = s.plan(abi.phase0_params(), abi.phase1_params(iter_max=30), batch)
        assert s.sched_error() == 0
        g0 = nat[mode][0]
        b1 = batch.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, batch.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
        for _ in range(2):
            fixed.setdefault(mode, []).append(s.solve(abi.phase1_params(iter_max=20, fixed_iters=1), b1))
            assert s.sched_error() == 0
        s.close()
    for mode in ("1", None):
        for a, b in zip(nat["0"], nat[mode]):
            for f in fields:
                assert np.array_equal(getattr(a, f), getattr(b, f)), (mode, f)
        for g in fixed[mode] + fixed["0"][1:]:
            for f in fields:
                assert np.array_equal(getattr(fixed["0"][0], f), getattr(g, f)), (mode, f)
    assert fixed["0"][0].iter_used.max() == 20


def test_help_only_tickets_with_row_slot_classes(built, monkeypatch):
    """The same with DIRECT_DDP_CLASSES=1 on a batch of two row-slot classes (every other corridor padded to ~40 planes per
    polytope): each class is a launch of its own over an index list (Batch::idx), side by side on forked streams, with its
    own ticket counter - help-only tickets index trajectories through that list as well."""
    base = problems.make_batch("corridor", 400, 30, seed=321)
    wide = helpers.with_extra_planes(base, 40, seed=5)
    n_planes = np.where((np.arange(400) % 2 == 0)[:, None], wide.n_planes, base.n_planes)
    batch = abi.HostBatch(base.n_seg, base.x0, base.xd, base.T0, n_planes, wide.planes, seeds=base.seeds).astype(np.float32)
    fields = ("rtn", "iter_used", "fwd_passes", "infeas_out", "cost", "costq", "opterr", "mu", "T", "poly", "bez")
    monkeypatch.setenv("DIRECT_DDP_CLASSES", "1")
    fixed, nat = {}, {}
    for mode in ("0", "8"):
        monkeypatch.setenv("DIRECT_DDP_TAIL", mode)
        s = solver.DdpSolver(400, 30, 40, np.float32)
        nat[mode] = s.plan(abi.phase0_params(), abi.phase1_params(iter_max=30), batch)
        assert s.sched_error() == 0
        g0 = nat[mode][0]
        b1 = batch.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, batch.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
        for _ in range(2):
            fixed.setdefault(mode, []).append(s.solve(abi.phase1_params(iter_max=20, fixed_iters=1), b1))
            assert s.sched_error() == 0
        s.close()
    for a, b in zip(nat["0"], nat["8"]):
        for f in fields:
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
    for g in fixed["8"] + fixed["0"][1:]:
        for f in fields:
            assert np.array_equal(getattr(fixed["0"][0], f), getattr(g, f)), f


@pytest.mark.parametrize("nb,kind,dt", [(1, "corridor", np.float32), (7, "free", np.float64), (300, "corridor", np.float32)])
def test_small_batches_on_the_ticket_scheduler_match_the_static_launch(built, nb, kind, dt, monkeypatch):
    """Below the resident waves the launch used to be one workgroup per trajectory; with the shared line search the
    ticket scheduler serves these batches too and the waves left over help (single steps, one wave each, when the batch
    is small).  Same bits as the static launch, natural exits, both phases."""
    batch = problems.make_batch(kind, nb, N, seed=300 + nb)
    res = {}
    for mode in ("static", "dynamic"):
        monkeypatch.setenv("DIRECT_DDP_SCHED", mode)
        s = solver.DdpSolver(nb, N, batch.p_max, dt)
        res[mode] = s.plan(abi.phase0_params(), abi.phase1_params(iter_max=40), batch.astype(dt))
        assert s.sched_error() == 0
        s.close()
    for a, b in zip(res["static"], res["dynamic"]):
        for f in ("rtn", "iter_used", "fwd_passes", "infeas_out", "cost", "costq", "opterr", "mu", "T", "poly", "bez"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f


def test_output_sampling_of_the_full_batch(built, free_batch):
    """direct_traj_sample_batch on 4096 solved trajectories: size-independent properties of the samples
    (the oracle is checked on a random subset)."""
    s = solver.DdpSolver(B, N, free_batch.p_max, np.float32)
    g0, g1 = s.plan(abi.phase0_params(), abi.phase1_params(), free_batch.astype(np.float32))
    cap, dt = 4096, 0.2
    d = s.sample(free_batch.n_seg, g1.bez, g1.T, dt, cap)
    s.close()
    cnt = d["count"]
    assert (cnt > N).all() and (cnt <= cap).all()
    assert np.abs(cnt - g1.T.astype(np.float64).sum(1) / dt).max() <= N + 1      # one partial step per segment
    # every trajectory starts at its start position and its first sample of segment k+1 continues segment k
    assert np.abs(d["pos"][:, 0] - free_batch.x0[:, :3]).max() < 1e-3
    sf = d["seg_first"]
    assert (np.diff(sf, axis=1) >= 1).all() and (sf[:, 0] == 0).all()
    # samples are dt apart in time: consecutive points are at most vmax * dt (+ rounding) apart
    idx = np.arange(cap)[None, :]
    step = np.linalg.norm(np.diff(d["pos"].astype(np.float64), axis=1), axis=2)
    valid = idx[:, 1:] < cnt[:, None]
    assert (step[valid] <= np.sqrt(3.0) * (d["vmax"].astype(np.float64)[:, None] * dt + 1e-2).repeat(cap - 1, 1)[valid]).all()
    # the polyline is at least as long as the start-goal distance minus the last partial step
    chord = np.linalg.norm(free_batch.xd[:, :3] - free_batch.x0[:, :3], axis=1)
    assert (d["length"] > 0.9 * chord).all() and np.allclose(d["length"], np.where(valid, step, 0).sum(1), rtol=1e-4)
    # converged trajectories respect the velocity / acceleration bounds at the samples
    ok = g1.rtn >= 0
    assert (d["vmax"][ok] < 2.0 + 1e-2).all() and (d["amax"][ok] < 2.0 + 1e-2).all()
    sub = np.random.default_rng(5).choice(B, 24, replace=False)
    o = refapi.sample_batch(free_batch.n_seg[sub], g1.bez[sub], g1.T[sub], dt, cap)
    assert (o["count"] == cnt[sub]).all()
    assert helpers.rel(d["pos"][sub], o["pos"]) < 2e-6 and helpers.rel(d["length"][sub], o["length"]) < 2e-6


@pytest.mark.parametrize("nb", [3073, 5000])
def test_ticket_scheduler_at_awkward_batch_sizes(built, nb, monkeypatch):
    """Batches just above the resident-wave count (3072 on MI355X) and far from a multiple of it, short
    trajectories with very different iteration counts: static and ticket-scheduled launches agree bit for bit."""
    batch = problems.make_batch("corridor", nb, 6, seed=77)
    res = {}
    for mode in ("static", "dynamic"):
        monkeypatch.setenv("DIRECT_DDP_SCHED", mode)
        s = solver.DdpSolver(nb, 6, batch.p_max, np.float64)
        res[mode] = s.plan(abi.phase0_params(), abi.phase1_params(), batch)
        s.close()
    for a, b in zip(res["static"], res["dynamic"]):
        for f in ("rtn", "iter_used", "fwd_passes", "cost", "T", "bez"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
    it = res["dynamic"][1].iter_used
    assert it.min() < it.max()
