"""BASELINE configs 4 and 5 on the device, and the boundary behaviours that only exist with device-resident
arrays: on-device validation of n_seg / n_planes, the sticky scheduler-error flag and the config-5 gather
through the library's C entry points (direct_ddp_gather_best over RCCL, world size 1 on this box).

Config 4: B = 16384, N = 300, double storage, config-3 generator.
Config 5: 131072 corridors over 8 GPUs = 16384 per GPU, N = 100, float storage; one shard through the device-resident
interface and the C gather, then ALL EIGHT shards one after another on this one device.
The oracle cannot solve these batches in seconds: parity is exact agreement on a sample plus the
size-independent properties of tests/test_gpu_fullsize.py on the whole batch."""
import os

import numpy as np
import pytest

from direct_amd import abi, devmem, distributed, problems, solver
from oracle import refapi
from tests import helpers, n100_lib
from tests.test_gpu_fullsize import check_properties

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torch():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    return torch


def test_config4_long_horizon_fp64(built, monkeypatch):
    B, N = 16384, 300
    batch = problems.make_batch("corridor", B, N, seed=1000)
    p0, p1 = abi.phase0_params(), abi.phase1_params()
    s = solver.DdpSolver(B, N, batch.p_max, np.float64)
    g0, g1 = s.plan(p0, p1, batch)
    assert s.sched_error() == 0
    check_properties(batch, g1, 1e-9)
    assert set(np.unique(g1.rtn)) <= {0, 1, -3, -4}
    # agreement with the oracle on 16 problems spread over the batch (both phases, every discrete decision), next to
    # the oracle against ITSELF with its inputs moved by one ulp (N = 300: three times the knots to amplify a bit)
    idx = np.arange(5, B, B // 16)[:16]
    sb = batch.select(idx)
    r0, r1 = refapi.plan_batch(p0, p1, sb)
    from tests import soak_lib
    same0 = (g0.rtn[idx] == r0.rtn) & (g0.iter_used[idx] == r0.iter_used)
    same1 = (g1.rtn[idx] == r1.rtn) & (g1.iter_used[idx] == r1.iter_used)
    # four controls; the device must reproduce at least as many outcomes as the WORST of them (no allowance, no floor)
    ctl0, ctl1, ctl_cost, ctl_T = [], [], 0.0, 0.0
    for cs in (4, 5, 6, 7):
        c0, c1 = refapi.plan_batch(p0, p1, soak_lib.perturb_ulp(sb, cs))
        ctl0.append((c0.rtn == r0.rtn) & (c0.iter_used == r0.iter_used))
        ctl1.append((c1.rtn == r1.rtn) & (c1.iter_used == r1.iter_used))
        okc = ctl0[-1] & ctl1[-1] & (r1.rtn >= 0)
        if okc.any():
            ctl_cost = max(ctl_cost, float(np.abs(c1.cost[okc] / r1.cost[okc] - 1).max()))
            ctl_T = max(ctl_T, helpers.rel(c1.T[okc], r1.T[okc]))
    assert same0.sum() >= min(c.sum() for c in ctl0), (same0, ctl0)
    assert same1.sum() >= min(c.sum() for c in ctl1), (same1, ctl1)
    ok = same0 & same1 & (r1.rtn >= 0)
    assert ok.sum() >= min((a & b & (r1.rtn >= 0)).sum() for a, b in zip(ctl0, ctl1))
    # the fused plan starts phase 1 from the device's phase-0 result (1e-13 from the oracle's): 300 knots amplify that like
    # any input perturbation - bounded by the controls' own deviation on the problems whose decisions they reproduce
    assert np.abs(g1.cost[idx][ok] / r1.cost[ok] - 1).max() < max(1e-8, 30 * ctl_cost), (np.abs(g1.cost[idx][ok] / r1.cost[ok] - 1).max(), ctl_cost)
    assert helpers.rel(g1.T[idx][ok], r1.T[ok]) < max(1e-6, 30 * ctl_T), (helpers.rel(g1.T[idx][ok], r1.T[ok]), ctl_T)
    s.close()
    # the ticket scheduler at N = 300 (3x longer chunks against the spin limit): bitwise equal to the static launch
    sub = batch.select(np.arange(4096))
    res = {}
    for mode in ("static", "dynamic"):
        monkeypatch.setenv("DIRECT_DDP_SCHED", mode)
        s2 = solver.DdpSolver(4096, N, sub.p_max, np.float64)
        res[mode] = s2.plan(p0, abi.phase1_params(iter_max=12), sub)
        assert s2.sched_error() == 0
        s2.close()
    for a, b in zip(res["static"], res["dynamic"]):
        for f in ("rtn", "iter_used", "fwd_passes", "cost", "T", "bez"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
    # a slice of the big batch solved alone is bit-identical to its rows in the big batch
    assert np.array_equal(res["dynamic"][0].bez, g0.bez[:4096]) and np.array_equal(res["dynamic"][0].rtn, g0.rtn[:4096])


def test_config4_shape_in_feasible_mode_runs_full_length_rollouts(built, monkeypatch):
    """Config 4's shape (N = 300, double storage, B = 16384) on the free-space generator: phase 0 leaves every trajectory
    feasible, so phase 1 runs FULL-LENGTH forward rollouts - 300 knots per trial, the part of the long-horizon path that
    config 4's own corridors never reach (they stay in infeasible mode, where every trial dies at the fraction-to-boundary
    rule within a few knots: 0.04 forward knots per backward knot, ddp_optimizer.cpp:684-688, 760-763).  The fixed-20
    launch of `bench.py --config 6`: at least one forward trial-knot per backward knot and most line searches accepted;
    16 problems against the oracle with its own one-ulp control; ticket scheduler == static launch and a slice == its rows
    in the big batch, bit for bit."""
    from tests import soak_lib
    B, N = 16384, 300
    batch = problems.make_batch("free", B, N, seed=1000)
    p0, p1 = abi.phase0_params(), abi.phase1_params()
    s = solver.DdpSolver(B, N, batch.p_max, np.float64)
    g0 = s.solve(p0, batch)
    assert (g0.rtn == 2).mean() > 0.99 and g0.infeas_out.mean() < 0.01        # feasible mode from here on
    b1 = batch.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, batch.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
    gf = s.solve(abi.phase1_params(iter_max=20, fixed_iters=1), b1)          # the timed launch of bench.py --config 6
    li = s.launch_info()
    assert s.sched_error() == 0 and int(gf.fwd_passes.sum()) == 20 * B
    assert li["fwd_knot_visits"] >= li["bwd_knot_visits"], li                   # >= 1 forward trial-knot per backward knot
    assert li["accepted_line_searches"] > 0.8 * 20 * B, li                      # the iterate really moves
    g1 = s.solve(p1, b1)                                                        # natural exits
    assert s.sched_error() == 0
    check_properties(batch, g1, 1e-9)
    s.close()
    # 16 problems spread over the batch against the oracle (phase 1 from the DEVICE's phase-0 result: identical inputs),
    # next to the oracle against itself with its inputs moved by one ulp
    idx = np.arange(7, B, B // 16)[:16]
    sb = b1.select(idx)
    r1, _ = refapi.solve_batch(p1, sb)
    same = (g1.rtn[idx] == r1.rtn) & (g1.iter_used[idx] == r1.iter_used)
    ctls, ctl_dev = [], 0.0
    for cs in (9, 10, 11, 12):   # four controls: the device reproduces at least as many outcomes as the worst of them
        c1, _ = refapi.solve_batch(p1, soak_lib.perturb_ulp(sb, cs))
        ctls.append((c1.rtn == r1.rtn) & (c1.iter_used == r1.iter_used))
        okc = ctls[-1] & (r1.rtn >= 0)
        with np.errstate(divide="ignore", invalid="ignore"):
            if okc.any():
                ctl_dev = max(ctl_dev, float(np.abs(c1.cost[okc] / r1.cost[okc] - 1).max()))
    assert same.sum() >= min(c.sum() for c in ctls), (same, ctls)
    ok = same & (r1.rtn >= 0)
    assert ok.sum() >= min((c & (r1.rtn >= 0)).sum() for c in ctls)
    assert np.abs(g1.cost[idx][ok] / r1.cost[ok] - 1).max() <= max(1e-8, 10 * ctl_dev)
    assert helpers.rel(g1.T[idx][ok], r1.T[ok]) < 1e-6
    # scheduling is invisible: ticket scheduler == one workgroup per trajectory, and a slice alone == its rows in the batch
    sub = b1.select(np.arange(4096))
    res = {}
    for mode in ("static", "dynamic"):
        monkeypatch.setenv("DIRECT_DDP_SCHED", mode)
        s2 = solver.DdpSolver(4096, N, sub.p_max, np.float64)
        res[mode] = s2.solve(abi.phase1_params(iter_max=12), sub)
        assert s2.sched_error() == 0
        s2.close()
    for f in ("rtn", "iter_used", "fwd_passes", "cost", "T", "bez"):
        assert np.array_equal(getattr(res["static"], f), getattr(res["dynamic"], f)), f
    monkeypatch.delenv("DIRECT_DDP_SCHED")
    s3 = solver.DdpSolver(B, N, batch.p_max, np.float64)
    big = s3.solve(abi.phase1_params(iter_max=12), b1)
    s3.close()
    assert np.array_equal(res["dynamic"].bez, big.bez[:4096]) and np.array_equal(res["dynamic"].iter_used, big.iter_used[:4096])


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_device_side_time_allocation(built, dt):
    """T0 == NULL: initTimeAllocation (teach_repeat_planner.cpp:583-639) runs on the device in front of the setup kernel,
    from the start / goal positions and the polytope seeds - for host arrays and for device-resident ones (where the
    seeds -> durations -> plan chain then never visits the host).  The durations it produces are those of the host twin
    direct_time_allocation (double arithmetic, 1e-14; rounded once more for float storage), and a plan from them is
    bit-identical to the plan from the same durations provided by the caller."""
    torch = _torch()
    import ctypes as C
    B, N = 300, 24
    batch = problems.make_batch("corridor", B, N, seed=31).astype(dt)
    p0, p1 = abi.phase0_params(), abi.phase1_params(iter_max=25)
    # the generator's durations ARE a time allocation of its seeds: recompute them with the C twin on what the device sees
    lib = solver.lib()
    lib.direct_time_allocation.argtypes = [C.c_int32, C.c_int32] + [C.c_void_p] * 4 + [C.c_double, C.c_double, C.c_void_p]
    T_host = np.zeros((B, N))
    st, gl, sd = (np.ascontiguousarray(a, np.float64) for a in (batch.x0[:, :3], batch.xd[:, :3], batch.seeds))
    assert lib.direct_time_allocation(B, N, batch.n_seg.ctypes.data, st.ctypes.data, gl.ctypes.data, sd.ctypes.data,
                                      p0.max_vel, p0.max_acc, T_host.ctypes.data) == 0
    ref = abi.HostBatch(batch.n_seg, batch.x0, batch.xd, T_host, batch.n_planes, batch.planes, seeds=batch.seeds, dtype=dt)
    s = solver.DdpSolver(B, N, batch.p_max, dt)
    # (a) host arrays without durations: what the device allocated (the setup kernel stores it as u[9] of every knot)
    s.begin(p0, ref.without_T0())
    T_dev = s.get(abi.FIELD_U)[:, :, 9].astype(np.float64)
    live = np.arange(N)[None, :] < batch.n_seg[:, None]
    assert (T_dev[~live] == 0).all() and (T_dev[live] > 0).all()
    # the host twin's values: to 1e-14 in double (the device's square root is not always the correctly rounded one:
    # 4 % of the entries differ by one ulp), to a float ulp with float storage
    assert np.abs(T_dev / np.where(live, T_host, 1.0) - 1.0)[live].max() <= (1e-14 if dt == np.float64 else 1.2e-7)
    ref = abi.HostBatch(batch.n_seg, batch.x0, batch.xd, T_dev, batch.n_planes, batch.planes, seeds=batch.seeds, dtype=dt)
    want = s.plan(p0, p1, ref)
    got = s.plan(p0, p1, ref.without_T0())
    for a, b in zip(want, got):
        for f in ("rtn", "iter_used", "cost", "T", "bez", "poly"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
    # (b) device-resident arrays without durations
    dev = torch.device("cuda:0")
    s.set_stream(torch.cuda.current_stream().cuda_stream)
    din = devmem.DeviceBatch(ref.without_T0(), dev)
    assert "T0" not in din.tens
    o0, o1 = devmem.DeviceResult(B, N, dt, dev), devmem.DeviceResult(B, N, dt, dev)
    s.plan_device(p0, p1, din.cin, o0.cout, o1.cout)
    torch.cuda.synchronize()
    g1 = o1.to_host()
    for f in ("rtn", "iter_used", "cost", "T", "bez", "poly"):
        assert np.array_equal(getattr(want[1], f), getattr(g1, f)), f
    # no durations AND no seeds: refused, nothing launched
    noseed = abi.HostBatch(batch.n_seg, batch.x0, batch.xd, T_host, batch.n_planes, batch.planes, dtype=dt).without_T0()
    with pytest.raises(solver.DirectError):
        s.solve(p0, noseed)
    s.close()


def test_bench_distributed_branches_run_at_world_size_one(built):
    """DIRECT_BENCH_FORCE_DIST=1: bench.py's N > 1 path - the RCCL process group, the barrier and the all-reduces around the
    timed region, the all-gather of the config-5 reduction, the object broadcast of the unique id and the library's own
    RCCL communicator - executes at world size 1, so that the first multi-GPU run is not the first execution of any of
    its lines (VERDICT r04 missing #2)."""
    import json
    import subprocess
    import sys
    _torch()
    env = dict(os.environ, DIRECT_BENCH_FORCE_DIST="1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "512", "--steps", "2", "--warmup", "1",
                        "--no-secondary", "--no-cpu-baseline", "--no-live-traffic"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    g = line["gather"]
    assert g["forced_dist_path"] is True and g["dist_world_size"] == 1
    assert g.get("rccl_ranks") == 1 and g.get("c_abi_matches_torch") is True, g
    assert line["n_gpus"] == 1 and line["value"] > 0 and len(line["kernel_ms_per_rank"]) == 1


def test_config5_shard_fp32_and_c_abi_gather(built):
    torch = _torch()
    B, N, rank = 16384, 100, 3           # the shard GPU 3 of 8 would own
    first = rank * B
    batch = problems.make_batch("corridor", B, N, seed=1000, first=first, dtype=np.float32)
    p0, p1 = abi.phase0_params(), abi.phase1_params()
    s = solver.DdpSolver(B, N, batch.p_max, np.float32)
    dev = torch.device("cuda:0")
    s.set_stream(torch.cuda.current_stream().cuda_stream)
    din = devmem.DeviceBatch(batch, dev)
    o0, o1 = devmem.DeviceResult(B, N, np.float32, dev), devmem.DeviceResult(B, N, np.float32, dev)
    s.plan_device(p0, p1, din.cin, o0.cout, o1.cout)
    torch.cuda.synchronize()
    assert s.sched_error() == 0
    g1 = o1.to_host()
    check_properties(batch, g1, 5e-4)
    # device-resident plan == host-interface plan, bit for bit
    h0, h1 = s.plan(p0, p1, batch)
    for f in ("rtn", "iter_used", "cost", "T", "bez", "poly"):
        assert np.array_equal(getattr(h1, f), getattr(g1, f)), f
    # the shard is position independent: the same rows generated as part of a larger stream
    other = problems.make_batch("corridor", 512, N, seed=1000, first=first + 1024, dtype=np.float32)
    assert np.array_equal(other.planes, batch.planes[1024:1536])
    # config-5 reduction through the C entry points: RCCL communicator of this one rank
    li, lc = distributed.local_best(g1.cost, g1.rtn)
    comm = s.rccl_comm_create(s.rccl_unique_id(), 1, 0)
    gi, gc, owner, gb, gT = s.gather_best(comm, 1, 0, g1.cost, g1.rtn, g1.bez, g1.T, first)
    assert (gi, owner) == (first + li, 0) and gc == lc
    assert np.array_equal(gb, g1.bez[li]) and np.array_equal(gT, g1.T[li])
    wb, wT = torch.zeros(N, 18, device=dev), torch.zeros(N, device=dev)
    di, dc, down = s.gather_best(comm, 1, 0, o1["cost"].data_ptr(), o1["rtn"].data_ptr(), o1["bez"].data_ptr(),
                                 o1["T"].data_ptr(), first, mem=abi.MEM_DEVICE, batch=B, out_bez=wb.data_ptr(), out_T=wT.data_ptr())
    assert (di, dc, down) == (gi, gc, 0)
    assert np.array_equal(wb.cpu().numpy(), g1.bez[li]) and np.array_equal(wT.cpu().numpy(), g1.T[li])
    # no feasible trajectory anywhere: index -1, cost inf
    ni, nc_, nown, nb, nT = s.gather_best(comm, 1, 0, g1.cost, np.full(B, -4, np.int32), g1.bez, g1.T, first)
    assert ni == -1 and nown == -1 and np.isinf(nc_) and not nb.any()
    s.rccl_comm_destroy(comm)
    s.close()


def test_config5_whole_workload_on_one_gpu(built):
    """BASELINE config 5 in full: all eight 16384-corridor shards of the 131072-corridor stream, generated and solved one
    after another on this one device (what ranks 0 .. 7 of an 8-GPU node would each do with theirs), reduced with the
    tie rule of the collective.  (a) the size-independent properties on all 131072 results; (b) the winner is the argmin
    over ALL costs (ties: smaller global index), and direct_ddp_gather_best fed the eight local bests - the records the
    eight ranks would contribute - picks that same winner; (c) a sample of eight problems of every shard against the oracle, with the oracle's own one-ulp control."""
    B, N, G = 16384, 100, 8
    p0, p1 = abi.phase0_params(), abi.phase1_params()
    s = solver.DdpSolver(B, N, 12, np.float32)
    loc = []           # per shard: (local index, cost) of its best, its block
    all_cost, all_rtn = [], []
    n_same, n_checked, n_ctl_same, devs, cdevs = 0, 0, 0, [], []
    for g in range(G):
        first = g * B
        batch = problems.make_batch("corridor", B, N, seed=1000, first=first, dtype=np.float32)
        g0, g1 = s.plan(p0, p1, batch)
        assert s.sched_error() == 0
        check_properties(batch, g1, 5e-4)                       # (a)
        li, lc = distributed.local_best(g1.cost, g1.rtn)
        assert li >= 0
        loc.append((first + li, lc, g1.bez[li].copy(), g1.T[li].copy()))
        all_cost.append(g1.cost.astype(np.float64))
        all_rtn.append(g1.rtn.copy())
        # (c) eight problems of the shard against the oracle on the same float-rounded inputs, fused two-phase plan of
        # both (float storage: SURVEY.md 8(c)'s fp32 tolerances; the N = 100 samples with their controls: test_gpu_n100.py)
        idx = np.array([17, B // 8 + 3, B // 4 + 9, B // 2 + 5, 5 * B // 8 + 1, 3 * B // 4 + 7, 7 * B // 8 + 2, B - 3])
        sb = batch.select(idx).astype(np.float64)
        r0, r1 = refapi.plan_batch(p0, p1, sb)
        assert ((g0.rtn[idx] >= 0) == (r0.rtn >= 0)).all()
        same = (g1.rtn[idx] == r1.rtn) & (r1.rtn >= 0)
        n_same += int(same.sum())
        n_checked += len(idx)
        devs += list(np.abs(g1.cost[idx][same] / r1.cost[same] - 1))
        # the control: the oracle against itself, every input moved by -1 / 0 / +1 ulp of a float
        _, q1 = refapi.plan_batch(p0, p1, n100_lib.perturb_float_ulp(sb, 100 + g))
        cs = (q1.rtn == r1.rtn) & (r1.rtn >= 0)
        n_ctl_same += int(cs.sum())
        cdevs += list(np.abs(q1.cost[cs] / r1.cost[cs] - 1))
    devs, cdevs = np.array(devs), np.array(cdevs)
    # float storage at N = 100: SURVEY.md 8(c)'s 1e-3 for the bulk; the tail is the problems the 100 iterations do not
    # converge, and how far those may end apart is what the control measures on these very problems
    # same return code on as many of the 64 problems as the one-ulp control reproduces (minus two), the bulk within 1e-5
    assert n_checked == 8 * G and n_same >= n_ctl_same - 2 and np.median(devs) < 1e-5, (n_same, n_ctl_same, n_checked, np.sort(devs)[::-1][:5])
    assert (devs < 1e-3).mean() >= (cdevs < 1e-3).mean() - 0.1 and devs.max() <= max(1e-3, 5 * cdevs.max()), (np.sort(devs)[::-1][:5], np.sort(cdevs)[::-1][:5])
    cost, rtn = np.concatenate(all_cost), np.concatenate(all_rtn)
    wi, wc = distributed.local_best(cost, rtn)                  # argmin over all 131072, ties to the smaller index
    key = [(c, i) for i, c, _, _ in loc]
    wg = int(np.lexsort(([k[1] for k in key], [k[0] for k in key]))[0])   # the reduction the ranks perform on their records
    assert loc[wg][0] == wi and loc[wg][1] == wc
    # (b) the same reduction through the library's C entry point: the eight records as a batch of eight
    comm = s.rccl_comm_create(s.rccl_unique_id(), 1, 0)
    lc8 = np.array([c for _, c, _, _ in loc], np.float32)
    bez8 = np.stack([b for _, _, b, _ in loc]).astype(np.float32)
    T8 = np.stack([t for _, _, _, t in loc]).astype(np.float32)
    gi, gc, owner, gb, gT = s.gather_best(comm, 1, 0, lc8, np.zeros(G, np.int32), bez8, T8, 0)
    s.rccl_comm_destroy(comm)
    s.close()
    assert gi == wg and gc == float(np.float32(wc))
    assert np.array_equal(gb, loc[wg][2]) and np.array_equal(gT, loc[wg][3])


def _two_rank_worker(rank, world, port, total, n_seg, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.is_available()   # torch's HIP runtime first (see conftest.py)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from direct_amd import abi as abi_, distributed as distributed_, problems as problems_, solver as solver_
    first, count = distributed_.shard_range(total, rank, world)
    batch = problems_.make_batch("corridor", count, n_seg, seed=77, first=first)
    s = solver_.DdpSolver(count, n_seg, batch.p_max, np.float64, device=0)   # both ranks on the one device of this box
    g0, g1 = s.plan(abi_.phase0_params(), abi_.phase1_params(), batch)
    s.close()
    i, c = distributed_.local_best(g1.cost, g1.rtn)
    block = torch.from_numpy(np.concatenate([g1.bez[i].ravel(), g1.T[i].ravel()]))
    cost, gidx, owner, blk = distributed_.gather_best(c, first + i, block)
    q.put((rank, cost, gidx, owner, blk.numpy().copy(), first, count, g1.cost.copy(), g1.rtn.copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_solve_gather_through_the_product_library(built):
    """World size 2 through libdirect_ddp.so: two processes (both on this box's one device; RCCL refuses two ranks on
    one GPU, so the exchange runs over gloo) shard the stream, solve their shard on the device and run the config-5
    gather; both learn the winner a single process finds on the whole batch.  tests/test_dist_gloo.py does the same
    without a GPU (the oracle in the device's place)."""
    _torch()
    import torch.multiprocessing as mp
    total, n_seg, world = 2 * problems.CHUNK, 12, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, total, n_seg, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    full = problems.make_batch("corridor", total, n_seg, seed=77)
    s = solver.DdpSolver(total, n_seg, full.p_max, np.float64)
    g0, g1 = s.plan(abi.phase0_params(), abi.phase1_params(), full)
    s.close()
    i, c = distributed.local_best(g1.cost, g1.rtn)
    want = np.concatenate([g1.bez[i].ravel(), g1.T[i].ravel()])
    for rank, cost, gidx, owner, blk, first, count, rc, rr in outs:
        assert np.array_equal(rc, g1.cost[first:first + count]) and np.array_equal(rr, g1.rtn[first:first + count])   # sharding invariance, bit for bit
        assert gidx == i and cost == c and owner == i // (total // world)
        assert np.array_equal(blk, want)


def test_device_resident_sizes_are_validated_on_the_device(built):
    """With DIRECT_MEM_DEVICE the host never sees n_seg / n_planes: bad rows must come back as
    DIRECT_RTN_INVALID without touching memory outside their slab, and the other rows must be unaffected."""
    torch = _torch()
    B, N = 64, 10
    batch = problems.make_batch("corridor", B, N, seed=5)
    p0 = abi.phase0_params()
    s = solver.DdpSolver(B, N, batch.p_max, np.float64)
    want = s.solve(p0, batch)
    bad = abi.HostBatch(batch.n_seg.copy(), batch.x0, batch.xd, batch.T0, batch.n_planes.copy(), batch.planes, seeds=batch.seeds)
    bad.n_seg[3] = 0
    bad.n_seg[5] = N + 7
    bad.n_seg[6] = -2
    bad.n_planes[7, 2] = batch.p_max + 3
    bad.n_planes[9, 0] = 0
    bad.n_planes[11, N - 1] = 1 << 20
    rows = np.array([3, 5, 6, 7, 9, 11])
    # host memory: rejected before anything is launched
    with pytest.raises(solver.DirectError) as e:
        s.solve(p0, bad)
    assert e.value.status == abi.DIRECT_ERR_INVALID
    dev = torch.device("cuda:0")
    s.set_stream(torch.cuda.current_stream().cuda_stream)
    din, out = devmem.DeviceBatch(bad, dev), devmem.DeviceResult(B, N, np.float64, dev)
    s.solve_device(p0, din.cin, out.cout)
    torch.cuda.synchronize()
    got = out.to_host()
    assert (got.rtn[rows] == abi.RTN_INVALID).all()
    assert not got.bez[rows].any() and not got.T[rows].any()
    good = np.setdiff1d(np.arange(B), rows)
    for f in ("rtn", "iter_used", "cost", "T", "bez"):
        assert np.array_equal(getattr(got, f)[good], getattr(want, f)[good]), f
    # the sampler clamps instead of walking past its arrays
    d = s.sample(bad.n_seg, want.bez, want.T, 0.2, 64)
    assert d["count"][3] == 0 and d["count"][6] == 0 and d["count"][5] > 0
    # a later, clean solve on the same handle is unaffected
    again = s.solve(p0, batch)
    assert np.array_equal(again.bez, want.bez) and s.sched_error() == 0
    s.close()
