"""BASELINE configs 4 and 5 on the device, and the boundary behaviours that only exist with device-resident
arrays: on-device validation of n_seg / n_planes, the sticky scheduler-error flag and the config-5 gather
through the library's C entry points (direct_ddp_gather_best over RCCL, world size 1 on this box).

Config 4: B = 16384, N = 300, double storage, config-3 generator.
Config 5: 131072 corridors over 8 GPUs = 16384 per GPU, N = 100, float storage; one shard is solved here.
The oracle cannot solve these batches in seconds: parity is exact agreement on a sample plus the
size-independent properties of tests/test_gpu_fullsize.py on the whole batch."""
import numpy as np
import pytest

from direct_amd import abi, devmem, distributed, problems, solver
from oracle import refapi
from tests import helpers
from tests.test_gpu_fullsize import check_properties

pytestmark = pytest.mark.gpu


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
    # exact agreement with the oracle on a sample (both phases, every discrete decision)
    idx = np.array([5, 8191, 16383])
    r0, r1 = refapi.plan_batch(p0, p1, batch.select(idx))
    assert (g0.rtn[idx] == r0.rtn).all() and (g0.iter_used[idx] == r0.iter_used).all()
    assert (g1.rtn[idx] == r1.rtn).all() and (g1.iter_used[idx] == r1.iter_used).all()
    assert np.abs(g1.cost[idx] / r1.cost - 1).max() < 1e-5   # Bezier vs monomial hand-off between the phases
    assert helpers.rel(g1.T[idx], r1.T) < 1e-3
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
