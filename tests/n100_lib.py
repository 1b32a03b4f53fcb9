"""Parity at the configurations' OWN size and precision (N = 100 segments; BASELINE configs 2, 3, 5), shared by
tests/test_gpu_n100.py and tests/soak/n100_report.py.  TEST INFRASTRUCTURE.

What is compared (a sample of a BASELINE batch, both phases of fastTrajPlanning, TRP:886-921):
  * the device with double storage against the fp64 oracle (identical inputs; phase 1 of both starts from the
    ORACLE's phase-0 result through the reference's own Bezier hand-off, TRP:911-921);
  * the device with float storage (DIRECT_F32) against the fp64 oracle given the same float-rounded inputs (phase 1
    from the oracle's phase-0 result, handed over as monomial coefficients: include/direct_ddp.h, init_poly);
  * the CONTROL experiments, oracle against ITSELF: every real input moved by -1 / 0 / +1 ulp of a DOUBLE (what an
    exact reimplementation in another operation order looks like) and by -1 / 0 / +1 ulp of a FLOAT (what merely
    storing the inputs in float does to the algorithm).  Whatever the controls show is conditioning of the algorithm
    at N = 100 (DDP:440-778), not a property of any implementation; the device is bounded BY them.

Also here: the sample of the TIMED launch of bench.py (fixed 20 phase-1 iterations from the device's own warm
start), stepped one outer iteration at a time next to the oracle."""
import numpy as np

from direct_amd import abi, problems
from oracle import refapi
from tests import soak_lib


def perturb_float_ulp(batch, seed):
    """every real input moved by -1, 0 or +1 ulp OF A FLOAT32 (independently, uniformly), kept in double"""
    rng = np.random.default_rng(seed)

    def p(a):
        a32 = np.array(a, np.float32)
        s = rng.integers(-1, 2, a32.shape)
        up, dn = np.nextafter(a32, np.float32(np.inf)), np.nextafter(a32, np.float32(-np.inf))
        return np.where(s > 0, up, np.where(s < 0, dn, a32)).astype(np.float64)
    return abi.HostBatch(batch.n_seg, p(batch.x0), p(batch.xd), p(batch.T0), batch.n_planes, p(batch.planes),
                         seeds=batch.seeds, init_bez=None if batch.init_bez is None else p(batch.init_bez),
                         infeas_in=batch.infeas_in, init_poly=None if batch.init_poly is None else p(batch.init_poly))


def compare(res, ref, detail=False):
    """what the caller sees at exit: return code, iteration count, cost, durations (detail: also per problem)"""
    both = (res.rtn >= 0) & (ref.rtn >= 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        dc = np.abs(res.cost / ref.cost - 1.0)
        dT = np.abs(res.T - ref.T).max(axis=1) / np.abs(ref.T).max(axis=1)
    q = lambda a: [float("%.3g" % x) for x in np.quantile(a, [0.5, 0.9, 1.0])] if len(a) else []
    same = (res.rtn == ref.rtn) & (res.iter_used == ref.iter_used)
    out = dict(n=int(len(ref.rtn)), same_rtn=int((res.rtn == ref.rtn).sum()), same_outcome=int(same.sum()),
               same_feasibility=int(((res.rtn >= 0) == (ref.rtn >= 0)).sum()),
               same_infeas_out=int((res.infeas_out == ref.infeas_out).sum()),
               cost_dev_q50_q90_max=q(dc[both]), cost_dev_same_outcome_q50_q90_max=q(dc[both & same]),
               T_dev_q50_q90_max=q(dT[both]), n_cost_dev_below_1e_8=int((dc[both] < 1e-8).sum()),
               n_cost_dev_below_1e_3=int((dc[both] < 1e-3).sum()), n_both_ok=int(both.sum()),
               iter_diff_max=int(np.abs(res.iter_used.astype(int) - ref.iter_used.astype(int)).max()))
    if detail:
        bez = np.abs(res.bez - ref.bez).reshape(len(ref.rtn), -1).max(axis=1) / np.maximum(np.abs(ref.bez).reshape(len(ref.rtn), -1).max(axis=1), 1e-300)
        out["per_problem"] = dict(same=[bool(v) for v in same], both=[bool(v) for v in both], cost_dev=[float(v) for v in dc],
                                  T_dev=[float(v) for v in dT], bez_dev=[float(v) for v in bez], ref_rtn=[int(v) for v in ref.rtn],
                                  ref_iter=[int(v) for v in ref.iter_used])
    return out


def two_phase(solve, p0, p1, batch0, handoff):
    """phase 0 on `batch0`, then phase 1 from `handoff(batch0-like)`; solve(params, batch) -> result"""
    r0 = solve(p0, batch0)
    return r0, solve(p1, handoff)


def sample_report(kind, B, N, idx, dev64, dev32, seed=1000, first=0, control_seeds=(11, 12)):
    """dev64 / dev32: solve(params, HostBatch) -> HostResult on the device (double / float storage), or None.
    Returns the comparison records of one BASELINE batch sample (both phases)."""
    full = problems.make_batch(kind, B, N, seed=seed, first=first)
    out = batch_report(full.select(idx), dev64, dev32, control_seeds)
    out.update(kind=kind, batch=B, n_seg=N, first=int(first))
    return out


def batch_report(sb, dev64, dev32, control_seeds=(11, 12), detail=False):
    """The comparison records of the problems `sb` (any HostBatch): device (double / float storage) against the oracle,
    both phases, phase 1 of every implementation from the ORACLE's phase-0 result, with the oracle-against-itself
    controls (inputs moved by one ulp of a double / of a float) next to them.  detail: per-problem records as well
    (which problems the double-ulp controls reproduce exactly - the device is then held to the oracle exactly)."""
    p0, p1 = abi.phase0_params(), abi.phase1_params()
    cmp = lambda a, b: compare(a, b, detail)
    ora = lambda p, b: refapi.solve_batch(p, b)[0]
    r0 = ora(p0, sb)
    b1 = soak_lib.phase1_inputs(sb, r0)             # the reference's own hand-off: time-scaled Bezier points
    r1 = ora(p1, b1)
    out = dict(sample=int(sb.batch),
               oracle=dict(phase0_rtn={str(int(v)): int(c) for v, c in zip(*np.unique(r0.rtn, return_counts=True))},
                           phase1_rtn={str(int(v)): int(c) for v, c in zip(*np.unique(r1.rtn, return_counts=True))},
                           phase0_iters_mean=float(r0.fwd_passes.mean()), phase1_iters_mean=float(r1.fwd_passes.mean())))
    out["control_double_ulp"] = [dict(phase0=cmp(ora(p0, soak_lib.perturb_ulp(sb, 1000 * cs)), r0),
                                      phase1=cmp(ora(p1, soak_lib.perturb_ulp(b1, 1000 * cs + 1)), r1))
                                 for cs in control_seeds]
    # float-level references: the oracle on float-rounded inputs (what DIRECT_F32 is given), monomial hand-off
    sb32 = sb.astype(np.float32).astype(np.float64)
    q0 = ora(p0, sb32)
    T1 = np.where((q0.rtn == 2)[:, None], q0.T, sb32.T0)
    c1 = sb32.with_init(None, T0=T1, infeas_in=q0.infeas_out.astype(np.uint8), init_poly=q0.poly)
    c1 = c1.astype(np.float32).astype(np.float64)
    q1 = ora(p1, c1)
    out["control_float_ulp"] = [dict(phase0=cmp(ora(p0, perturb_float_ulp(sb32, 1000 * cs)), q0),
                                     phase1=cmp(ora(p1, perturb_float_ulp(c1, 1000 * cs + 1)), q1))
                                for cs in control_seeds]
    if dev64 is not None:
        out["device_f64"] = dict(phase0=cmp(dev64(p0, sb), r0), phase1=cmp(dev64(p1, b1), r1))
    if dev32 is not None:
        out["device_f32"] = dict(phase0=cmp(dev32(p0, sb32), q0), phase1=cmp(dev32(p1, c1), q1))
    return out


def timed_launch_inputs(kind, B, N, idx, dev32_solve, iters=20):
    """The inputs of bench.py's timed launch for the problems `idx` of the batch: phase 1 warm-started from the
    DEVICE's own float-storage phase-0 result (monomial hand-off), early exits disabled, `iters` iterations."""
    sb = problems.make_batch(kind, B, N, seed=1000).select(idx).astype(np.float32)
    g0 = dev32_solve(abi.phase0_params(), sb)
    b1 = sb.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, sb.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
    return b1, abi.phase1_params(iter_max=iters, fixed_iters=1)


def stepped(impl, iters):
    """scalar state after every outer iteration (soak_lib.trace without the early stop)"""
    rows = [impl.scalars()]
    for _ in range(iters):
        impl.iterate(1)
        rows.append(impl.scalars())
    return {n: np.stack([np.asarray(r[n], np.float64) for r in rows]) for n in abi.SCALAR_NAMES}


def compare_stepped(tr, ref):
    """-> per-problem first differing discrete decision (-1 = none) and the cost deviation per iteration"""
    disc = np.zeros(ref["cost"].shape, bool)
    for n in ("reg", "step", "fp_failed", "bp_failed", "filter_n", "infeas"):
        disc |= tr[n] != ref[n]
    first = np.where(disc.any(axis=0), disc.argmax(axis=0), -1)
    with np.errstate(divide="ignore", invalid="ignore"):
        dev = np.abs(tr["cost"] / ref["cost"] - 1.0)
    return first, dev
