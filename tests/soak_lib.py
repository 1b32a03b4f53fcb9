"""Randomised parity soak, shared by tests/test_gpu_soak.py (device), tests/test_soak_control.py (CPU control
experiment) and tests/soak/parity_soak.py (report).  TEST INFRASTRUCTURE.

Protocol.  A stream of small random problems (N = 3..25 segments, free-space and polyhedron corridors, up to 32 planes
per polytope), both phases of fastTrajPlanning.  An implementation is anything with begin / iterate(1) / scalars:
the device solver, the lane-loop emulator, or the oracle's Stepper.  Every implementation is STEPPED one outer
iteration at a time next to the oracle and the scalar state (cost, log-cost, mu, regulariser, line-search step,
failure flags, return code) is compared after every iteration.  Phase 1 of every implementation starts from the
ORACLE's phase-0 result through the reference's own hand-off (time-scaled Bezier control points, TRP:911-921), so
both phases compare identical inputs.

Quantities per (problem, phase):
  first_flip   first iteration whose discrete state (reg, step, fp_failed, bp_failed, done, rtn, filter size)
               differs from the oracle's; None when the whole solve agrees
  dev[k]       |cost_k / cost_k(oracle) - 1| after iteration k; pre_flip_dev = max over k < first_flip
The CONTROL experiment runs the oracle against ITSELF with every real input moved by -1 / 0 / +1 ulp: whatever it
shows is conditioning of the algorithm (DDP:440-778), not a property of any implementation."""
import numpy as np

from direct_amd import abi, problems
from oracle import refapi
from tests import helpers

DISCRETE = ("reg", "step", "fp_failed", "bp_failed", "done", "rtn", "filter_n", "infeas")


def problem_stream(n_batches, B=32, seed=2024):
    rng = np.random.default_rng(seed)
    for t in range(n_batches):
        kind = "corridor" if t % 3 else "free"
        N = int(rng.integers(3, 26))
        batch = problems.make_batch(kind, B, N, seed=int(rng.integers(1, 10 ** 6)))
        if t % 4 == 3:
            batch = helpers.with_extra_planes(batch, int(rng.integers(13, 33)), seed=t)
        yield t, kind, N, batch


def perturb_ulp(batch, seed):
    """every real input moved by -1, 0 or +1 ulp (independently, uniformly)"""
    rng = np.random.default_rng(seed)

    def p(a):
        a = np.array(a, np.float64)
        s = rng.integers(-1, 2, a.shape)
        return np.where(s > 0, np.nextafter(a, np.inf), np.where(s < 0, np.nextafter(a, -np.inf), a))
    return abi.HostBatch(batch.n_seg, p(batch.x0), p(batch.xd), p(batch.T0), batch.n_planes, p(batch.planes),
                         seeds=batch.seeds, init_bez=None if batch.init_bez is None else p(batch.init_bez),
                         infeas_in=batch.infeas_in, init_poly=None if batch.init_poly is None else p(batch.init_poly))


class OracleStepper:
    """begin / iterate / scalars over a batch, on the oracle (one refapi.Stepper per problem)."""

    def __init__(self, params, batch):
        self.st = [refapi.Stepper(params, batch, b) for b in range(batch.batch)]
        self.done = np.zeros(batch.batch, bool)
        self.iter_max = params.iter_max

    def iterate(self, n=1):
        # the oracle's stepper keeps iterating past the reference's exits: freeze a problem once it has left the
        # loop (break, DDP:335-409) or used up iter_max, which is where the device sets `done`
        for b, s in enumerate(self.st):
            for _ in range(n):
                if self.done[b]:
                    break
                brk = s.iterate(1)
                self.done[b] = bool(brk) or s.get(abi.FIELD_SCALARS)[12] >= self.iter_max

    def scalars(self):
        rows = [s.get(abi.FIELD_SCALARS) for s in self.st]
        d = {n: np.array([r[i] for r in rows]) for i, n in enumerate(abi.SCALAR_NAMES)}
        d["done"] = self.done.astype(np.float64)
        return d

    def close(self):
        for s in self.st:
            s.close()


def trace(impl, iter_max):
    """scalar state after every outer iteration: dict name -> [iter_max + 1][B] (row 0 = after begin)"""
    rows = [impl.scalars()]
    for _ in range(iter_max):
        impl.iterate(1)
        rows.append(impl.scalars())
        if np.all(rows[-1]["done"] != 0):
            break
    while len(rows) < iter_max + 1:
        rows.append(rows[-1])
    return {n: np.stack([np.asarray(r[n], np.float64) for r in rows]) for n in abi.SCALAR_NAMES}


def compare(tr, ref):
    """-> first_flip [B] (-1 = never), pre_flip_dev [B], final_dev [B], dev [K][B], same outcome (rtn, iter) [B]"""
    K, B = ref["cost"].shape
    disc = np.zeros((K, B), bool)
    for n in DISCRETE:
        disc |= tr[n] != ref[n]
    with np.errstate(divide="ignore", invalid="ignore"):
        dev = np.abs(tr["cost"] / ref["cost"] - 1.0)
    dev = np.where(np.isfinite(dev), dev, np.where(tr["cost"] == ref["cost"], 0.0, np.inf))
    first = np.where(disc.any(axis=0), disc.argmax(axis=0), -1)
    pre = np.zeros(B)
    for b in range(B):
        k = K if first[b] < 0 else first[b]
        pre[b] = dev[:k, b].max() if k > 0 else 0.0
    outcome = (tr["rtn"][-1] == ref["rtn"][-1]) & (tr["iter"][-1] == ref["iter"][-1])
    return first, pre, dev[-1], dev, outcome


def phase1_inputs(batch, r0):
    """the second polyCurveGeneration call of fastTrajPlanning from a phase-0 result (TRP:911-921)"""
    T1 = np.where((r0.rtn == 2)[:, None], r0.T, batch.T0)
    return batch.with_init(r0.bez, T0=T1, infeas_in=r0.infeas_out.astype(np.uint8))


def soak(make_impl, n_batches, B=32, seed=2024, control_seeds=(11,)):
    """Runs the protocol.  make_impl(params, batch) -> object with iterate / scalars / close (already begun).
    Returns a list of per-(batch, phase) records with the comparison of the implementation and of every control
    perturbation against the oracle."""
    p0, p1 = abi.phase0_params(), abi.phase1_params()
    out = []
    for t, kind, N, batch in problem_stream(n_batches, B, seed):
        r0, _ = refapi.solve_batch(p0, batch)
        for phase, params, pb in ((0, p0, batch), (1, p1, phase1_inputs(batch, r0))):
            o = OracleStepper(params, pb)
            ref = trace(o, params.iter_max)
            o.close()
            rec = dict(batch=t, kind=kind, N=N, p_max=int(batch.p_max), phase=phase, ref=ref)
            if make_impl is not None:
                impl = make_impl(params, pb)
                rec["impl"] = compare(trace(impl, params.iter_max), ref)
                impl.close()
            rec["control"] = []
            for cs in control_seeds:
                c = OracleStepper(params, perturb_ulp(pb, cs * 1000 + t * 2 + phase))
                rec["control"].append(compare(trace(c, params.iter_max), ref))
                c.close()
            out.append(rec)
    return out


def summarise(recs, key):
    """population figures of one comparison (key = 'impl' or ('control', i))"""
    def get(r):
        return r[key] if isinstance(key, str) else r[key[0]][key[1]]
    first = np.concatenate([get(r)[0] for r in recs])
    pre = np.concatenate([get(r)[1] for r in recs])
    fin = np.concatenate([get(r)[2] for r in recs])
    same = first < 0
    q = lambda a: [float("%.3g" % x) for x in np.quantile(a, [0.5, 0.9, 0.99, 0.999])] if len(a) else []
    # outcome = what the caller sees: return code and iteration count at exit
    outcome_same = np.concatenate([(get(r)[4]) for r in recs])
    return dict(solves=int(len(first)),
                same_outcome=int(outcome_same.sum()), same_outcome_frac=float(outcome_same.mean()),
                final_dev_same_outcome_max=float(fin[outcome_same].max()) if outcome_same.any() else 0.0,
                final_dev_same_outcome_quantiles_50_90_99_999=q(fin[outcome_same]),
                final_dev_other_outcome_top=[float("%.3g" % x) for x in sorted(fin[~outcome_same], reverse=True)[:12]],
                every_decision_identical=int(same.sum()), every_decision_identical_frac=float(same.mean()),
                first_flip_iteration_quantiles_10_50_90=[float(x) for x in np.quantile(first[~same], [0.1, 0.5, 0.9])] if (~same).any() else [],
                pre_flip_dev_max=float(pre.max()), pre_flip_dev_quantiles_50_90_99_999=q(pre),
                n_pre_flip_dev_above_1e_8=int((pre > 1e-8).sum()))


def certificate(recs, thresh_impl=1e-8, thresh_control=1e-11):
    """Per-problem check of "ill-conditioned, not wrong": every solve in which the implementation leaves the oracle
    (pre-flip deviation above thresh_impl, or any decision flips) should be one in which the ORACLE ITSELF amplifies
    a 1-ulp (1e-16) input perturbation to above thresh_control or flips a decision.  Returns (number of such solves,
    number certified by at least one control perturbation, early_dev_max = largest deviation after the first
    iteration over all solves, before anything can have been amplified)."""
    n_out = n_cert = 0
    early = 0.0
    for r in recs:
        first, pre, _, dev, _ = r["impl"]
        early = max(early, float(dev[1].max()))
        off = (pre > thresh_impl) | (first >= 0)
        sens = np.zeros_like(off)
        for c in r["control"]:
            sens |= (c[1] > thresh_control) | (c[0] >= 0)
        n_out += int(off.sum())
        n_cert += int((off & sens).sum())
    return n_out, n_cert, early


def float_storage_distribution(plan_f32, n_batches, B=32, seed=2024):
    """DIRECT_F32 (float storage, double arithmetic) against the fp64 oracle on the soak stream, whole two-phase
    plans.  plan_f32(p0, p1, batch) -> (res0, res1) of the implementation.  Returns the distribution of the
    deviations at exit: the evidence behind the float-storage tolerances of the parity tests."""
    p0, p1 = abi.phase0_params(), abi.phase1_params()
    dc, dT, same_rtn, dit, feas = [], [], [], [], []
    for t, kind, N, batch in problem_stream(n_batches, B, seed):
        r0, r1 = refapi.plan_batch(p0, p1, batch)
        g0, g1 = plan_f32(p0, p1, batch)
        ok = (r1.rtn >= 0) & (g1.rtn >= 0)
        dc += list(np.abs(g1.cost[ok] / r1.cost[ok] - 1))
        with np.errstate(divide="ignore", invalid="ignore"):
            rt = np.abs(g1.T[ok] - r1.T[ok]).max(axis=1) / np.abs(r1.T[ok]).max(axis=1)
        dT += list(rt)
        same_rtn += list(g1.rtn == r1.rtn)
        dit += list(np.abs(g1.iter_used.astype(int) - r1.iter_used.astype(int))[ok])
        feas += list((g1.rtn >= 0) == (r1.rtn >= 0))
    dc, dT, dit = np.array(dc), np.array(dT), np.array(dit)
    q = lambda a: [float("%.3g" % x) for x in np.quantile(a, [0.5, 0.9, 0.99, 1.0])]
    return dict(plans=len(same_rtn), same_rtn_frac=float(np.mean(same_rtn)), same_feasibility_frac=float(np.mean(feas)),
                cost_rel_dev_quantiles_50_90_99_max=q(dc), T_rel_dev_quantiles_50_90_99_max=q(dT),
                iter_diff_quantiles_50_90_99_max=q(dit), frac_cost_dev_below_1e_3=float((dc < 1e-3).mean()),
                frac_cost_dev_below_2e_2=float((dc < 2e-2).mean()))
