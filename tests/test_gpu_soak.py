"""Randomised parity soak on the device with explicit bounds (tests/soak_lib.py has the protocol).

1152 random small solves (both phases of 576 problems), fp64, stepped ONE OUTER ITERATION AT A TIME next to the
oracle.  The whole-solve agreement SURVEY.md 8(c) asks for (identical rtn / iteration count, cost 1e-8) cannot hold
for every random problem, because the ALGORITHM is ill-conditioned on about 1 % of them: the oracle run against
ITSELF with its inputs moved by one ulp (the control experiment, same problems, same protocol) flips decisions and
ends 10-30 % apart on the same fraction.  The bounds below therefore tie the device to the control:
  * after the first iteration (nothing amplified yet) every solve agrees to 1e-9;
  * the fractions "same outcome" and "every decision identical" are within half a percent of the control's;
  * the number of solves whose cost leaves the oracle by more than 1e-8 BEFORE any decision differs is no larger
    than the control's (x 1.5 + 5);
  * at least 90 % of the solves in which the device leaves the oracle are solves in which the oracle itself turns a
    1e-16 input perturbation into more than 1e-11 or flips a decision (per-problem certificate)."""
import json
import os

import numpy as np
import pytest

from direct_amd import solver
from tests import soak_lib

pytestmark = pytest.mark.gpu
N_BATCHES = 18


class DeviceStepper:
    def __init__(self, params, batch, dtype=np.float64):
        self.s = solver.DdpSolver(batch.batch, batch.n_seg_max, batch.p_max, dtype)
        self.s.begin(params, batch)

    def iterate(self, n):
        self.s.iterate(n)

    def scalars(self):
        return self.s.scalars()

    def close(self):
        self.s.close()


def test_fp64_soak_is_bounded_by_the_oracles_own_conditioning(built):
    recs = soak_lib.soak(DeviceStepper, N_BATCHES, control_seeds=(11, 12))
    dev = soak_lib.summarise(recs, "impl")
    ctl = [soak_lib.summarise(recs, ("control", i)) for i in range(2)]
    n_out, n_cert, early = soak_lib.certificate(recs)
    report = dict(device=dev, control=ctl, certificate=dict(left_oracle=n_out, certified_ill_conditioned=n_cert,
                                                           max_dev_after_first_iteration=early))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        json.dump(report, open(os.path.join(out, "soak_test_report.json"), "w"), indent=1)
    assert dev["solves"] == N_BATCHES * 64
    assert early < 1e-9, early
    assert dev["same_outcome_frac"] >= min(c["same_outcome_frac"] for c in ctl) - 0.005, report
    assert dev["same_outcome_frac"] >= 0.985
    assert dev["every_decision_identical_frac"] >= min(c["every_decision_identical_frac"] for c in ctl) - 0.005, report
    assert dev["n_pre_flip_dev_above_1e_8"] <= 1.5 * max(c["n_pre_flip_dev_above_1e_8"] for c in ctl) + 5, report
    assert dev["pre_flip_dev_quantiles_50_90_99_999"][1] < 1e-9      # 90 % of the solves never leave 1e-9
    assert n_cert >= 0.9 * n_out, report


def test_float_storage_deviation_distribution(built):
    """DIRECT_F32 = float storage + double arithmetic.  The distribution of its deviation from the fp64 oracle at
    exit is what the float tolerances of the parity tests rest on: the bulk agrees to 1e-5, and the tail is the same
    ill-conditioned ~1-2 % that fp64 shows against itself (SURVEY.md 8c states 1e-3 for the whole solve)."""
    def plan(p0, p1, batch):
        s = solver.DdpSolver(batch.batch, batch.n_seg_max, batch.p_max, np.float32)
        r = s.plan(p0, p1, batch)
        s.close()
        return r
    d = soak_lib.float_storage_distribution(plan, 12)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        json.dump(d, open(os.path.join(out, "soak_f32_report.json"), "w"), indent=1)
    assert d["same_feasibility_frac"] >= 0.995 and d["same_rtn_frac"] >= 0.98, d
    q50, q90, q99, qmax = d["cost_rel_dev_quantiles_50_90_99_max"]
    assert q50 < 1e-5 and q90 < 1e-4, d
    assert d["frac_cost_dev_below_1e_3"] >= 0.97, d
    assert d["T_rel_dev_quantiles_50_90_99_max"][1] < 1e-3, d
