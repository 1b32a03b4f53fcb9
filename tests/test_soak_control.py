"""The control experiment of the parity soak on the CPU (no device): the oracle against ITSELF with every real input
moved by -1 / 0 / +1 ulp, and the kernel SOURCE (lane-loop emulator) against the oracle, same problems, same
stepwise protocol (tests/soak_lib.py).  Shows that the ~1 % of random problems on which an implementation leaves the
oracle is the algorithm's own conditioning: the emulated kernels are no further from the oracle than the oracle is
from itself.  (tests/test_gpu_soak.py holds the device to the same yardstick.)"""
import numpy as np

from tests import soak_lib
from tests.emu import emuapi


class EmuStepper:
    def __init__(self, params, batch):
        self.e = emuapi.EmuSolver(params, batch)

    def iterate(self, n):
        self.e.iterate(n)

    def scalars(self):
        return self.e.scalars()

    def close(self):
        self.e.close()


def test_emulated_kernels_are_as_close_to_the_oracle_as_the_oracle_is_to_itself():
    recs = soak_lib.soak(EmuStepper, 8, control_seeds=(11, 12))
    emu = soak_lib.summarise(recs, "impl")
    ctl = [soak_lib.summarise(recs, ("control", i)) for i in range(2)]
    n_out, n_cert, early = soak_lib.certificate(recs)
    assert emu["solves"] == 512 and early < 1e-9
    assert emu["same_outcome_frac"] >= min(c["same_outcome_frac"] for c in ctl) - 0.01
    assert emu["every_decision_identical_frac"] >= min(c["every_decision_identical_frac"] for c in ctl) - 0.01
    assert emu["n_pre_flip_dev_above_1e_8"] <= 1.5 * max(c["n_pre_flip_dev_above_1e_8"] for c in ctl) + 5
    assert n_cert >= 0.9 * n_out
    # the control itself: a 1-ulp perturbation is amplified beyond 1e-8 somewhere, and most solves are untouched
    assert all(c["pre_flip_dev_quantiles_50_90_99_999"][0] < 1e-11 for c in ctl)
    assert max(c["pre_flip_dev_max"] for c in ctl) > 1e-9


def test_n100_controls_run_without_a_device():
    """The N = 100 comparison library (tests/n100_lib.py) on the CPU: oracle against itself with its inputs moved by one
    ulp of a double / of a float on 4 corridors of the config-3 batch.  A double ulp leaves every exit and the cost
    (1e-8) untouched on these; a float ulp moves the phase-1 cost by more than 1e-8 - the yardstick the device's float
    storage is held to in tests/test_gpu_n100.py."""
    from tests import n100_lib
    r = n100_lib.sample_report("corridor", 4096, 100, np.arange(0, 4096, 1024), None, None, control_seeds=(11,))
    c64, c32 = r["control_double_ulp"][0], r["control_float_ulp"][0]
    assert c64["phase0"]["same_outcome"] == 4 and c64["phase0"]["n_cost_dev_below_1e_8"] == 4
    assert c64["phase1"]["same_feasibility"] == 4 and c64["phase1"]["cost_dev_q50_q90_max"][0] < 1e-8
    assert c32["phase0"]["same_feasibility"] == 4 and c32["phase0"]["cost_dev_q50_q90_max"][0] > 1e-9
    assert c32["phase1"]["cost_dev_q50_q90_max"][2] > 1e-8
