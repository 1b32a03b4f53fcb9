"""Output sampling: the C restatement of the caller's sampling loop
(teach_repeat_planner.cpp:1551-1566 over utils/bezier_base.h:77-127) against the committed golden
vectors of the independent NumPy restatement, and against closed forms."""
import os

import numpy as np
import pytest

from oracle import refapi
from tests import helpers

SAMPLE_CASES = ("corridor_n8", "free_n5", "config1_n50")


def load(case):
    return np.load(os.path.join(helpers.GOLDEN_DIR, "sample_" + case + ".npz"))


@pytest.mark.parametrize("case", SAMPLE_CASES)
def test_oracle_matches_golden(case):
    g = load(case)
    o = refapi.sample_batch(g["n_seg"], g["bez"], g["T"], float(g["dt"]), int(g["capacity"]))
    assert (o["count"] == g["count"]).all()
    assert (o["seg_first"] == g["seg_first"]).all()
    for k in ("pos", "vel", "acc", "length", "vmax", "amax"):
        assert helpers.rel(o[k], g[k]) < 1e-13, k


def test_straight_segment_closed_form_and_sample_count_quirk():
    """Control points on a line: position is linear in t, velocity constant, acceleration zero.  T = 2,
    dt = 0.2 gives ELEVEN samples: ten additions of 0.1 stay below 1.0 in binary floating point."""
    T = np.array([[2.0]])
    p0, p1 = np.array([1.0, 2.0, 3.0]), np.array([3.0, 2.0, -1.0])
    bez = np.zeros((1, 1, 18))
    for d in range(3):
        for j in range(6):
            bez[0, 0, d * 6 + j] = (p0[d] + (p1[d] - p0[d]) * j / 5) / T[0, 0]
    o = refapi.sample_batch([1], bez, T, 0.2, 32)
    assert o["count"][0] == 11
    t = np.cumsum(np.r_[0.0, np.full(10, 0.1)])
    assert np.allclose(o["pos"][0, :11], p0 + t[:, None] * (p1 - p0), atol=1e-13)
    assert np.allclose(o["vel"][0, :11], (p1 - p0) / 2.0, atol=1e-13)
    assert np.abs(o["acc"][0, :11]).max() < 1e-13
    assert abs(o["length"][0] - np.linalg.norm(p1 - p0) * t[-1]) < 1e-12


def test_negative_duration_and_capacity():
    g = load("corridor_n8")
    T = g["T"].copy()
    T[1, 3] = -0.5
    o = refapi.sample_batch(g["n_seg"], g["bez"], T, 0.2, 16)
    assert o["count"][1] == -1 and o["count"][0] == g["count"][0]          # count is not clamped to capacity
    assert helpers.rel(o["pos"][0], g["pos"][0, :16]) < 1e-13 and helpers.rel(o["length"][0], g["length"][0]) < 1e-13
