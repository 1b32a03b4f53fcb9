"""Keeps the golden generator honest: the NumPy restatement reproduces a committed fixture, and
its dense Jacobians equal the C oracle's (two independently written codes)."""
import numpy as np

from direct_amd import abi
from oracle import ddp_numpy, refapi
from tests import helpers


def test_numpy_reproduces_committed_fixture():
    g, batch = helpers.load_case("free_n5")
    p0, p1 = helpers.case_params("free_n5")
    for b in range(batch.batch):
        _, res = ddp_numpy.solve_problem(batch, b, p0)
        assert res["rtn"] == int(g["p0_rtn"][b]) and res["iter_used"] == int(g["p0_iter_used"][b])
        assert abs(res["cost"] / g["p0_cost"][b] - 1) < 1e-12
        assert np.abs(res["bez"] - g["p0_bez"][b]).max() < 1e-10


def test_dense_jacobians_agree_between_restatements():
    g, batch = helpers.load_case("corridor_n8")
    p0, _ = helpers.case_params("corridor_n8")
    d, _ = ddp_numpy.solve_problem(batch, 0, abi.phase0_params(iter_max=0))
    d.computeall()
    st = refapi.Stepper(p0, batch, 0)
    st.computeall()
    cx, cu, fx, fu, qu, quu = (st.get(i) for i in (100, 101, 102, 103, 104, 105))
    for k in range(d.N):
        nc = d.c[k].size
        assert np.abs(cx[k, :nc] - d.cx[k]).max() < 1e-12
        assert np.abs(cu[k, :nc] - d.cu[k]).max() < 1e-12
        assert np.abs(fx[k] - d.fx[k]).max() < 1e-12 and np.abs(fu[k] - d.fu[k]).max() < 1e-12
        assert np.abs(qu[k] - d.qu[k]).max() < 1e-10 and np.abs(quu[k] - d.quu[k]).max() < 1e-9
