"""The C++ drop-in boundary compiles where it will live: one translation unit with a header shaped like the
reference's data_type.h (decomp_cvx_space::Polytope / FlightCorridor with 4-vector planes read through operator())
AND direct_amd/host/ddp_optimizer.hpp, building the INTEGRATION.md snippet VERBATIM.  No GPU needed: the unit is
compiled and linked against libdirect_ddp.so, not run."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TU = r'''
#include "%(root)s/tests/cpp/fake_data_type.h"              // stands for global_planner/utils/data_type.h
#include "%(root)s/direct_amd/host/ddp_optimizer.hpp"
// the node's members / locals the snippet refers to (teach_repeat_planner.cpp:792-842, 36-80)
static Eigen::MatrixXd bezier_coeff_;
static Eigen::VectorXd bezier_range_;
static double _minimize_order = 3, _MAX_Vel = 2, _MAX_Acc = 2, _MAX_Jer = 10;
static double _w_snap_zero = 1, _w_terminal_zero = 1, _w_time_zero = 1, _w_snap = 1, _w_terminal = 100, _w_time = 20;
static int _iter_max_zero = 50, _iter_max = 100, _time_power = 2;
int fastTrajPlanning(decomp_cvx_space::FlightCorridor& corridor, const Eigen::MatrixXd& Qo_u, const Eigen::MatrixXd& Qo_l,
                     const Eigen::MatrixXd& pos, const Eigen::MatrixXd& vel, const Eigen::MatrixXd& acc,
                     const Eigen::MatrixXd& jer, Eigen::MatrixXd initbezCoeff) {
%(snippet)s
  (void)obj; (void)jerk; (void)iters;
  return rtn;
}
int main() {
  decomp_cvx_space::FlightCorridor c;
  decomp_cvx_space::Polytope p;
  Eigen::Vector4d h;
  h(3) = -1.0;
  p.appendPlane(h);
  c.appendPolytope(p);
  c.appendTime(1.0);
  Eigen::MatrixXd z(2, 3), q(1, 1), b(1, 18);
  // the corridor helpers take the node's own types as well
  std::vector<uint8_t> msg = direct::writeCorridorMsg(7, c);
  decomp_cvx_space::FlightCorridor back;
  int pid = 0;
  direct::readCorridorMsg(msg, back, pid, 8, 8);
  return fastTrajPlanning(c, q, q, z, z, z, z, b) + (int)back.polyhedrons.size() + pid;
}
'''


def test_integration_snippet_compiles_next_to_reference_shaped_types(built, tmp_path):
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"<!-- snippet:fastTrajPlanning begin -->\s*```cpp\n(.*?)```\s*<!-- snippet:fastTrajPlanning end -->", md, re.S)
    assert m, "INTEGRATION.md lost its fastTrajPlanning snippet markers"
    src = tmp_path / "tu.cpp"
    src.write_text(TU % {"root": ROOT, "snippet": m.group(1)})
    lib = os.path.join(ROOT, "direct_amd", "lib")
    r = subprocess.run(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", str(src), "-o", str(tmp_path / "tu"),
                        "-L" + lib, "-ldirect_ddp", "-Wl,-rpath," + lib + ":/opt/rocm/lib",
                        "-Wl,--unresolved-symbols=ignore-in-shared-libs"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


def test_mirror_header_defines_nothing_in_the_reference_namespace():
    txt = open(os.path.join(ROOT, "direct_amd", "host", "ddp_optimizer.hpp")).read()
    code = re.sub(r"//.*", "", txt)
    assert "namespace decomp_cvx_space" not in code
