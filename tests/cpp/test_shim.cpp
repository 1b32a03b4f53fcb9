// C++ host-side check (GPU needed): drives the reference-shaped class of direct_amd/host/ddp_optimizer.hpp
// exactly as fastTrajPlanning does (teach_repeat_planner.cpp:886-921: phase 0, UpdateTime, phase 1) and
// compares the result with the CPU oracle linked in as the checker.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../direct_amd/host/ddp_optimizer.hpp"

extern "C" int direct_ref_plan_batch(const direct_ddp_params_t*, const direct_ddp_params_t*, const direct_ddp_batch_in_t*,
                                     direct_ddp_batch_out_t*, direct_ddp_batch_out_t*, int);

using direct::DenseMatrix;
using direct::DenseVector;

extern "C" int direct_ref_solve_batch(const direct_ddp_params_t*, const direct_ddp_batch_in_t*, direct_ddp_batch_out_t*, int, double*, int);
extern "C" int direct_ref_sample(int n_seg, const double* bez, const double* T, double dt, int capacity, int derivs,
                                 int* seg_first, double* pos, double* vel, double* acc, double* length, double* vmax,
                                 double* amax);

int main() {
  const int N = 6, B = 2;
  std::vector<direct::PlainCorridor> cors(B);
  std::vector<DenseMatrix> pos, vel, acc, jer, bez0;
  for (int b = 0; b < B; b++) {
    DenseMatrix p(2, 3), z(2, 3);
    p(0, 0) = 1.0 + b; p(0, 1) = -2.0; p(0, 2) = 1.0;
    p(1, 0) = p(0, 0) + 2.5 * N; p(1, 1) = -2.0 + 0.3 * N; p(1, 2) = 1.2;
    for (int k = 0; k < N; k++) {
      direct::PlainPolytope pl;
      const double cx = p(0, 0) + 2.5 * (k + 0.5), cy = -2.0 + 0.3 * (k + 0.5), cz = 1.1;
      // axis-aligned box of half-width (2.6, 1.5, 1.0) around the segment midpoint, plus one slanted cut
      const double hx = 2.6, hy = 1.5, hz = 1.0;
      pl.appendPlane({1, 0, 0, -(cx + hx)}); pl.appendPlane({-1, 0, 0, cx - hx});
      pl.appendPlane({0, 1, 0, -(cy + hy)}); pl.appendPlane({0, -1, 0, cy - hy});
      pl.appendPlane({0, 0, 1, -(cz + hz)}); pl.appendPlane({0, 0, -1, cz - hz});
      const double s = std::sqrt(0.5);
      pl.appendPlane({0, s, s, -(s * cy + s * cz + 1.2)});
      pl.seed_coord = {{p(0, 0) + 2.5 * k, -2.0 + 0.3 * k, 1.0}};
      cors[b].appendPolytope(pl);
      cors[b].appendTime(2.2);
    }
    pos.push_back(p); vel.push_back(z); acc.push_back(z); jer.push_back(z);
    bez0.push_back(DenseMatrix(N, 18));
  }
  direct::DdpDevice dev(B, N, 8, DIRECT_F64);
  DenseMatrix none(1, 1);
  std::vector<uint8_t> infeas(B, 1), line_failed(B, 1);
  direct::ddpTrajOptimizer<> opt0(dev), opt1(dev);
  // phase 0 (TRP:895-897)
  auto rtn0 = opt0.polyCurveGenerationBatch(cors, none, none, pos, vel, acc, jer, 3.0, 2.0, 2.0, 10.0, bez0, 1.0, 1.0, 1.0, 50,
                                       infeas, true, false, line_failed, 2, false);
  std::vector<DenseMatrix> bez1;
  auto cors1 = cors;
  for (int b = 0; b < B; b++) {
    if (rtn0[b] == 2) {  // UpdateTime (TRP:911-915)
      auto T = opt0.getPolyTime(b);
      cors1[b].durations.clear();
      for (int k = 0; k < T.size(); k++) cors1[b].appendTime(T(k));
    }
    bez1.push_back(opt0.getBezCoeff(b));  // TRP:918
  }
  // phase 1 (TRP:919-921)
  auto rtn1 = opt1.polyCurveGenerationBatch(cors1, none, none, pos, vel, acc, jer, 3.0, 2.0, 2.0, 10.0, bez1, 1.0, 100.0, 20.0, 100,
                                       infeas, false, false, line_failed, 2, false);

  // the oracle on the same flat inputs
  std::vector<int32_t> n_seg(B, N), n_planes((size_t)B * N, 7), r0(B), r1(B), it1(B);
  std::vector<double> x0(B * 9, 0.0), xd(B * 9, 0.0), T0((size_t)B * N), planes((size_t)B * N * 8 * 4, 0.0), c1(B), Tout((size_t)B * N);
  for (int b = 0; b < B; b++) {
    for (int d = 0; d < 3; d++) { x0[b * 9 + d] = pos[b](0, d); xd[b * 9 + d] = pos[b](1, d); }
    for (int k = 0; k < N; k++) {
      T0[(size_t)b * N + k] = 2.2;
      for (int p = 0; p < 7; p++)
        for (int q = 0; q < 4; q++) planes[(((size_t)b * N + k) * 8 + p) * 4 + q] = cors[b].polyhedrons[k].planes[p][q];
    }
  }
  direct_ddp_params_t p0{2.0, 2.0, 1.0, 1.0, 1.0, 50, 2, 1, 0, 0, 1, 0, 0}, p1{2.0, 2.0, 1.0, 100.0, 20.0, 100, 2, 0, 0, 0, 0, 0, 0};
  direct_ddp_batch_in_t in{};
  in.batch = B; in.n_seg_max = N; in.p_max = 8; in.mem = DIRECT_MEM_HOST; in.n_seg = n_seg.data(); in.x0 = x0.data();
  in.xd = xd.data(); in.T0 = T0.data(); in.n_planes = n_planes.data(); in.planes = planes.data();
  direct_ddp_batch_out_t o0{}, o1{};
  o0.rtn = r0.data(); o1.rtn = r1.data(); o1.iter_used = it1.data(); o1.cost = c1.data(); o1.T = Tout.data();
  direct_ref_plan_batch(&p0, &p1, &in, &o0, &o1, 1);
  int bad = 0;
  for (int b = 0; b < B; b++) {
    const double rel = std::fabs(opt1.getDDPObjective(b) / c1[b] - 1.0);
    std::printf("corridor %d: rtn0 %d/%d rtn1 %d/%d iters %d/%d cost %.9g/%.9g rel %.2e T0 %.6f/%.6f\n", b, rtn0[b], r0[b], rtn1[b],
                r1[b], opt1.getIterUsed(b), it1[b], opt1.getDDPObjective(b), c1[b], rel, opt1.getPolyTime(b)(0), Tout[(size_t)b * N]);
    if (rtn0[b] != r0[b] || rtn1[b] != r1[b] || opt1.getIterUsed(b) != it1[b] || rel > 1e-8) bad++;
  }
  // the steps around the path: corridor message round trip and output sampling against the oracle
  {
    for (size_t k = 0; k < cors[0].polyhedrons.size(); k++) cors[0].polyhedrons[k].center = {{0.5 + k, -1.0, 1.0}};
    const std::vector<uint8_t> msg = direct::writeCorridorMsg(42, cors[0]);
    direct::PlainCorridor back;
    int pid = 0;
    direct::readCorridorMsg(msg, back, pid, 16, 8);
    bool same = pid == 42 && back.polyhedrons.size() == cors[0].polyhedrons.size();
    for (size_t k = 0; same && k < back.polyhedrons.size(); k++) {
      const auto &a = back.polyhedrons[k], &b = cors[0].polyhedrons[k];
      same = a.planes == b.planes && a.center == b.center && a.seed_coord == b.seed_coord;
    }
    std::printf("corridor message: %zu bytes, round trip %s\n", msg.size(), same ? "ok" : "MISMATCH");
    if (!same) bad++;
    double len = 0.0, len_ref = 0.0;
    const auto pts = direct::sampleBezierTrajectory(dev, opt1.getBezCoeff(0), opt1.getPolyTime(0), 0.2, &len);
    std::vector<double> bz((size_t)N * 18), Tt(N), pref(pts.size() * 3 + 3);
    for (int k = 0; k < N; k++) {
      Tt[k] = opt1.getPolyTime(0)(k);
      for (int q = 0; q < 18; q++) bz[(size_t)k * 18 + q] = opt1.getBezCoeff(0)(k, q);
    }
    const int cnt = direct_ref_sample(N, bz.data(), Tt.data(), 0.2, (int)pts.size(), 0, nullptr, pref.data(), nullptr, nullptr,
                                      &len_ref, nullptr, nullptr);
    double err = 0.0;
    for (size_t i = 0; i < pts.size() && (int)i < cnt; i++)
      for (int d = 0; d < 3; d++) err = std::fmax(err, std::fabs(pts[i][d] - pref[i * 3 + d]));
    std::printf("sampling: %zu points (oracle %d), max |dp| %.2e, length %.9f / %.9f\n", pts.size(), cnt, err, len, len_ref);
    if ((int)pts.size() != cnt || err > 1e-10 || std::fabs(len / len_ref - 1.0) > 1e-12) bad++;
  }
  // the reference's exact single-corridor signature (ddp_optimizer.h:267-289: bool& infeas, bool& line_failed) on a
  // default-constructed optimiser (`new ddpTrajOptimizer()`, TRP:853): same answer as row 0 of the batch
  {
    direct::DdpDevice::configure_shared(1, N, 8, DIRECT_F64);
    direct::ddpTrajOptimizer<> a, c;
    bool inf = true, lf = true;
    const int s0 = a.polyCurveGeneration(cors[0], none, none, pos[0], vel[0], acc[0], jer[0], 3.0, 2.0, 2.0, 10.0, bez0[0], 1.0, 1.0, 1.0,
                                         50, inf, true, false, lf, 2, false);
    const int s1 = c.polyCurveGeneration(cors1[0], none, none, pos[0], vel[0], acc[0], jer[0], 3.0, 2.0, 2.0, 10.0, a.getBezCoeff(), 1.0,
                                         100.0, 20.0, 100, inf, false, false, lf, 2, false);
    const bool ok = s0 == rtn0[0] && s1 == rtn1[0] && c.getIterUsed() == opt1.getIterUsed(0) &&
                    c.getDDPObjective() == opt1.getDDPObjective(0) && lf == true && inf == (infeas[0] != 0);
    std::printf("single-corridor overload: rtn %d/%d, cost %.9g, line_failed %d: %s\n", s0, s1, c.getDDPObjective(), (int)lf,
                ok ? "ok" : "MISMATCH");
    if (!ok) bad++;
    // line_init_flag: `line_failed` is cleared only by the line-init success exit (ddp_optimizer.cpp:384)
    direct::ddpTrajOptimizer<> l;
    bool inf2 = true, lf2 = true;
    const int s2 = l.polyCurveGeneration(cors[0], none, none, pos[0], vel[0], acc[0], jer[0], 3.0, 2.0, 2.0, 10.0, bez0[0], 1.0, 100.0, 20.0,
                                         60, inf2, false, true, lf2, 2, false);
    std::vector<int32_t> rl(1), il(1);
    std::vector<uint8_t> lfo(1, 9);
    direct_ddp_params_t pl{2.0, 2.0, 1.0, 100.0, 20.0, 60, 2, 0, 1, 0, 1, 0, 0};
    std::vector<double> sd((size_t)N * 3);
    for (int k = 0; k < N; k++)
      for (int d = 0; d < 3; d++) sd[(size_t)k * 3 + d] = cors[0].polyhedrons[k].seed_coord[d];
    direct_ddp_batch_in_t in1 = in;
    in1.batch = 1; in1.seeds = sd.data();
    std::vector<double> cl(1);
    direct_ddp_batch_out_t ol{};
    ol.rtn = rl.data(); ol.iter_used = il.data(); ol.line_failed_out = lfo.data(); ol.cost = cl.data();
    direct_ref_solve_batch(&pl, &in1, &ol, 1, nullptr, 0);
    const bool okl = s2 == rl[0] && l.getIterUsed() == il[0] && (lf2 ? 1 : 0) == (lfo[0] ? 1 : 0) &&
                     std::fabs(l.getDDPObjective() / cl[0] - 1.0) < 1e-8;
    std::printf("line-init: rtn %d/%d iters %d/%d line_failed %d/%d: %s\n", s2, rl[0], l.getIterUsed(), il[0], (int)lf2, (int)lfo[0],
                okl ? "ok" : "MISMATCH");
    if (!okl) bad++;
  }
  std::printf(bad ? "FAIL\n" : "PASS\n");
  return bad ? 1 : 0;
}
