// Host-side C++ mirror of the reference's optimiser interface on top of the C-ABI (include/direct_ddp.h).
//
// The reference class is `ddpTrajOptimizer` (global_planner/include/global_planner/ddp_optimizer.h:241-342)
// with `int polyCurveGeneration(const decomp_cvx_space::FlightCorridor& corridor, MQM_u, MQM_l, pos, vel, acc, jer,
// minimize_order, max_vel, max_acc, max_jer, initbezCoeff, w_snap, w_terminal, w_time, iter_max, bool& infeas,
// zero_init_flag, line_init_flag, bool& line_failed, time_power, minvo_flag)` (ddp_optimizer.h:267-289) and the
// getters getPolyCoeff(), getBezCoeff(), getPolyTime(), getDDPObjective(), getCompTime(), getTerminalNorm(),
// getIterUsed(), getJerkCost() (:299-340).  This header keeps those names, the argument order and the return codes:
//   * polyCurveGeneration(corridor, ...)       the reference's exact single-corridor signature
//   * polyCurveGenerationBatch(corridors, ...) the same for B corridors in one launch
// It defines NO type of namespace decomp_cvx_space: the corridor type is a template parameter, read through the
// member names of global_planner/include/global_planner/utils/data_type.h:124-245 (durations, polyhedrons, planes,
// seed_coord, center), so inside the node it takes the node's own FlightCorridor (Eigen::Vector4d planes) and in the
// tests, which have no Eigen, direct::PlainCorridor.  Likewise the dense types: Eigen::MatrixXd / VectorXd in the
// node, direct::DenseMatrix / DenseVector in the tests.
//
// Requirements on Mat: Mat(rows, cols), rows(), cols(), operator()(i, j).  On Vec: Vec(n), size(), operator()(i).
// Plane / point element access: v(i) (Eigen) or v[i] (std::array), whichever the type has.
#pragma once
#include <algorithm>
#include <array>
#include <chrono>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/direct_ddp.h"

namespace direct {

// element i of a small fixed vector: Eigen's v(i) when it exists, otherwise v[i]
template <class V>
auto elem(const V& v, int i, int) -> decltype((double)v(i)) { return (double)v(i); }
template <class V>
auto elem(const V& v, int i, long) -> decltype((double)v[i]) { return (double)v[i]; }
template <class V>
double elem(const V& v, int i) { return elem(v, i, 0); }
template <class V>
auto set_elem(V& v, int i, double x, int) -> decltype((void)(v(i) = x)) { v(i) = x; }
template <class V>
auto set_elem(V& v, int i, double x, long) -> decltype((void)(v[i] = x)) { v[i] = x; }
template <class V>
void set_elem(V& v, int i, double x) { set_elem(v, i, x, 0); }

// Plain stand-ins for decomp_cvx_space::Polytope / FlightCorridor (data_type.h:124-245) where Eigen is absent:
// same member names, std::array instead of Eigen::Vector4d / Vector3d.
struct PlainPolytope {
  std::array<double, 3> center{{0, 0, 0}}, seed_coord{{0, 0, 0}};
  std::vector<std::array<double, 4>> planes;  // (a,b,c,d): outward normal, inside <=> ax+by+cz+d <= 0
  void appendPlane(const std::array<double, 4>& p) { planes.push_back(p); }
};
struct PlainCorridor {
  std::vector<double> durations;
  std::vector<PlainPolytope> polyhedrons;
  double scale_factor = 1.0;
  void appendPolytope(const PlainPolytope& p) { polyhedrons.push_back(p); }
  void appendTime(double t) { durations.push_back(t); }
  void clear() { durations.clear(); polyhedrons.clear(); }
  bool isEmpty() const { return polyhedrons.empty(); }
};

struct DenseMatrix {  // minimal row-major stand-in for Eigen::MatrixXd in the tests
  int r = 0, c = 0;
  std::vector<double> d;
  DenseMatrix() {}
  DenseMatrix(int rows, int cols) : r(rows), c(cols), d((size_t)rows * cols, 0.0) {}
  int rows() const { return r; }
  int cols() const { return c; }
  double& operator()(int i, int j) { return d[(size_t)i * c + j]; }
  double operator()(int i, int j) const { return d[(size_t)i * c + j]; }
};
struct DenseVector {
  std::vector<double> d;
  DenseVector() {}
  explicit DenseVector(int n) : d((size_t)n, 0.0) {}
  int size() const { return (int)d.size(); }
  double& operator()(int i) { return d[i]; }
  double operator()(int i) const { return d[i]; }
};

// One GPU handle shared by the optimiser objects of a process (the reference news/deletes an optimiser
// per call, teach_repeat_planner.cpp:853-854, 950-951; creating device buffers per call would be wasteful).
class DdpDevice {
 public:
  DdpDevice(int max_batch, int n_seg_max, int p_max, direct_dtype_t dtype = DIRECT_F64, int device = 0)
      : max_batch_(max_batch), n_seg_max_(n_seg_max), p_max_(p_max), dtype_(dtype) {
    direct_ddp_config_t cfg{(int32_t)dtype, device, max_batch, n_seg_max, p_max, 0};
    if (direct_ddp_create(&cfg, &h_) != DIRECT_OK) throw std::runtime_error(direct_ddp_last_error());
  }
  ~DdpDevice() { direct_ddp_destroy(h_); }
  DdpDevice(const DdpDevice&) = delete;
  DdpDevice& operator=(const DdpDevice&) = delete;
  direct_ddp_handle_t handle() const { return h_; }
  int max_batch() const { return max_batch_; }
  int n_seg_max() const { return n_seg_max_; }
  int p_max() const { return p_max_; }
  direct_dtype_t dtype() const { return dtype_; }

  // The process-wide device behind default-constructed optimisers (`new ddpTrajOptimizer()`, TRP:853-854):
  // created on first use with these sizes; call configure_shared() before that to change them.
  static DdpDevice& shared() {
    auto& slot = shared_slot();
    if (!slot) slot.reset(new DdpDevice(shared_cfg()[0], shared_cfg()[1], shared_cfg()[2], (direct_dtype_t)shared_cfg()[3], shared_cfg()[4]));
    return *slot;
  }
  static void configure_shared(int max_batch, int n_seg_max, int p_max, direct_dtype_t dtype = DIRECT_F64, int device = 0) {
    shared_slot().reset();
    shared_cfg() = {{max_batch, n_seg_max, p_max, (int)dtype, device}};
  }

 private:
  static std::unique_ptr<DdpDevice>& shared_slot() { static std::unique_ptr<DdpDevice> s; return s; }
  static std::array<int, 5>& shared_cfg() { static std::array<int, 5> c{{1, 128, DIRECT_P_LIMIT, (int)DIRECT_F64, 0}}  /* every polytope polyhedronGenerator can emit (128 planes); a knot only pays for the row slots its own polytope fills */; return c; }
  direct_ddp_handle_t h_ = nullptr;
  int max_batch_, n_seg_max_, p_max_;
  direct_dtype_t dtype_;
};

template <class Mat = DenseMatrix, class Vec = DenseVector>
class ddpTrajOptimizer {
 public:
  ddpTrajOptimizer() : dev_(DdpDevice::shared()) {}  // the reference's `new ddpTrajOptimizer()` (ddp_optimizer.h:264)
  explicit ddpTrajOptimizer(DdpDevice& dev) : dev_(dev) {}

  // The reference's call, argument for argument (ddp_optimizer.h:267-289).  Corridor = the caller's
  // decomp_cvx_space::FlightCorridor (or direct::PlainCorridor).
  template <class Corridor>
  int polyCurveGeneration(const Corridor& corridor, const Mat& MQM_u, const Mat& MQM_l, const Mat& pos, const Mat& vel,
                          const Mat& acc, const Mat& jer, const double minimize_order, const double max_vel,
                          const double max_acc, const double max_jer, Mat initbezCoeff, const double w_snap,
                          const double w_terminal, const double w_time, const int iter_max, bool& infeas,
                          bool zero_init_flag, bool line_init_flag, bool& line_failed, int time_power, bool minvo_flag) {
    std::vector<uint8_t> inf{(uint8_t)(infeas ? 1 : 0)}, lf{(uint8_t)(line_failed ? 1 : 0)};
    const std::vector<int> rtn = polyCurveGenerationBatch(
        std::vector<Corridor>{corridor}, MQM_u, MQM_l, std::vector<Mat>{pos}, std::vector<Mat>{vel}, std::vector<Mat>{acc},
        std::vector<Mat>{jer}, minimize_order, max_vel, max_acc, max_jer, std::vector<Mat>{std::move(initbezCoeff)}, w_snap,
        w_terminal, w_time, iter_max, inf, zero_init_flag, line_init_flag, lf, time_power, minvo_flag);
    infeas = inf[0] != 0;
    line_failed = lf[0] != 0;
    return rtn[0];
  }

  // The same for a batch.  pos/vel/acc/jer: one 2x3 matrix per corridor (row 0 start, row 1 goal); initbezCoeff:
  // one N x 18 matrix per corridor (ignored when zero_init_flag or line_init_flag); infeas / line_failed: in-out
  // per corridor (`bool&` of the reference).  Returns the per-corridor return codes.
  template <class Corridor>
  std::vector<int> polyCurveGenerationBatch(const std::vector<Corridor>& corridors,
                                            const Mat& /*MQM_u: unused by the reference, ddp_optimizer.cpp:7*/,
                                            const Mat& /*MQM_l: unused*/, const std::vector<Mat>& pos,
                                            const std::vector<Mat>& vel, const std::vector<Mat>& acc,
                                            const std::vector<Mat>& /*jer: only read when sys_order == 4*/,
                                            double /*minimize_order: unused*/, double max_vel, double max_acc,
                                            double /*max_jer: unused*/, const std::vector<Mat>& initbezCoeff, double w_snap,
                                            double w_terminal, double w_time, int iter_max, std::vector<uint8_t>& infeas,
                                            bool zero_init_flag, bool line_init_flag, std::vector<uint8_t>& line_failed,
                                            int time_power, bool minvo_flag) {
    const int B = (int)corridors.size(), nm = dev_.n_seg_max(), pm = dev_.p_max();
    if (B < 1 || B > dev_.max_batch()) throw std::invalid_argument("batch size");
    if ((int)infeas.size() != B || (int)line_failed.size() != B) throw std::invalid_argument("infeas / line_failed need one entry per corridor");
    if ((int)pos.size() != B || (int)vel.size() != B || (int)acc.size() != B) throw std::invalid_argument("pos / vel / acc need one matrix per corridor");
    const bool needs_bez = !zero_init_flag && !line_init_flag;
    if (needs_bez && (int)initbezCoeff.size() != B) throw std::invalid_argument("initbezCoeff needs one matrix per corridor");
    const bool f64 = dev_.dtype() == DIRECT_F64;
    n_seg_.assign(B, 0);
    std::vector<int32_t> n_planes((size_t)B * nm, 1);
    std::vector<double> x0((size_t)B * 9, 0.0), xd((size_t)B * 9, 0.0), T0((size_t)B * nm, 1.0);
    std::vector<double> planes((size_t)B * nm * pm * 4, 0.0), seeds((size_t)B * nm * 3, 0.0), bez((size_t)B * nm * 18, 0.0);
    for (int b = 0; b < B; b++) {
      const auto& cor = corridors[b];
      const int N = (int)cor.polyhedrons.size();
      if (N < 1 || N > nm || (int)cor.durations.size() < N) throw std::invalid_argument("corridor size");
      if (needs_bez && (initbezCoeff[b].rows() < N || initbezCoeff[b].cols() < 18)) throw std::invalid_argument("initbezCoeff must be N x 18");
      n_seg_[b] = N;
      for (int d = 0; d < 3; d++) {
        x0[(size_t)b * 9 + d] = pos[b](0, d); x0[(size_t)b * 9 + 3 + d] = vel[b](0, d); x0[(size_t)b * 9 + 6 + d] = acc[b](0, d);
        xd[(size_t)b * 9 + d] = pos[b](1, d); xd[(size_t)b * 9 + 3 + d] = vel[b](1, d); xd[(size_t)b * 9 + 6 + d] = acc[b](1, d);
      }
      for (int k = 0; k < N; k++) {
        const auto& pl = cor.polyhedrons[k];
        if ((int)pl.planes.size() < 1 || (int)pl.planes.size() > pm) throw std::invalid_argument("planes per polytope");
        n_planes[(size_t)b * nm + k] = (int32_t)pl.planes.size();
        T0[(size_t)b * nm + k] = cor.durations[k];
        for (size_t p = 0; p < pl.planes.size(); p++)
          for (int q = 0; q < 4; q++) planes[(((size_t)b * nm + k) * pm + p) * 4 + q] = elem(pl.planes[p], q);
        for (int d = 0; d < 3; d++) seeds[((size_t)b * nm + k) * 3 + d] = elem(pl.seed_coord, d);
        if (needs_bez)  // line-init builds its own start from the seeds (DDP:194-248)
          for (int q = 0; q < 18; q++) bez[((size_t)b * nm + k) * 18 + q] = initbezCoeff[b](k, q);
      }
    }
    direct_ddp_params_t p{max_vel, max_acc, w_snap, w_terminal, w_time, iter_max, time_power, zero_init_flag,
                          line_init_flag, minvo_flag, 1, 0, 0};
    // the C-ABI is typed by the handle's storage dtype: narrow once here when it is float
    std::vector<float> fx0, fxd, fT0, fpl, fsd, fbz;
    auto narrow = [](const std::vector<double>& a, std::vector<float>& o) { o.assign(a.begin(), a.end()); return (const void*)o.data(); };
    direct_ddp_batch_in_t in{};
    in.batch = B; in.n_seg_max = nm; in.p_max = pm; in.mem = DIRECT_MEM_HOST;
    in.n_seg = n_seg_.data(); in.n_planes = n_planes.data(); in.infeas_in = infeas.data();
    in.x0 = f64 ? (const void*)x0.data() : narrow(x0, fx0);
    in.xd = f64 ? (const void*)xd.data() : narrow(xd, fxd);
    in.T0 = f64 ? (const void*)T0.data() : narrow(T0, fT0);
    in.planes = f64 ? (const void*)planes.data() : narrow(planes, fpl);
    in.seeds = f64 ? (const void*)seeds.data() : narrow(seeds, fsd);
    in.init_bez = needs_bez ? (f64 ? (const void*)bez.data() : narrow(bez, fbz)) : nullptr;
    rtn_.assign(B, 0); iter_.assign(B, 0);
    std::vector<uint8_t> inf_out(B), lf_out(B);
    const size_t rs = f64 ? 8 : 4;
    raw_cost_.assign(B * rs, 0); raw_jerk_.assign(B * rs, 0); raw_tn_.assign(B * rs, 0);
    raw_bez_.assign((size_t)B * nm * 18 * rs, 0); raw_poly_.assign((size_t)B * nm * 18 * rs, 0); raw_T_.assign((size_t)B * nm * rs, 0);
    direct_ddp_batch_out_t out{};
    out.mem = DIRECT_MEM_HOST; out.rtn = rtn_.data(); out.iter_used = iter_.data();
    out.infeas_out = inf_out.data(); out.line_failed_out = lf_out.data();
    out.cost = raw_cost_.data(); out.jerk_cost = raw_jerk_.data(); out.terminal_norm2 = raw_tn_.data();
    out.bez = raw_bez_.data(); out.poly = raw_poly_.data(); out.T = raw_T_.data();
    const auto t0 = std::chrono::steady_clock::now();
    if (direct_ddp_solve_batch(dev_.handle(), &p, &in, &out) != DIRECT_OK) throw std::runtime_error(direct_ddp_last_error());
    compTime_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    infeas = inf_out;
    // `line_failed` is only ever WRITTEN by the line-init success exit, which sets it to false
    // (ddp_optimizer.cpp:384); every other path leaves the caller's value alone
    for (int b = 0; b < B; b++)
      if (line_init_flag && lf_out[b] == 0) line_failed[b] = 0;
    return std::vector<int>(rtn_.begin(), rtn_.end());
  }

  // getters (ddp_optimizer.h:299-340), per corridor b
  Mat getPolyCoeff(int b = 0) const { return rows18(raw_poly_, b); }
  Mat getBezCoeff(int b = 0) const { return rows18(raw_bez_, b); }
  Vec getPolyTime(int b = 0) const {
    Vec v(n_seg_[b]);
    for (int k = 0; k < n_seg_[b]; k++) v(k) = at(raw_T_, (size_t)b * dev_.n_seg_max() + k);
    return v;
  }
  double getDDPObjective(int b = 0) const { return at(raw_cost_, b); }
  double getCompTime() const { return compTime_; }
  double getTerminalNorm(int b = 0) const { return at(raw_tn_, b); }
  int getIterUsed(int b = 0) const { return iter_[b]; }
  double getJerkCost(int b = 0) const { return at(raw_jerk_, b); }

 private:
  double at(const std::vector<uint8_t>& raw, size_t i) const {
    return dev_.dtype() == DIRECT_F64 ? ((const double*)raw.data())[i] : (double)((const float*)raw.data())[i];
  }
  Mat rows18(const std::vector<uint8_t>& raw, int b) const {
    Mat m(n_seg_[b], 18);
    for (int k = 0; k < n_seg_[b]; k++)
      for (int q = 0; q < 18; q++) m(k, q) = at(raw, ((size_t)b * dev_.n_seg_max() + k) * 18 + q);
    return m;
  }
  DdpDevice& dev_;
  std::vector<int32_t> n_seg_, rtn_, iter_;
  std::vector<uint8_t> raw_cost_, raw_jerk_, raw_tn_, raw_bez_, raw_poly_, raw_T_;
  double compTime_ = 0.0;
};

// ---- the steps around the path, same names as the node's helpers ---------------------------------
// writeCorridorMsg / readCorridorMsg (teach_repeat_planner.cpp:354-410): msgs/corridor in ROS 1 wire
// serialisation <-> FlightCorridor.
template <class Corridor>
std::vector<uint8_t> writeCorridorMsg(int path_id, const Corridor& corridor) {
  const int N = (int)corridor.polyhedrons.size();
  int pm = 1;
  for (const auto& pl : corridor.polyhedrons) pm = std::max(pm, (int)pl.planes.size());
  std::vector<int32_t> n_planes(N);
  std::vector<double> planes((size_t)N * pm * 4, 0.0), seeds((size_t)N * 3), centers((size_t)N * 3);
  for (int k = 0; k < N; k++) {
    const auto& pl = corridor.polyhedrons[k];
    n_planes[k] = (int32_t)pl.planes.size();
    for (size_t j = 0; j < pl.planes.size(); j++)
      for (int q = 0; q < 4; q++) planes[((size_t)k * pm + j) * 4 + q] = elem(pl.planes[j], q);
    for (int d = 0; d < 3; d++) {
      seeds[k * 3 + d] = elem(pl.seed_coord, d);
      centers[k * 3 + d] = elem(pl.center, d);
    }
  }
  std::vector<uint8_t> buf(direct_corridor_wire_size(N, n_planes.data()));
  size_t written = 0;
  if (direct_corridor_pack(path_id, N, n_planes.data(), planes.data(), pm, seeds.data(), centers.data(), buf.data(),
                           buf.size(), &written) != DIRECT_OK)
    throw std::runtime_error(direct_ddp_last_error());
  buf.resize(written);
  return buf;
}
// Corridor needs polyhedrons (a vector of a Polytope with center, seed_coord and appendPlane) as in data_type.h
template <class Corridor>
void readCorridorMsg(const std::vector<uint8_t>& msg, Corridor& corridor, int& path_id,
                            int n_seg_max = 256, int p_max = DIRECT_P_LIMIT) {
  std::vector<int32_t> n_planes(n_seg_max);
  std::vector<double> planes((size_t)n_seg_max * p_max * 4), seeds((size_t)n_seg_max * 3), centers((size_t)n_seg_max * 3);
  int32_t pid = 0, N = 0;
  if (direct_corridor_unpack(msg.data(), msg.size(), n_seg_max, p_max, &pid, &N, n_planes.data(), planes.data(),
                             seeds.data(), centers.data(), nullptr) != DIRECT_OK)
    throw std::runtime_error(direct_ddp_last_error());
  corridor.polyhedrons.clear();
  for (int k = 0; k < N; k++) {
    typename std::decay<decltype(corridor.polyhedrons[0])>::type pl;
    typedef typename std::decay<decltype(pl.planes[0])>::type Plane;
    for (int d = 0; d < 3; d++) {
      set_elem(pl.center, d, centers[k * 3 + d]);
      set_elem(pl.seed_coord, d, seeds[k * 3 + d]);
    }
    for (int j = 0; j < n_planes[k]; j++) {
      const double* q = &planes[((size_t)k * p_max + j) * 4];
      Plane h;
      for (int c = 0; c < 4; c++) set_elem(h, c, q[c]);
      pl.appendPlane(h);
    }
    corridor.polyhedrons.push_back(pl);
  }
  path_id = pid;
}

// The sampling loop of visBezierTrajectory & co. (teach_repeat_planner.cpp:1551-1566) for one trajectory on
// the device: polyCoeff = getBezCoeff() (N x 18), time = getPolyTime(); returns the sampled positions and,
// through traj_len, the reference's polyline length.  Batched callers use direct_traj_sample_batch directly.
template <class Mat, class Vec>
std::vector<std::array<double, 3>> sampleBezierTrajectory(DdpDevice& dev, const Mat& polyCoeff, const Vec& time, double dt,
                                                        double* traj_len = nullptr) {
  const int N = (int)time.size(), nm = N;
  const bool f64 = dev.dtype() == DIRECT_F64;
  double total = 0.0;
  for (int k = 0; k < N; k++) total += time(k) > 0 ? time(k) : 0.0;
  const int cap = (int)(total / dt) + 2 * N + 8;
  std::vector<double> bez((size_t)nm * 18), T(nm), pos((size_t)cap * 3), len(1);
  for (int k = 0; k < N; k++) {
    T[k] = time(k);
    for (int q = 0; q < 18; q++) bez[(size_t)k * 18 + q] = polyCoeff(k, q);
  }
  std::vector<float> fb, fT, fpos, flen;
  int32_t n_seg = N, count = 0;
  direct_sample_in_t in{};
  in.batch = 1; in.n_seg_max = nm; in.capacity = cap; in.derivs = 0; in.mem = DIRECT_MEM_HOST; in.n_seg = &n_seg; in.dt = dt;
  direct_sample_out_t out{};
  out.count = &count;
  if (f64) {
    in.bez = bez.data(); in.T = T.data(); out.pos = pos.data(); out.length = len.data();
  } else {
    fb.assign(bez.begin(), bez.end()); fT.assign(T.begin(), T.end()); fpos.resize(pos.size()); flen.resize(1);
    in.bez = fb.data(); in.T = fT.data(); out.pos = fpos.data(); out.length = flen.data();
  }
  if (direct_traj_sample_batch(dev.handle(), &in, &out) != DIRECT_OK) throw std::runtime_error(direct_ddp_last_error());
  if (count < 0) return {};  // "time less than 0, returning" (TRP:1552-1555)
  std::vector<std::array<double, 3>> pts((size_t)std::min(count, cap));
  for (size_t i = 0; i < pts.size(); i++)
    for (int d = 0; d < 3; d++) pts[i][d] = f64 ? pos[i * 3 + d] : (double)fpos[i * 3 + d];
  if (traj_len) *traj_len = f64 ? len[0] : (double)flen[0];
  return pts;
}

}  // namespace direct
