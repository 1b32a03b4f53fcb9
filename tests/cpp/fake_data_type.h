// TEST-ONLY header SHAPED LIKE global_planner/include/global_planner/utils/data_type.h:123-245 of the reference: the
// same namespace, struct and member names and the same access syntax (planes are 4-vectors read with operator()),
// on top of a few-line Eigen look-alike, because the build container has no Eigen.  It exists so that the C++
// mirror (direct_amd/host/ddp_optimizer.hpp) and the INTEGRATION.md snippet can be compiled in ONE translation unit
// next to types named decomp_cvx_space::Polytope / FlightCorridor, as they will be inside teach_repeat_planner.cpp.
// Nothing here is copied from the reference (its structs carry more members and methods); it is not product code.
#pragma once
#include <cstddef>
#include <vector>

namespace Eigen {
template <int N>
struct FixedVec {
  double d[N] = {};
  double& operator()(int i) { return d[i]; }
  double operator()(int i) const { return d[i]; }
};
typedef FixedVec<3> Vector3d;
typedef FixedVec<4> Vector4d;
struct MatrixXd {
  int r = 0, c = 0;
  std::vector<double> d;
  MatrixXd() {}
  MatrixXd(int rows, int cols) : r(rows), c(cols), d((std::size_t)rows * cols, 0.0) {}
  int rows() const { return r; }
  int cols() const { return c; }
  double& operator()(int i, int j) { return d[(std::size_t)i * c + j]; }
  double operator()(int i, int j) const { return d[(std::size_t)i * c + j]; }
};
struct VectorXd {
  std::vector<double> d;
  VectorXd() {}
  explicit VectorXd(int n) : d((std::size_t)n, 0.0) {}
  int size() const { return (int)d.size(); }
  double& operator()(int i) { return d[i]; }
  double operator()(int i) const { return d[i]; }
};
}  // namespace Eigen

namespace decomp_cvx_space {
struct Polytope {
  Eigen::Vector3d center, seed_coord;
  std::vector<Eigen::Vector4d> planes;
  void appendPlane(Eigen::Vector4d plane) { planes.push_back(plane); }
};
struct FlightCorridor {
  std::vector<double> durations;
  std::vector<Polytope> polyhedrons;
  double scale_factor = 1.0;
  void appendPolytope(Polytope pltp) { polyhedrons.push_back(pltp); }
  void appendTime(double t) { durations.push_back(t); }
  void clear() { durations.clear(); polyhedrons.clear(); }
};
}  // namespace decomp_cvx_space
