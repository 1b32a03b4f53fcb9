// TEST INFRASTRUCTURE.  C entry point for the REFERENCE's convex hull (global_planner/third_party/quickhull, the
// library polyhedronGenerator::getConvexPoly calls at global_planner/src/utils/poly_utils.cpp:340 and :371), compiled
// together with the reference's own QuickHull.cpp (from where it lies under /root/reference, never copied) into
// oracle/_ref/libquickhull_ref.so by `make -C oracle _ref`.  The call below is the reference's call:
// getConvexHull(points, true, false) followed by getVertexBuffer(); the index buffer (three vertex indices per
// triangle) is returned as well so that the facet planes can be pinned.
#include <vector>

#include "quickhull/QuickHull.hpp"

extern "C" int ref_quickhull(int n, const double* pts, int vcap, double* verts, int* n_verts, int icap, int* tri, int* n_idx) {
  std::vector<quickhull::Vector3<double>> cloud;
  cloud.reserve(n);
  for (int i = 0; i < n; i++) cloud.emplace_back(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
  quickhull::QuickHull<double> qh;
  auto hull = qh.getConvexHull(cloud, true, false);
  auto& vb = hull.getVertexBuffer();
  auto& ib = hull.getIndexBuffer();
  *n_verts = (int)vb.size();
  *n_idx = (int)ib.size();
  if ((int)vb.size() > vcap || (int)ib.size() > icap) return 1;
  int o = 0;
  for (const auto& v : vb) { verts[3 * o] = v.x; verts[3 * o + 1] = v.y; verts[3 * o + 2] = v.z; o++; }
  for (size_t i = 0; i < ib.size(); i++) tri[i] = (int)ib[i];
  return 0;
}
