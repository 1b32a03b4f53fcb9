// TEST-ONLY host build of direct_amd/csrc/hull_core.h: the phases of direct_cluster_hull_planes_batch run serially,
// one "thread" after the other, on the SAME predicates and formulas the kernels use (hull::lattice_point,
// line_extreme, edge_test, plane_through, plane_cmp, plane_world, world_coord).  Lets the CPU test suite check the
// device algorithm against the oracle (oracle/hull_ref.c) where there is no GPU; the kernels' own glue (atomics,
// compaction) is covered by tests/test_gpu_hull.py.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>
#define HULL_HD inline
#include "../../direct_amd/csrc/hull_core.h"

extern "C" int hull_emu(int n, const int32_t* idx, double res, const double* lower, int plane_cap, int64_t* plane_int,
                        double* planes, int* n_planes, int vert_cap, int32_t* vert_q, double* vertices, int* n_vertices,
                        double* center, int* degenerate, int* n_cand_out) {
  using hull::i64;
  *n_planes = 0; *n_vertices = 0;
  if (n <= 0) return 3;
  int mx[3] = {0, 0, 0};
  int diff[3] = {0, 0, 0};
  for (int t = 0; t < n; t++)
    for (int a = 0; a < 3; a++) { mx[a] = std::max(mx[a], idx[3 * t + a]); diff[a] |= idx[3 * t + a] != idx[a]; }
  const int deg = (!diff[0] || !diff[1] || !diff[2]) ? 1 : 0;
  *degenerate = deg;
  hull::Lines L;
  L.QX = 2 * mx[0] + 3; L.QY = 2 * mx[1] + 3; L.QZ = 2 * mx[2] + 3;
  std::vector<int> xmin(L.QY * L.QZ, hull::LINE_MIN_INIT), xmax(L.QY * L.QZ, hull::LINE_MAX_INIT);
  std::vector<int> ymin(L.QX * L.QZ, hull::LINE_MIN_INIT), ymax(L.QX * L.QZ, hull::LINE_MAX_INIT);
  std::vector<int> zmin(L.QX * L.QY, hull::LINE_MIN_INIT), zmax(L.QX * L.QY, hull::LINE_MAX_INIT);
  L.xmin = xmin.data(); L.xmax = xmax.data(); L.ymin = ymin.data(); L.ymax = ymax.data(); L.zmin = zmin.data(); L.zmax = zmax.data();
  const int m = deg ? 8 * n : n;
  for (int t = 0; t < m; t++) {  // k_hull_lines
    const int32_t* p = idx + 3 * (deg ? t >> 3 : t);
    int qx, qy, qz;
    hull::lattice_point(p[0], p[1], p[2], deg, t & 7, qx, qy, qz);
    const int ix = qy * L.QZ + qz, iy = qx * L.QZ + qz, iz = qx * L.QY + qy;
    xmin[ix] = std::min(xmin[ix], qx); xmax[ix] = std::max(xmax[ix], qx);
    ymin[iy] = std::min(ymin[iy], qy); ymax[iy] = std::max(ymax[iy], qy);
    zmin[iz] = std::min(zmin[iz], qz); zmax[iz] = std::max(zmax[iz], qz);
  }
  std::vector<int> cx, cy, cz;  // k_hull_cand
  int overflow = 0;
  for (int t = 0; t < m; t++) {
    const int32_t* p = idx + 3 * (deg ? t >> 3 : t);
    int qx, qy, qz;
    hull::lattice_point(p[0], p[1], p[2], deg, t & 7, qx, qy, qz);
    if (!hull::line_extreme(L, qx, qy, qz)) continue;
    if ((int)cx.size() < hull::kCandCap) { cx.push_back(qx); cy.push_back(qy); cz.push_back(qz); }
    else overflow = 1;
  }
  const int nc = (int)cx.size();
  if (n_cand_out) *n_cand_out = nc;
  auto P = [&](int i, int& x, int& y, int& z) { x = cx[i]; y = cy[i]; z = cz[i]; };
  std::vector<int> first(nc, -1), isv(nc, 0);
  std::vector<i64> raw;
  int flat = 0;
  for (int a = 0; a < nc; a++)  // k_hull_edges
    for (int b = a + 1; b < nc; b++) {
      int ir, il;
      const int r = hull::edge_test(P, nc, a, b, ir, il);
      if (r == 2) flat = 1;
      if (r != 1) continue;
      if ((int)raw.size() / 4 + 2 <= hull::kRawCap) {
        i64 pl[4];
        hull::plane_through(P, a, b, ir, il, pl); raw.insert(raw.end(), pl, pl + 4);
        hull::plane_through(P, a, b, il, ir, pl); raw.insert(raw.end(), pl, pl + 4);
      } else overflow = 1;
      for (int side = 0; side < 2; side++) {
        const int v = side ? b : a, o = side ? a : b;
        const int old = first[v];
        if (old < 0) { first[v] = o; continue; }
        if (old == o) continue;
        const i64 ux = cx[o] - cx[v], uy = cy[o] - cy[v], uz = cz[o] - cz[v];
        const i64 wx = cx[old] - cx[v], wy = cy[old] - cy[v], wz = cz[old] - cz[v];
        if (uy * wz - uz * wy != 0 || uz * wx - ux * wz != 0 || ux * wy - uy * wx != 0) isv[v] = 1;
      }
    }
  if (overflow) return 1;
  const int mr = (int)raw.size() / 4;
  if (flat || mr == 0) return 3;
  std::vector<unsigned char> uniq(mr);  // k_hull_finish
  for (int i = 0; i < mr; i++) {
    int f = 1;
    for (int j = 0; j < i && f; j++) f = hull::plane_cmp(&raw[4 * i], &raw[4 * j]) != 0;
    uniq[i] = (unsigned char)f;
  }
  std::vector<i64> sorted(4 * (size_t)mr);
  int np = 0;
  for (int i = 0; i < mr; i++) {
    if (!uniq[i]) continue;
    int rank = 0;
    for (int j = 0; j < mr; j++) rank += (uniq[j] && hull::plane_cmp(&raw[4 * j], &raw[4 * i]) < 0) ? 1 : 0;
    for (int c = 0; c < 4; c++) sorted[4 * (size_t)rank + c] = raw[4 * (size_t)i + c];
    np++;
  }
  for (int t = 0; t < np && t < plane_cap; t++) {
    for (int c = 0; c < 4; c++) plane_int[4 * t + c] = sorted[4 * (size_t)t + c];
    hull::plane_world(&sorted[4 * (size_t)t], res, lower, deg, planes + 4 * t);
  }
  std::vector<int> vq;
  for (int t = 0; t < nc; t++) {
    if (!isv[t]) continue;
    int keep = 1;
    for (int j = 0; j < t && keep; j++) keep = !(cx[j] == cx[t] && cy[j] == cy[t] && cz[j] == cz[t]);
    if (!keep) continue;
    const int off = (int)vq.size() / 3;
    vq.push_back(cx[t]); vq.push_back(cy[t]); vq.push_back(cz[t]);
    if (off < vert_cap) {
      for (int a = 0; a < 3; a++) vert_q[3 * off + a] = vq[3 * off + a];
      vertices[3 * off] = hull::world_coord(cx[t], res, lower[0], deg);
      vertices[3 * off + 1] = hull::world_coord(cy[t], res, lower[1], deg);
      vertices[3 * off + 2] = hull::world_coord(cz[t], res, lower[2], deg);
    }
  }
  const int nv = (int)vq.size() / 3;
  double cs[3] = {0, 0, 0};
  for (int t = 0; t < np; t++) {
    const i64* Pl = &sorted[4 * (size_t)t];
    for (int i = 0; i < nv; i++)
      if (Pl[0] * vq[3 * i] + Pl[1] * vq[3 * i + 1] + Pl[2] * vq[3 * i + 2] + Pl[3] == 0) {
        for (int a = 0; a < 3; a++) cs[a] = cs[a] + hull::world_coord(vq[3 * i + a], res, lower[a], deg);
        break;
      }
  }
  for (int a = 0; a < 3; a++) center[a] = cs[a] / (double)np;
  *n_planes = np; *n_vertices = nv;
  return (np > plane_cap || nv > vert_cap) ? 1 : 0;
}
