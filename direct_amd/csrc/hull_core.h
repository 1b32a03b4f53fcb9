// Convex hull -> facet planes of a voxel cluster, exact integer arithmetic (include/direct_cluster.h,
// direct_cluster_hull_planes_batch).  Replaces, for a batch of clusters,
//   polyhedronGenerator::getConvexPoly   global_planner/src/utils/poly_utils.cpp:282-389  (point set, quickhull, snap)
//   Polyhedron::hrep (eigen-cdd)         :404-449 (call site)                               (V-rep -> A x <= b)
//   polyhedronGenerator::polyHrep2Utils  :127-206  (unit normals pointing outwards, half-voxel inflation, centre)
// The reference goes through two floating-point libraries (quickhull with an epsilon, cddlib).  Every point it feeds
// them lies on the half-voxel lattice q = 2 index + 1 (+/- 1 for the corners of a flat cluster), so here every
// predicate is an integer determinant and the result is the exact one the libraries approximate.
//
// Shape of the computation (no sequential hull construction, nothing data-dependent in the control flow between
// kernels; the phases are the kernels of direct_cluster.hip and, compiled for the host, tests/emu/hull_emu.cpp):
//   1 lines     a corner of the hull is the first or last cluster point of its x-, y- AND z-line: min / max per
//               line with atomics, then an ordered compaction -> a few hundred CANDIDATES out of thousands of voxels
//   2 edges     one thread per candidate pair (a, b): project the other candidates onto the plane perpendicular to
//               b - a and keep the wedge they span.  The pair is a hull edge iff the wedge stays below 180 degrees
//               (most pairs fail within a few points); its two bounding half-planes are facet planes.  Pairs with
//               a candidate strictly between a and b are skipped (the shorter pairs report the same planes).
//   3 finish    duplicate planes removed and ranked by (nx, ny, nz, K) of the primitive integer normal (O(m^2)
//               compares, m ~ 10^3), corners = candidates with hull edges in two non-parallel directions, unit
//               normals / offsets / centre in double.
#pragma once
#include <math.h>
#include <stdint.h>

#ifndef HULL_HD
#define HULL_HD __host__ __device__ __forceinline__
#endif

namespace hull {

typedef long long i64;

constexpr int kCandCap = 2048;     // candidates per cluster (24 KB of LDS in the edge kernel)
constexpr int kRawCap = 8192;      // facet planes reported by edges, duplicates included (a few hundred for real clusters;
                                   // bounds the O(m^2) duplicate removal of k_hull_finish)
constexpr int LINE_MIN_INIT = 0x7f7f7f7f, LINE_MAX_INIT = (int)0x80808080;  // byte patterns: hipMemset can write them

struct Lines {  // first / last point of every axis-parallel lattice line of one cluster (half-voxel lattice)
  int *xmin, *xmax;  // [QY * QZ]
  int *ymin, *ymax;  // [QX * QZ]
  int *zmin, *zmax;  // [QX * QY]
  int QX, QY, QZ;
};

HULL_HD i64 gcd(i64 a, i64 b) {
  a = a < 0 ? -a : a;
  b = b < 0 ? -b : b;
  while (b) {
    const i64 t = a % b;
    a = b;
    b = t;
  }
  return a;
}

// the lattice point of cluster element t (corner c of it when the cluster is flat): getVoxelVertex's loop order
// (:208-223: x, y, z over -1, +1, z fastest)
HULL_HD void lattice_point(int ix, int iy, int iz, int degenerate, int c, int& qx, int& qy, int& qz) {
  qx = 2 * ix + 1;
  qy = 2 * iy + 1;
  qz = 2 * iz + 1;
  if (degenerate) {
    qx += (c & 4) ? 1 : -1;
    qy += (c & 2) ? 1 : -1;
    qz += (c & 1) ? 1 : -1;
  }
}

HULL_HD bool line_extreme(const Lines& L, int qx, int qy, int qz) {
  const int ix = qy * L.QZ + qz, iy = qx * L.QZ + qz, iz = qx * L.QY + qy;
  return (qx == L.xmin[ix] || qx == L.xmax[ix]) && (qy == L.ymin[iy] || qy == L.ymax[iy]) && (qz == L.zmin[iz] || qz == L.zmax[iz]);
}

// Is the candidate pair (a, b) a hull edge with no candidate strictly between its ends?  1: yes, ir / il are the
// candidates that span the wedge (ir clockwise-most, il counter-clockwise-most, looking along b - a); 0: no;
// 2: every other candidate lies in ONE half-plane through the line (a flat point set).
template <typename CoordFn>
HULL_HD int edge_test(CoordFn P, int nc, int a, int b, int& ir, int& il) {
  int ax, ay, az, bx, by, bz;
  P(a, ax, ay, az);
  P(b, bx, by, bz);
  const i64 dx = bx - ax, dy = by - ay, dz = bz - az;
  if (!dx && !dy && !dz) return 0;  // the same lattice point twice (shared corners of a flat cluster)
  const i64 dd = dx * dx + dy * dy + dz * dz;
  // (e1, e2, d) right-handed and orthogonal: e1 = d x (axis of the smallest |d_i|), e2 = d x e1
  const i64 adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy, adz = dz < 0 ? -dz : dz;
  i64 e1x, e1y, e1z;
  if (adx <= ady && adx <= adz) { e1x = 0; e1y = dz; e1z = -dy; }        // d x (1, 0, 0)
  else if (ady <= adz)          { e1x = -dz; e1y = 0; e1z = dx; }        // d x (0, 1, 0)
  else                          { e1x = dy; e1y = -dx; e1z = 0; }        // d x (0, 0, 1)
  const i64 e2x = dy * e1z - dz * e1y, e2y = dz * e1x - dx * e1z, e2z = dx * e1y - dy * e1x;
  i64 Ru = 0, Rv = 0, Lu = 0, Lv = 0;
  ir = il = -1;
  for (int c = 0; c < nc; c++) {
    int cx, cy, cz;
    P(c, cx, cy, cz);
    const i64 wx = cx - ax, wy = cy - ay, wz = cz - az;
    const i64 u = wx * e1x + wy * e1y + wz * e1z, v = wx * e2x + wy * e2y + wz * e2z;
    if (!u && !v) {  // on the line through a and b
      const i64 t = wx * dx + wy * dy + wz * dz;
      if (t > 0 && t < dd) return 0;
      continue;
    }
    if (ir < 0) {
      Ru = Lu = u;
      Rv = Lv = v;
      ir = il = c;
      continue;
    }
    const i64 cR = Ru * v - Rv * u, cL = Lu * v - Lv * u;
    if (cR >= 0 && cL <= 0) {  // inside the wedge - or opposite to a wedge of zero width
      if (cR == 0 && cL == 0 && Ru * u + Rv * v < 0) return 0;
    } else if (cR > 0 && cL > 0) {  // beyond the counter-clockwise end, still within 180 degrees of the other end
      Lu = u;
      Lv = v;
      il = c;
    } else if (cR < 0 && cL < 0) {  // beyond the clockwise end
      if (u * Lv - v * Lu <= 0) return 0;
      Ru = u;
      Rv = v;
      ir = c;
    } else {
      return 0;  // on the far side of both ends: the candidates surround the line
    }
  }
  if (ir < 0) return 0;
  return (Ru * Lv - Rv * Lu == 0) ? 2 : 1;
}

// primitive outward plane through a, b, r (l lies strictly inside): n . q + K <= 0 for every point of the cluster
template <typename CoordFn>
HULL_HD void plane_through(CoordFn P, int a, int b, int r, int l, i64* out) {
  int ax, ay, az, bx, by, bz, rx, ry, rz, lx, ly, lz;
  P(a, ax, ay, az);
  P(b, bx, by, bz);
  P(r, rx, ry, rz);
  P(l, lx, ly, lz);
  const i64 dx = bx - ax, dy = by - ay, dz = bz - az, wx = rx - ax, wy = ry - ay, wz = rz - az;
  i64 nx = dy * wz - dz * wy, ny = dz * wx - dx * wz, nz = dx * wy - dy * wx;
  const i64 g = gcd(gcd(nx, ny), nz);
  nx /= g;
  ny /= g;
  nz /= g;
  if (nx * (lx - ax) + ny * (ly - ay) + nz * (lz - az) > 0) {
    nx = -nx;
    ny = -ny;
    nz = -nz;
  }
  out[0] = nx;
  out[1] = ny;
  out[2] = nz;
  out[3] = -(nx * ax + ny * ay + nz * az);
}

HULL_HD int plane_cmp(const i64* a, const i64* b) {
  for (int i = 0; i < 4; i++)
    if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return 0;
}

// unit normal and offset in world coordinates x = q res / 2 + lower; axis-aligned faces of a solid cluster move out by
// half a voxel (polyHrep2Utils :172-188).  Plain IEEE operations in this order, no contraction (the CPU checker does the same).
HULL_HD void plane_world(const i64* P, double res, const double* lower, int degenerate, double* out) {
#pragma clang fp contract(off)
  const double h = res * 0.5;
  const double L = sqrt((double)(P[0] * P[0] + P[1] * P[1] + P[2] * P[2]));
  double s = (double)P[0] * lower[0];
  s = s + (double)P[1] * lower[1];
  s = s + (double)P[2] * lower[2];
  double K = ((double)P[3] * h - s) / L;
  const bool axis = ((P[0] != 0) + (P[1] != 0) + (P[2] != 0)) == 1;
  if (!degenerate && axis) K = K - h;
  out[0] = (double)P[0] / L;
  out[1] = (double)P[1] / L;
  out[2] = (double)P[2] / L;
  out[3] = K;
}

// world coordinate of a lattice point: index2Coord (:20-41) for a voxel centre, round2Voxel (:225-234) for a corner
HULL_HD double world_coord(int q, double res, double lower, int degenerate) {
#pragma clang fp contract(off)
  return degenerate ? (double)q * res * 0.5 + lower : (double)((q - 1) / 2) * res + 0.5 * res + lower;
}

}  // namespace hull
