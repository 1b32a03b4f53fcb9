/*
 * hull_ref.c -- CPU restatement of the hull -> H-rep step of the corridor generator.  TEST INFRASTRUCTURE: only
 * tests/, __graft_entry__.smoke() and bench legs may use it; nothing under direct_amd/ does.
 *
 * What it restates (global_planner/src/utils/poly_utils.cpp):
 *   getConvexPoly          :282-389  cluster voxels -> point set (voxel centres; the 8 corners of every voxel when the
 *                                    cluster is flat along an axis, checkDegeneratePoly :236-273, getVoxelVertex :208-223)
 *                                    -> convex hull (third_party/quickhull) -> vertex buffer snapped to the lattice
 *   Polyhedron::hrep       :404-449  (call site) V-rep -> H-rep A x <= b through eigen-cdd / cddlib
 *   polyHrep2Utils         :127-206  rows normalised to unit normals pointing outwards, plane = (n, K) with
 *                                    n . x + K <= 0 inside, axis-aligned faces moved out by half a voxel, centre = mean
 *                                    of one vertex per plane
 *
 * cddlib is not in this image and quickhull is floating point with an epsilon, so the restatement does not follow
 * their code paths: it states the RESULT they approximate, exactly.  All points lie on the half-voxel lattice
 * (q = 2 index + 1 for a voxel centre, q = 2 index + 1 +/- 1 for a corner), so every predicate is integer arithmetic:
 * a plane is a facet plane iff it passes through three non-collinear points and no point lies strictly outside it.
 * PINNING: tests/test_hull.py checks this file against the reference's own quickhull built from its sources into
 * oracle/_ref/libquickhull_ref.so - every triangle quickhull returns lies in one of the facet planes found here, every
 * facet plane found here carries one of its triangles, and its vertex buffer is the vertex set found here.  What cannot
 * be pinned (cdd's row ORDER and its rounding noise) is stated in DESIGN.md 7d: rows are sorted by (nx, ny, nz, K) of
 * the primitive integer normal, and the per-plane vertex of polyHrep2Utils (argmin of residuals that are all ~1e-16)
 * is the first vertex ON the plane.
 *
 * Method (independent of the device's): brute force over triples with an early exit, after dropping every voxel
 * that is the midpoint of two others of the set along an axis (it cannot be a vertex and cannot lie outside a plane
 * that bounds the rest).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef long long i64;

static i64 gcd64(i64 a, i64 b) {
  if (a < 0) a = -a;
  if (b < 0) b = -b;
  while (b) { i64 t = a % b; a = b; b = t; }
  return a;
}

static int cmp_plane(const void* pa, const void* pb) {
  const i64* a = (const i64*)pa; const i64* b = (const i64*)pb;
  for (int i = 0; i < 4; i++) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return 0;
}

static int cmp_pt(const void* pa, const void* pb) {
  const int* a = (const int*)pa; const int* b = (const int*)pb;
  for (int i = 0; i < 3; i++) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return 0;
}

/* is p in the sorted point list? */
static int has_pt(const int* sorted, int n, int x, int y, int z) {
  int key[3] = {x, y, z};
  return bsearch(key, sorted, (size_t)n, 3 * sizeof(int), cmp_pt) != NULL;
}

/* return: 0 ok, 1 a capacity was exceeded, 3 the points do not span three dimensions */
int hull_ref(int n, const int32_t* idx, double res, const double* lower, int plane_cap, int64_t* plane_int,
             double* planes, int* n_planes, int vert_cap, int32_t* vert_q, double* vertices, int* n_vertices,
             double* center, int* degenerate) {
  *n_planes = 0; *n_vertices = 0;
  if (n <= 0) return 3;
  /* checkDegeneratePoly (:236-273): all voxels share x, or y, or z */
  int deg = 0;
  for (int a = 0; a < 3; a++) {
    int same = 1;
    for (int i = 0; i + 1 < n; i++) if (idx[3 * i + a] != idx[3 * (i + 1) + a]) { same = 0; break; }
    deg |= same;
  }
  *degenerate = deg;
  /* the point set on the half-voxel lattice, in the reference's order (:306-318, :208-223: x, y, z loops over -1, +1) */
  int m = deg ? 8 * n : n;
  int* Q = (int*)malloc((size_t)m * 3 * sizeof(int));
  if (deg) {
    int o = 0;
    for (int i = 0; i < n; i++)
      for (int x = -1; x < 2; x += 2) for (int y = -1; y < 2; y += 2) for (int z = -1; z < 2; z += 2) {
        Q[3 * o] = 2 * idx[3 * i] + 1 + x; Q[3 * o + 1] = 2 * idx[3 * i + 1] + 1 + y; Q[3 * o + 2] = 2 * idx[3 * i + 2] + 1 + z;
        o++;
      }
  } else {
    for (int i = 0; i < 3 * n; i++) Q[i] = 2 * idx[i] + 1;
  }
  /* sorted copy for membership tests; first occurrences only */
  int* S = (int*)malloc((size_t)m * 3 * sizeof(int));
  memcpy(S, Q, (size_t)m * 3 * sizeof(int));
  qsort(S, (size_t)m, 3 * sizeof(int), cmp_pt);
  int ms = 0;
  for (int i = 0; i < m; i++) if (i == 0 || cmp_pt(&S[3 * i], &S[3 * (i - 1)])) { memmove(&S[3 * ms], &S[3 * i], 3 * sizeof(int)); ms++; }
  /* working set W: distinct points (first occurrences, in order) that are not an axis midpoint of two others */
  int* W = (int*)malloc((size_t)m * 3 * sizeof(int));
  int w = 0;
  for (int i = 0; i < m; i++) {
    const int x = Q[3 * i], y = Q[3 * i + 1], z = Q[3 * i + 2];
    int dup = 0;
    for (int j = 0; j < w && !dup; j++) dup = (W[3 * j] == x && W[3 * j + 1] == y && W[3 * j + 2] == z);
    if (dup) continue;
    const int st = 2;  /* neighbouring centres (odd q) and neighbouring corners (even q) are both two lattice units apart */
    if ((has_pt(S, ms, x - st, y, z) && has_pt(S, ms, x + st, y, z)) || (has_pt(S, ms, x, y - st, z) && has_pt(S, ms, x, y + st, z)) ||
        (has_pt(S, ms, x, y, z - st) && has_pt(S, ms, x, y, z + st))) continue;
    W[3 * w] = x; W[3 * w + 1] = y; W[3 * w + 2] = z; w++;
  }
  /* facet planes: every non-collinear triple whose plane has no point strictly on one of its sides */
  int pc = 0, pcap = 1024;
  i64* PL = (i64*)malloc((size_t)pcap * 4 * sizeof(i64));
  for (int i = 0; i < w; i++) for (int j = i + 1; j < w; j++) {
    const i64 ax = W[3 * j] - W[3 * i], ay = W[3 * j + 1] - W[3 * i + 1], az = W[3 * j + 2] - W[3 * i + 2];
    for (int k = j + 1; k < w; k++) {
      const i64 bx = W[3 * k] - W[3 * i], by = W[3 * k + 1] - W[3 * i + 1], bz = W[3 * k + 2] - W[3 * i + 2];
      i64 nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
      if (!nx && !ny && !nz) continue;
      int pos = 0, neg = 0;
      for (int t = 0; t < w && !(pos && neg); t++) {
        const i64 d = nx * (W[3 * t] - W[3 * i]) + ny * (W[3 * t + 1] - W[3 * i + 1]) + nz * (W[3 * t + 2] - W[3 * i + 2]);
        pos |= d > 0; neg |= d < 0;
      }
      if (pos && neg) continue;
      if (!pos && !neg) { free(Q); free(S); free(W); free(PL); return 3; }  /* everything in one plane */
      const i64 g = gcd64(gcd64(nx, ny), nz);
      nx /= g; ny /= g; nz /= g;
      if (pos) { nx = -nx; ny = -ny; nz = -nz; }  /* outward: n . (p - p_i) <= 0 for every point */
      const i64 K = -(nx * W[3 * i] + ny * W[3 * i + 1] + nz * W[3 * i + 2]);
      int seen = 0;
      for (int t = 0; t < pc && !seen; t++) seen = (PL[4 * t] == nx && PL[4 * t + 1] == ny && PL[4 * t + 2] == nz && PL[4 * t + 3] == K);
      if (seen) continue;
      if (pc == pcap) { pcap *= 2; PL = (i64*)realloc(PL, (size_t)pcap * 4 * sizeof(i64)); }
      PL[4 * pc] = nx; PL[4 * pc + 1] = ny; PL[4 * pc + 2] = nz; PL[4 * pc + 3] = K; pc++;
    }
  }
  int rc = 0;
  if (pc == 0) { free(Q); free(S); free(W); free(PL); return 3; }
  qsort(PL, (size_t)pc, 4 * sizeof(i64), cmp_plane);
  /* vertices: points that lie on three facet planes with linearly independent normals, in order of appearance */
  int vc = 0;
  int* VQ = (int*)malloc((size_t)w * 3 * sizeof(int));
  for (int i = 0; i < w; i++) {
    int on[64], no = 0;
    for (int t = 0; t < pc && no < 64; t++)
      if (PL[4 * t] * W[3 * i] + PL[4 * t + 1] * W[3 * i + 1] + PL[4 * t + 2] * W[3 * i + 2] + PL[4 * t + 3] == 0) on[no++] = t;
    int is_v = 0;
    for (int a = 0; a < no && !is_v; a++) for (int b = a + 1; b < no && !is_v; b++) for (int c = b + 1; c < no && !is_v; c++) {
      const i64* A = &PL[4 * on[a]]; const i64* Bp = &PL[4 * on[b]]; const i64* Cp = &PL[4 * on[c]];
      const i64 det = A[0] * (Bp[1] * Cp[2] - Bp[2] * Cp[1]) - A[1] * (Bp[0] * Cp[2] - Bp[2] * Cp[0]) + A[2] * (Bp[0] * Cp[1] - Bp[1] * Cp[0]);
      is_v = det != 0;
    }
    if (is_v) { VQ[3 * vc] = W[3 * i]; VQ[3 * vc + 1] = W[3 * i + 1]; VQ[3 * vc + 2] = W[3 * i + 2]; vc++; }
  }
  /* outputs */
  const double h = res * 0.5;
  *n_planes = pc; *n_vertices = vc;
  if (pc > plane_cap || vc > vert_cap) rc = 1;
  for (int i = 0; i < vc && i < vert_cap; i++)
    for (int a = 0; a < 3; a++) {
      vert_q[3 * i + a] = VQ[3 * i + a];
      /* index2Coord (:20-41) for a voxel centre: index * res + 0.5 * res + lower; round2Voxel (:225-234) for a corner */
      vertices[3 * i + a] = deg ? (double)VQ[3 * i + a] * res * 0.5 + lower[a]
                                : (double)((VQ[3 * i + a] - 1) / 2) * res + 0.5 * res + lower[a];
    }
  double cs[3] = {0, 0, 0};
  for (int t = 0; t < pc; t++) {
    const i64* P = &PL[4 * t];
    if (t < plane_cap) {
      const double L = sqrt((double)(P[0] * P[0] + P[1] * P[1] + P[2] * P[2]));
      double s = (double)P[0] * lower[0];
      s = s + (double)P[1] * lower[1];
      s = s + (double)P[2] * lower[2];
      double K = ((double)P[3] * h - s) / L;
      const int axis = ((P[0] != 0) + (P[1] != 0) + (P[2] != 0)) == 1;
      if (!deg && axis) K = K - h;  /* :172-188 */
      for (int a = 0; a < 3; a++) { plane_int[4 * t + a] = (int64_t)P[a]; planes[4 * t + a] = (double)P[a] / L; }
      plane_int[4 * t + 3] = (int64_t)P[3];
      planes[4 * t + 3] = K;
    }
    /* polyHrep2Utils' per-plane vertex (:95-125): the first vertex on the plane */
    for (int i = 0; i < vc; i++)
      if (P[0] * VQ[3 * i] + P[1] * VQ[3 * i + 1] + P[2] * VQ[3 * i + 2] + P[3] == 0) {
        for (int a = 0; a < 3; a++)
          cs[a] = cs[a] + (deg ? (double)VQ[3 * i + a] * res * 0.5 + lower[a] : (double)((VQ[3 * i + a] - 1) / 2) * res + 0.5 * res + lower[a]);
        break;
      }
  }
  for (int a = 0; a < 3; a++) center[a] = cs[a] / (double)pc;
  free(Q); free(S); free(W); free(PL); free(VQ);
  return rc;
}
