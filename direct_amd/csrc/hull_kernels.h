// Kernels of direct_cluster_hull_planes_batch (the phases of hull_core.h).  Included by direct_cluster.hip inside its
// anonymous namespace, after Dev / Elem / pack3.
//
//   k_hull_src     cluster sizes (and caller-provided voxels packed into the handle's cluster storage)
//   k_hull_lines   flat-cluster test (checkDegeneratePoly), first / last point of every lattice line (atomicMin / Max)
//   k_hull_cand    ordered compaction of the line-extreme points: the candidates, in cluster order
//   k_hull_edges   one thread per candidate pair (hull::edge_test); facet planes and corner evidence of the edges found
//   k_hull_finish  duplicate planes out, rank by key, doubles, corners, centre
// Integer work on a few hundred points per cluster; the voxels themselves (thousands per cluster) are read twice, by
// k_hull_lines and k_hull_cand, and never leave the device.

struct HullElem {
  int n, degenerate, n_cand, n_raw, flat, overflow, pad0, pad1;
};

struct HullDev {
  HullElem* he;    // [batch]
  int* lines;      // [batch][line_words]: xmin | ymin | zmin | xmax | ymax | zmax
  size_t line_words, half_words;
  int QX, QY, QZ;
  int* cand;       // [batch][3][kCandCap]
  int* first;      // [batch][kCandCap] first edge partner of a candidate (-1: none yet)
  int* isv;        // [batch][kCandCap] has edges in two non-parallel directions
  hull::i64* raw;  // [batch][kRawCap][4]
  hull::i64* sorted;  // [batch][kRawCap][4] distinct planes in key order
  int* vq;         // [batch][kCandCap][3] corners
};

__device__ __forceinline__ hull::Lines hull_lines_of(const HullDev& H, int e) {
  hull::Lines L;
  int* base = H.lines + (size_t)e * H.line_words;
  L.QX = H.QX; L.QY = H.QY; L.QZ = H.QZ;
  L.xmin = base;
  L.ymin = L.xmin + H.QY * H.QZ;
  L.zmin = L.ymin + H.QX * H.QZ;
  L.xmax = base + H.half_words;
  L.ymax = L.xmax + H.QY * H.QZ;
  L.zmax = L.ymax + H.QX * H.QZ;
  return L;
}

__global__ void k_hull_src(Dev D, HullDev H, int batch, const int32_t* xyz /* or null */, const int32_t* num) {
  const int e = blockIdx.y;
  int n, bad = 0;
  if (xyz) {
    n = num[e];
    n = n < 0 ? 0 : (n > D.ccap ? D.ccap : n);
    // caller-provided voxels are validated like direct_cluster_convex_test validates its own: an index outside the map
    // would address lattice lines outside the per-cluster line arrays (and one above 1023 would alias another axis in
    // pack3).  One bad voxel disqualifies the cluster: rtn = DIRECT_HULL_BAD_VOXEL, nothing is computed for it.
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
      const int32_t* p = xyz + ((size_t)e * D.ccap + t) * 3;
      const int x = p[0], y = p[1], z = p[2];
      const bool ok = x >= 0 && x < D.max_x && y >= 0 && y < D.max_y && z >= 0 && z < D.max_z;
      bad |= ok ? 0 : 1;
      D.cluster[(size_t)e * D.ccap + t] = ok ? pack3(x, y, z) : 0;
    }
    if (bad) atomicOr(&H.he[e].pad0, 1);  // pad0 / overflow were zeroed by the host before this launch
  } else {
    // resident clusters: only a generation that ended well left a usable cluster (an overflowed one is a truncated
    // prefix, DIRECT_CLUSTER_OVERFLOW: "not usable"); the count is clamped to the storage whatever the element says
    const int rt = D.el[e].rtn;
    n = rt == DIRECT_CLUSTER_OK ? D.el[e].n_cluster : 0;
    n = n < 0 ? 0 : (n > D.ccap ? D.ccap : n);
    if (rt == DIRECT_CLUSTER_OVERFLOW && blockIdx.x == 0 && threadIdx.x == 0) H.he[e].overflow = 1;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) H.he[e].n = n;
  (void)batch;
}

__global__ __launch_bounds__(256) void k_hull_lines(Dev D, HullDev H) {
  const int e = blockIdx.x, tid = threadIdx.x;
  const int n = H.he[e].n;
  if (n <= 0 || H.he[e].pad0) return;  // pad0: a caller-provided voxel outside the map (k_hull_src)
  const int* cl = D.cluster + (size_t)e * D.ccap;
  // checkDegeneratePoly (poly_utils.cpp:236-273): all voxels share x, or y, or z
  const int p0 = cl[0];
  int dx = 0, dy = 0, dz = 0;
  for (int t = tid; t < n; t += 256) {
    const int p = cl[t];
    dx |= px(p) != px(p0);
    dy |= py(p) != py(p0);
    dz |= pz(p) != pz(p0);
  }
  dx = __syncthreads_or(dx);
  dy = __syncthreads_or(dy);
  dz = __syncthreads_or(dz);
  const int deg = (!dx || !dy || !dz) ? 1 : 0;
  if (tid == 0) H.he[e].degenerate = deg;
  const hull::Lines L = hull_lines_of(H, e);
  const int m = deg ? 8 * n : n;
  for (int t = tid; t < m; t += 256) {
    const int p = cl[deg ? t >> 3 : t];
    int qx, qy, qz;
    hull::lattice_point(px(p), py(p), pz(p), deg, t & 7, qx, qy, qz);
    const int ix = qy * L.QZ + qz, iy = qx * L.QZ + qz, iz = qx * L.QY + qy;
    atomicMin(&L.xmin[ix], qx); atomicMax(&L.xmax[ix], qx);
    atomicMin(&L.ymin[iy], qy); atomicMax(&L.ymax[iy], qy);
    atomicMin(&L.zmin[iz], qz); atomicMax(&L.zmax[iz], qz);
  }
}

__global__ __launch_bounds__(256) void k_hull_cand(Dev D, HullDev H) {
  __shared__ int wsum[4];
  __shared__ int base_s;
  const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  HullElem* E = &H.he[e];
  const int n = E->n;
  if (n <= 0 || E->pad0) return;
  const int deg = E->degenerate;
  const int* cl = D.cluster + (size_t)e * D.ccap;
  const hull::Lines L = hull_lines_of(H, e);
  int* cx = H.cand + (size_t)e * 3 * hull::kCandCap;
  int* cy = cx + hull::kCandCap;
  int* cz = cy + hull::kCandCap;
  const int m = deg ? 8 * n : n;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int b0 = 0; b0 < m; b0 += 256) {  // ordered compaction: candidates keep the cluster's order
    const int t = b0 + tid;
    int qx = 0, qy = 0, qz = 0, keep = 0;
    if (t < m) {
      const int p = cl[deg ? t >> 3 : t];
      hull::lattice_point(px(p), py(p), pz(p), deg, t & 7, qx, qy, qz);
      keep = hull::line_extreme(L, qx, qy, qz) ? 1 : 0;
    }
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) wsum[wv] = __popcll(bal);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wv; w++) off += wsum[w];
    off += __popcll(bal & ((1ull << lane) - 1ull));
    if (keep && off < hull::kCandCap) {
      cx[off] = qx; cy[off] = qy; cz[off] = qz;
    }
    __syncthreads();
    if (tid == 0) base_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (tid == 0) {
    const int nc = base_s;
    E->n_cand = nc < hull::kCandCap ? nc : hull::kCandCap;
    if (nc > hull::kCandCap) E->overflow = 1;
  }
  const int nc = base_s < hull::kCandCap ? base_s : hull::kCandCap;
  for (int t = tid; t < nc; t += 256) {
    H.first[(size_t)e * hull::kCandCap + t] = -1;
    H.isv[(size_t)e * hull::kCandCap + t] = 0;
  }
}

__global__ __launch_bounds__(256) void k_hull_edges(HullDev H) {
  __shared__ int sx[hull::kCandCap], sy[hull::kCandCap], sz[hull::kCandCap];
  const int e = blockIdx.y, tid = threadIdx.x;
  HullElem* E = &H.he[e];
  const int nc = E->n_cand;
  if (nc < 2) return;
  const int* cx = H.cand + (size_t)e * 3 * hull::kCandCap;
  for (int t = tid; t < nc; t += 256) {
    sx[t] = cx[t];
    sy[t] = cx[hull::kCandCap + t];
    sz[t] = cx[2 * hull::kCandCap + t];
  }
  __syncthreads();
  auto P = [&](int i, int& x, int& y, int& z) { x = sx[i]; y = sy[i]; z = sz[i]; };
  int* first = H.first + (size_t)e * hull::kCandCap;
  int* isv = H.isv + (size_t)e * hull::kCandCap;
  hull::i64* raw = H.raw + (size_t)e * hull::kRawCap * 4;
  const long long pairs = (long long)nc * nc;
  for (long long p = (long long)blockIdx.x * 256 + tid; p < pairs; p += (long long)gridDim.x * 256) {
    const int a = (int)(p / nc), b = (int)(p - (long long)a * nc);
    if (a >= b) continue;
    int ir, il;
    const int r = hull::edge_test(P, nc, a, b, ir, il);
    if (r == 2) E->flat = 1;
    if (r != 1) continue;
    const int slot = atomicAdd(&E->n_raw, 2);
    if (slot + 2 <= hull::kRawCap) {
      hull::plane_through(P, a, b, ir, il, raw + (size_t)slot * 4);
      hull::plane_through(P, a, b, il, ir, raw + (size_t)(slot + 1) * 4);
    } else {
      E->overflow = 1;
    }
    // a corner has hull edges in two non-parallel directions; a point in the middle of a hull edge only along it
    for (int side = 0; side < 2; side++) {
      const int v = side ? b : a, o = side ? a : b;
      const int old = atomicCAS(&first[v], -1, o);
      if (old >= 0 && old != o) {
        const hull::i64 ux = sx[o] - sx[v], uy = sy[o] - sy[v], uz = sz[o] - sz[v];
        const hull::i64 wx = sx[old] - sx[v], wy = sy[old] - sy[v], wz = sz[old] - sz[v];
        if (uy * wz - uz * wy != 0 || uz * wx - ux * wz != 0 || ux * wy - uy * wx != 0) isv[v] = 1;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_hull_finish(HullDev H, double res, double lx, double ly, double lz, int plane_cap, int vert_cap,
                                                     double* planes, long long* plane_int, int32_t* n_planes, double* vertices,
                                                     int32_t* n_vertices, double* center, int32_t* degenerate, int32_t* rtn) {
  __shared__ __attribute__((aligned(8))) unsigned char uniq[hull::kRawCap];
  __shared__ int wsum[4];
  __shared__ int base_s, np_s;
  const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const HullElem E = H.he[e];
  const double lower[3] = {lx, ly, lz};
  int code = DIRECT_HULL_OK;
  if (E.n <= 0 || E.flat || E.n_raw == 0) code = DIRECT_HULL_FLAT;
  if (E.overflow) code = DIRECT_HULL_OVERFLOW;
  if (E.pad0) code = DIRECT_HULL_BAD_VOXEL;
  if (degenerate && tid == 0) degenerate[e] = E.degenerate;
  if (code != DIRECT_HULL_OK) {
    if (tid == 0) {
      if (rtn) rtn[e] = code;
      if (n_planes) n_planes[e] = 0;
      if (n_vertices) n_vertices[e] = 0;
    }
    return;
  }
  const int m = E.n_raw;
  const hull::i64* raw = H.raw + (size_t)e * hull::kRawCap * 4;
  hull::i64* sorted = H.sorted + (size_t)e * hull::kRawCap * 4;
  // 1: first occurrence of every distinct plane
  if (tid == 0) np_s = 0;
  for (int i = tid; i < m; i += 256) {
    int first = 1;
    for (int j = 0; j < i && first; j++) first = hull::plane_cmp(raw + 4 * (size_t)i, raw + 4 * (size_t)j) != 0;
    uniq[i] = (unsigned char)first;
  }
  __syncthreads();
  // 2: rank among the distinct planes by key
  int mine = 0;
  for (int i = tid; i < m; i += 256) {
    if (!uniq[i]) continue;
    int rank = 0;
    for (int j = 0; j < m; j++) rank += (uniq[j] && hull::plane_cmp(raw + 4 * (size_t)j, raw + 4 * (size_t)i) < 0) ? 1 : 0;
    for (int c = 0; c < 4; c++) sorted[4 * (size_t)rank + c] = raw[4 * (size_t)i + c];
    mine++;
  }
  atomicAdd(&np_s, mine);
  __syncthreads();
  const int np = np_s;
  for (int t = tid; t < np && t < plane_cap; t += 256) {
    if (plane_int)
      for (int c = 0; c < 4; c++) plane_int[((size_t)e * plane_cap + t) * 4 + c] = sorted[4 * (size_t)t + c];
    if (planes) hull::plane_world(sorted + 4 * (size_t)t, res, lower, E.degenerate, planes + ((size_t)e * plane_cap + t) * 4);
  }
  // 3: corners, in candidate order, first occurrence of a lattice point only
  const int nc = E.n_cand;
  const int* cx = H.cand + (size_t)e * 3 * hull::kCandCap;
  const int* cy = cx + hull::kCandCap;
  const int* cz = cy + hull::kCandCap;
  const int* isv = H.isv + (size_t)e * hull::kCandCap;
  int* vq = H.vq + (size_t)e * hull::kCandCap * 3;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nc; b0 += 256) {
    const int t = b0 + tid;
    int keep = 0;
    if (t < nc && isv[t]) {
      keep = 1;
      for (int j = 0; j < t && keep; j++) keep = !(cx[j] == cx[t] && cy[j] == cy[t] && cz[j] == cz[t]);
    }
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) wsum[wv] = __popcll(bal);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wv; w++) off += wsum[w];
    off += __popcll(bal & ((1ull << lane) - 1ull));
    if (keep) {
      vq[3 * off] = cx[t]; vq[3 * off + 1] = cy[t]; vq[3 * off + 2] = cz[t];
      if (vertices && off < vert_cap) {
        double* o = vertices + ((size_t)e * vert_cap + off) * 3;
        o[0] = hull::world_coord(cx[t], res, lx, E.degenerate);
        o[1] = hull::world_coord(cy[t], res, ly, E.degenerate);
        o[2] = hull::world_coord(cz[t], res, lz, E.degenerate);
      }
    }
    __syncthreads();
    if (tid == 0) base_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  const int nv = base_s;
  __threadfence_block();
  // 4: centre = mean over the planes of the first corner on the plane (polyHrep2Utils :129-144 with getVerticesPlane
  // :95-125, whose argmin over residuals of ~1e-16 is some corner ON the plane), summed in plane order by one thread
  int* fv = (int*)uniq;  // the flags are dead: kRawCap bytes hold kRawCap / 4 indices
  for (int t = tid; t < np; t += 256) {
    const hull::i64* P = sorted + 4 * (size_t)t;
    int f = -1;
    for (int i = 0; i < nv && f < 0; i++)
      if (P[0] * vq[3 * i] + P[1] * vq[3 * i + 1] + P[2] * vq[3 * i + 2] + P[3] == 0) f = i;
    if (t < hull::kRawCap / 4) fv[t] = f;
  }
  __syncthreads();
  if (tid == 0) {
    if (center && np <= hull::kRawCap / 4) {
#pragma clang fp contract(off)
      double cs[3] = {0, 0, 0};
      for (int t = 0; t < np; t++) {
        const int f = fv[t];
        if (f < 0) continue;
        for (int a = 0; a < 3; a++) cs[a] = cs[a] + hull::world_coord(vq[3 * f + a], res, lower[a], E.degenerate);
      }
      for (int a = 0; a < 3; a++) center[(size_t)e * 3 + a] = cs[a] / (double)np;
    }
    if (n_planes) n_planes[e] = np;
    if (n_vertices) n_vertices[e] = nv;
    // a capacity only counts for an output the caller asked for
    const bool over = ((planes || plane_int) && np > plane_cap) || (vertices && nv > vert_cap) ||
                      (center && np > hull::kRawCap / 4);  // the centre needs one scratch index per plane (2048)
    if (rtn) rtn[e] = over ? DIRECT_HULL_OVERFLOW : DIRECT_HULL_OK;
  }
}
