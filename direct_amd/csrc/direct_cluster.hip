// Corridor-cluster generation on gfx950 (include/direct_cluster.h).
//
// Replaces cudaPolytopeGeneration::polygonGeneration of the reference's SHIPPED CPU build
// (polyhedron_generator/src/cluster_server_cpu.cpp:394-528, "CS") for a batch of seed voxels, bit for bit:
//   cubeInflation_cpu    CS:257-293  -> k_inflate   (one workgroup per seed runs ALL rounds and directions on the
//                                       device; the reference's CUDA twin paraCubeInflation, cluster_engine.cu:185-350,
//                                       pays one H2D + launch + D2H per direction per round and reduces through a
//                                       racy single bool; here a face is reduced with __syncthreads_or)
//   getVoxelsInCube + surface extraction CS:62-81, 441-506 -> k_inflate (ordered block-scan compaction)
//   candidate generation CS:301-353  -> k_mark + k_compact (first-discoverer order reproduced with an atomicMin
//                                       of the discovery key per voxel, then an ordered compaction)
//   serialConvexTest     cluster_engine_cpu.cpp:31-136 -> k_convex (one workgroup per candidate, one lane per
//                                       target ray; float DDA exactly as the reference's; a candidate blocked by a
//                                       cluster voxel stops at once)
//   the sequential accept loop CS:360-384 -> k_resolve (one wave; bit-matrix of candidate-candidate rays, the
//                                       device form of paraResultCheck's packed triangle, cluster_engine.cu:37-67)
// Not in the reference: a ray whose axis-aligned box holds no obstacle cannot be blocked (summed-area table of the map,
// k_sat_*; per ray and per chunk of 256 cluster voxels, k_chunk_box) - the results are the reference's, most walks are not run.
// Integer / byte work, HBM- and L2-bound: no MFMA anywhere.  Flag bytes of a seed's grid: bit0 use_data,
// bit1 invalid_data, bit2 inside_data, bit3 map_data == 1 (one load per DDA step instead of two).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/direct_cluster.h"
#include "hull_core.h"

namespace {

constexpr uint8_t F_USE = 1, F_INVALID = 2, F_INSIDE = 4, F_OBS = 8;
constexpr int KEY_NONE = 0x7f7f7f7f;
constexpr int kDimLimit = 1024;  // voxels are packed as x << 20 | y << 10 | z

__device__ __forceinline__ int pack3(int x, int y, int z) { return (x << 20) | (y << 10) | z; }
__device__ __forceinline__ int px(int p) { return p >> 20; }
__device__ __forceinline__ int py(int p) { return (p >> 10) & 1023; }
__device__ __forceinline__ int pz(int p) { return p & 1023; }

struct Elem {  // per-seed counters, device resident
  int n_cluster, n_active, n_cand, iters, live, rtn, pad0, pad1;
  int vertex[24];
};

constexpr int kCompactBlocks = 64;  // workgroups per seed of the candidate compaction
struct Dev {
  int max_x, max_y, max_z, max_yz, G;
  int ccap, kcap, kwords;  // cluster / candidate capacity, 64-bit words per candidate row
  const uint8_t* map;      // [G]
  int* sat;                // [(max_x + 1)(max_y + 1)(max_z + 1)] obstacles in [0, x) x [0, y) x [0, z): summed-area table of map == 1
  int sat_yz, sat_z;       // its strides
  const float* inv;        // [kDimLimit] (float)(1.0 / (double)d), inv[0] = 0: the DDA's tDelta per |d| (ray_walk_lin)
  int* cbox;               // [batch][nchunk][6] bounding box (lo xyz, hi xyz) of every 256 consecutive cluster voxels
  int nchunk;              // (ccap + 255) / 256
  uint8_t* flags;          // [batch][G]
  int* key;                // [batch][G]
  int *cluster, *active;   // [batch][ccap] packed voxels
  int* cand;               // [batch][kcap]
  uint8_t *can_clu, *accept;        // [batch][kcap]
  unsigned long long* blocked;      // [batch][kcap][kwords]
  unsigned long long* accbits;      // [batch][kwords] accepted candidates of the round (k_resolve_fast -> k_apply)
  int* accpre;                      // [batch][kwords] accepted candidates before word w
  Elem* el;                // [batch]
  int* ccnt;               // [batch][kCompactBlocks] first-discoverer counts of k_compact_count's ranges
  int* csnap;              // [batch][2] (live, n_active * 26) as the round's candidate generation saw them
};

// ---- flags <- map ---------------------------------------------------------------------------------------
__global__ void k_flags_init(Dev D, const uint8_t* inside /* or null */) {
  const size_t e = blockIdx.y;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)D.G; i += (size_t)gridDim.x * blockDim.x) {
    uint8_t f = D.map[i] == 1 ? F_OBS : 0;
    if (inside && inside[i] == 1) f |= F_INSIDE;
    D.flags[e * D.G + i] = f;
    D.key[e * D.G + i] = KEY_NONE;
  }
}

// ---- summed-area table of the obstacles (once per map) ----------------------------------------------------
// A ray's walk only visits voxels of the axis-aligned box spanned by its two ends (every axis makes exactly |d| steps,
// see ray_walk_t), and it reports "blocked" only at a voxel with map == 1: a ray whose box holds no obstacle is clear
// whatever else happens on the way.  In free space that is 85 - 97 % of the rays that pass the two cheap tests, and a
// box sum is eight loads against ~25 dependent steps.
__global__ void k_sat_fill(Dev D) {
  const int n = (D.max_x + 1) * D.sat_yz;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int x = i / D.sat_yz, r = i - x * D.sat_yz, y = r / D.sat_z, z = r - y * D.sat_z;
    D.sat[i] = (x > 0 && y > 0 && z > 0 && D.map[(x - 1) * D.max_yz + (y - 1) * D.max_z + (z - 1)] == 1) ? 1 : 0;
  }
}
__global__ void k_sat_scan(Dev D, int axis) {  // running sums along one axis, one thread per line
  const int nx = D.max_x + 1, ny = D.max_y + 1, nz = D.max_z + 1;
  const int lines = axis == 0 ? ny * nz : (axis == 1 ? nx * nz : nx * ny);
  const int len = axis == 0 ? nx : (axis == 1 ? ny : nz);
  const int stride = axis == 0 ? D.sat_yz : (axis == 1 ? D.sat_z : 1);
  for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < lines; l += gridDim.x * blockDim.x) {
    int base;
    if (axis == 0) base = l;                                          // (y, z)
    else if (axis == 1) base = (l / nz) * D.sat_yz + (l % nz);        // (x, z)
    else base = l * nz;                                               // (x, y)
    int acc = 0;
    for (int k = 0; k < len; k++) {
      acc += D.sat[base + k * stride];
      D.sat[base + k * stride] = acc;
    }
  }
}
// obstacles in the box [lo, hi] (inclusive voxel coordinates)
__device__ __forceinline__ int box_obstacles(const Dev& D, int x0, int y0, int z0, int x1, int y1, int z1) {
  // unsigned BYTE offsets from the table's (wave-uniform) base: one 32-bit add per corner, no 64-bit address arithmetic
  const char* S = (const char*)D.sat;
  const unsigned a0 = (unsigned)(x0 * D.sat_yz) << 2, a1 = (unsigned)((x1 + 1) * D.sat_yz) << 2, b0 = (unsigned)(y0 * D.sat_z) << 2,
                 b1 = (unsigned)((y1 + 1) * D.sat_z) << 2, c0 = (unsigned)z0 << 2, c1 = (unsigned)(z1 + 1) << 2;
  auto at = [&](unsigned off) { return *(const int*)(S + off); };
  const unsigned a0b0 = a0 + b0, a0b1 = a0 + b1, a1b0 = a1 + b0, a1b1 = a1 + b1;
  return at(a1b1 + c1) - at(a0b1 + c1) - at(a1b0 + c1) - at(a1b1 + c0) + at(a0b0 + c1) + at(a0b1 + c0) + at(a1b0 + c0) - at(a0b0 + c0);
}

// exclusive position of `flag` among the flags of a 256-thread block + the block total (ordered compaction)
__device__ __forceinline__ int block_scan256(int flag, int* total, int* wsum /* shared[4] */) {
  const unsigned long long m = __ballot(flag);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int pos = __popcll(m & ((1ull << lane) - 1ull));
  __syncthreads();
  if (lane == 0) wsum[w] = __popcll(m);
  __syncthreads();
  int base = 0, tot = 0;
  for (int q = 0; q < 4; q++) {
    if (q < w) base += wsum[q];
    tot += wsum[q];
  }
  *total = tot;
  return base + pos;
}

// ---- cube inflation + initial surface cluster (CS:257-293, 400-518) ---------------------------------------
__global__ __launch_bounds__(256) void k_inflate(Dev D, const int* seeds, int itr_inflate_max) {
  __shared__ int v[24], vl[24], wsum[4];
  const int e = blockIdx.x, tid = threadIdx.x;
  Elem* E = &D.el[e];
  uint8_t* fl = D.flags + (size_t)e * D.G;
  const int sx = seeds[3 * e], sy = seeds[3 * e + 1], sz = seeds[3 * e + 2];
  if (sx < 0 || sx >= D.max_x || sy < 0 || sy >= D.max_y || sz < 0 || sz >= D.max_z) {
    if (tid == 0) {
      E->n_cluster = E->n_active = E->n_cand = E->iters = E->live = 0;
      E->rtn = DIRECT_CLUSTER_BAD_SEED;
      for (int q = 0; q < 24; q++) E->vertex[q] = 0;
    }
    return;
  }
  if (tid < 8) { v[tid] = vl[tid] = sx; v[tid + 8] = vl[tid + 8] = sy; v[tid + 16] = vl[tid + 16] = sz; }  // CS:400-408
  __syncthreads();
  const int step = 1;  // inf_step, CS:117
  for (int iter = 0; iter < itr_inflate_max; iter++) {
    for (int dir = 0; dir < 6; dir++) {  // Y-, Y+, X-, X+, Z-, Z+ (CS:263-268)
      // the face slab one voxel beyond the cube: ranges (a, b) and the fixed coordinate, per CS:126-255
      int a0, a1, b0, b1, fixed, edge;
      switch (dir) {
        case 0: edge = v[8] == 0;            fixed = v[8] - 1;  a0 = v[3]; a1 = v[0]; b0 = v[20]; b1 = v[16]; break;
        case 1: edge = v[9] == D.max_y - 1;  fixed = v[9] + 1;  a0 = v[2]; a1 = v[1]; b0 = v[21]; b1 = v[17]; break;
        case 2: edge = v[3] == 0;            fixed = v[3] - 1;  a0 = v[11]; a1 = v[10]; b0 = v[23]; b1 = v[19]; break;
        case 3: edge = v[0] == D.max_x - 1;  fixed = v[0] + 1;  a0 = v[8]; a1 = v[9]; b0 = v[20]; b1 = v[16]; break;
        case 4: edge = v[20] == 0;           fixed = v[20] - 1; a0 = v[7]; a1 = v[4]; b0 = v[12]; b1 = v[13]; break;
        default: edge = v[16] == D.max_z - 1; fixed = v[16] + 1; a0 = v[3]; a1 = v[0]; b0 = v[8]; b1 = v[9]; break;
      }
      int hit = 0;
      if (!edge) {
        const int nb = b1 - b0 + 1, cells = (a1 - a0 + 1) * nb;
        for (int c = tid; c < cells; c += 256) {
          const int a = a0 + c / nb, b = b0 + c % nb;
          int id;
          if (dir < 2) id = a * D.max_yz + fixed * D.max_z + b;        // (x, fixed y, z)
          else if (dir < 4) id = fixed * D.max_yz + a * D.max_z + b;   // (fixed x, y, z)
          else id = a * D.max_yz + b * D.max_z + fixed;                // (x, y, fixed z)
          hit |= D.map[id] > 0;
        }
      }
      const int blocked = __syncthreads_or(hit);
      if (tid == 0 && !edge && !blocked) {
        switch (dir) {
          case 0: v[8] -= step; v[11] -= step; v[12] -= step; v[15] -= step; break;
          case 1: v[9] += step; v[10] += step; v[13] += step; v[14] += step; break;
          case 2: v[2] -= step; v[3] -= step; v[6] -= step; v[7] -= step; break;
          case 3: v[0] += step; v[1] += step; v[4] += step; v[5] += step; break;
          case 4: v[20] -= step; v[21] -= step; v[22] -= step; v[23] -= step; break;
          default: v[16] += step; v[17] += step; v[18] += step; v[19] += step; break;
        }
      }
      __syncthreads();
    }
    const int changed = __syncthreads_or(tid < 24 ? (v[tid] != vl[tid]) : 0);  // CS:270-281
    if (!changed) break;
    if (tid < 24) vl[tid] = v[tid];
    __syncthreads();
  }
  if (tid < 24) E->vertex[tid] = v[tid];
  // getVoxelsInCube (CS:62-81) and the surface cells (CS:441-489), in x, y, z order
  const int x0 = v[7], x1 = v[1], y0 = v[15], y1 = v[9], z0 = v[23], z1 = v[17];
  const int nx = x1 - x0 + 1, ny = y1 - y0 + 1, nz = z1 - z0 + 1;
  const long ncube = (long)nx * ny * nz;
  int* cl = D.cluster + (size_t)e * D.ccap;
  int* ac = D.active + (size_t)e * D.ccap;
  int n_out = 0, overflow = 0;
  if (ncube == 1) {  // CS:433-439: the cell is its own surface; use_data is NOT set on this branch
    if (tid == 0) cl[0] = ac[0] = pack3(x0, y0, z0);
    n_out = 1;
  } else {
    for (long base = 0; base < ncube; base += 256) {
      const long c = base + tid;
      int surf = 0, p = 0;
      if (c < ncube) {
        const int x = x0 + (int)(c / ((long)ny * nz)), r = (int)(c % ((long)ny * nz)), y = y0 + r / nz, z = z0 + r % nz;
        p = pack3(x, y, z);
        // a cell is interior iff all 26 neighbours lie in the cube (and hence in the map): strictly inside the box
        surf = x == x0 || x == x1 || y == y0 || y == y1 || z == z0 || z == z1;
        const int id = x * D.max_yz + y * D.max_z + z;
        fl[id] = (fl[id] & F_OBS) | F_USE | (surf ? 0 : F_INSIDE);  // CS:447, 76, 491-494
      }
      int tot;
      const int pos = block_scan256(surf, &tot, wsum);
      if (surf) {
        if (n_out + pos < D.ccap) cl[n_out + pos] = ac[n_out + pos] = p;
        else overflow = 1;
      }
      n_out += tot;
      __syncthreads();
    }
  }
  overflow = __syncthreads_or(overflow);
  if (tid == 0) {
    const int degenerate = x0 == x1 || y0 == y1 || z0 == z1;  // CS:509-518
    E->n_cluster = E->n_active = overflow ? D.ccap : n_out;
    E->n_cand = 0;
    E->iters = 0;
    E->rtn = overflow ? DIRECT_CLUSTER_OVERFLOW : DIRECT_CLUSTER_OK;
    E->live = (!overflow && !degenerate) ? 1 : 0;
  }
}

// ---- candidate generation (CS:301-353) -------------------------------------------------------------------
// discovery key of (active voxel a, neighbour n): the order in which the sequential loop would reach it
__device__ __forceinline__ int neighbour_id(const Dev& D, int p, int n, int* packed) {
  const int q = n < 13 ? n : n + 1;  // 27 offsets without the centre, dx outermost (CS:315-319)
  const int x = px(p) + q / 9 - 1, y = py(p) + (q / 3) % 3 - 1, z = pz(p) + q % 3 - 1;
  if (x < 0 || x > D.max_x - 1 || y < 0 || y > D.max_y - 1 || z < 0 || z > D.max_z - 1) return -1;
  *packed = pack3(x, y, z);
  return x * D.max_yz + y * D.max_z + z;
}
__global__ void k_mark(Dev D) {
  const int e = blockIdx.y;
  const Elem* E = &D.el[e];
  if (!E->live) return;
  const int total = E->n_active * 26;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {  // grid stride: the launch
                                                                                                 // does not depend on n_active
    int p;
    const int id = neighbour_id(D, D.active[(size_t)e * D.ccap + t / 26], t % 26, &p);
    // map == 1 || use || invalid || inside -> skip (CS:332-338): all four are bits of the one flag byte
    if (id >= 0 && D.flags[(size_t)e * D.G + id] == 0) atomicMin(&D.key[(size_t)e * D.G + id], t);
  }
}
// Ordered compaction of the first discoverers into the candidate list, kCompactBlocks workgroups per seed (one
// workgroup walking all n_active x 26 neighbours was 84 us per round, 8 % of the generation): every workgroup owns a
// contiguous range of the discovery order; k_compact_count counts its first discoverers, k_compact_write places them
// behind those of the ranges before it.  The Elem is only written by k_compact_write's workgroup 0, and nobody in that
// kernel reads it (csnap carries what the round started from).
__device__ __forceinline__ void compact_range(int total, int b, int* lo, int* hi) {
  const int per = (((total + kCompactBlocks - 1) / kCompactBlocks) + 255) & ~255;
  *lo = b * per;
  *hi = *lo + per < total ? *lo + per : total;
}
__global__ __launch_bounds__(256) void k_compact_count(Dev D) {
  __shared__ int wsum[4];
  const int e = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
  const Elem* E = &D.el[e];
  const int live = E->live, total = live ? E->n_active * 26 : 0;
  if (b == 0 && tid == 0) { D.csnap[2 * e] = live; D.csnap[2 * e + 1] = total; }
  int lo, hi, n = 0;
  compact_range(total, b, &lo, &hi);
  for (int base = lo; base < hi; base += 256) {
    const int t = base + tid;
    int first = 0, p = 0;
    if (t < hi) {
      const int id = neighbour_id(D, D.active[(size_t)e * D.ccap + t / 26], t % 26, &p);
      first = id >= 0 && D.key[(size_t)e * D.G + id] == t;
    }
    n += __popcll(__ballot(first));
  }
  if ((tid & 63) == 0) wsum[tid >> 6] = n;
  __syncthreads();
  if (tid == 0) D.ccnt[e * kCompactBlocks + b] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(256) void k_compact_write(Dev D) {
  __shared__ int wsum[4];
  __shared__ int s_off, s_all;
  const int e = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
  if (!D.csnap[2 * e]) return;
  const int total = D.csnap[2 * e + 1];
  if (tid < 64) {  // kCompactBlocks == 64: one count per lane
    const int c = D.ccnt[e * kCompactBlocks + tid];
    int before = tid < b ? c : 0, all = c;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { before += __shfl_xor(before, d); all += __shfl_xor(all, d); }
    if (tid == 0) { s_off = before; s_all = all; }
  }
  __syncthreads();
  int n_out = s_off;
  const int all = s_all;
  int lo, hi;
  compact_range(total, b, &lo, &hi);
  for (int base = lo; base < hi; base += 256) {
    const int t = base + tid;
    int first = 0, p = 0, id = -1;
    if (t < hi) {
      id = neighbour_id(D, D.active[(size_t)e * D.ccap + t / 26], t % 26, &p);
      first = id >= 0 && D.key[(size_t)e * D.G + id] == t;
    }
    int tot;
    const int pos = block_scan256(first, &tot, wsum);
    if (first) {
      D.key[(size_t)e * D.G + id] = KEY_NONE;
      D.flags[(size_t)e * D.G + id] |= F_USE;  // CS:347
      if (n_out + pos < D.kcap) D.cand[(size_t)e * D.kcap + n_out + pos] = p;
    }
    n_out += tot;
    __syncthreads();
  }
  if (b == 0 && tid == 0) {
    Elem* E = &D.el[e];
    const int overflow = all > D.kcap;
    E->n_cand = overflow ? 0 : all;
    if (overflow) { E->rtn = DIRECT_CLUSTER_OVERFLOW; E->live = 0; }
    else if (all == 0) E->live = 0;  // CS:356-357
  }
}

// ---- serialConvexTest, one ray (cluster_engine_cpu.cpp:44-131).  Returns 1 when an obstacle blocks it. -------
__device__ __forceinline__ float intbound_half(int ds) {
  // intbound_cpu(0.5, ds), cluster_engine_cpu.cpp:13-29: mod(+-0.5, 1) is 0.5 for either sign, so the value is
  // 0.5f / |ds| in float; computed in double and rounded once, which is the correctly rounded float quotient
  if (ds == 0) return INFINITY;  // numeric_limits<double>::max() returned through float
  return (float)(0.5 / (double)(ds < 0 ? -ds : ds));
}
// RD(x, y, z) -> flag byte of a voxel (only F_INSIDE and F_OBS are looked at)
// the two cheap tests in front of the walk: 1 when the ray has to be traced
template <typename RD>
__device__ __forceinline__ int ray_needs_walk_t(const RD& rd, int cx, int cy, int cz, int target) {
  const int ex = px(target), ey = py(target), ez = pz(target);
  if (rd(ex, ey, ez) & F_INSIDE) return 0;  // only targets with inside_data == 0 are traced
  const int mx = cx / 2 + (ex >> 1), my = cy / 2 + (ey >> 1), mz = cz / 2 + (ez >> 1);
  if (rd(mx, my, mz) & F_INSIDE) return 0;  // "midpoint" inside the cube: skipped
  return 1;
}
template <typename RD>
__device__ __forceinline__ int ray_walk_t(const RD& rd, int cx, int cy, int cz, int target);
template <typename RD>
__device__ __forceinline__ int ray_blocked_t(const RD& rd, int cx, int cy, int cz, int target) {
  return ray_needs_walk_t(rd, cx, cy, cz, target) ? ray_walk_t(rd, cx, cy, cz, target) : 0;
}
template <typename RD>
__device__ __forceinline__ int ray_walk_t(const RD& rd, int cx, int cy, int cz, int target) {
  const int ex = px(target), ey = py(target), ez = pz(target);
  int x = cx, y = cy, z = cz;
  const int dx = ex - x, dy = ey - y, dz = ez - z;
  const int sx = (dx > 0) - (dx < 0), sy = (dy > 0) - (dy < 0), sz = (dz > 0) - (dz < 0);
  float tMaxX = intbound_half(dx), tMaxY = intbound_half(dy), tMaxZ = intbound_half(dz);
  // tDelta = ((float)step) / d: 1 / |d| correctly rounded, NaN (0 / 0) for d == 0 - never added: tMax stays +inf
  const float tDX = dx ? (float)(1.0 / (double)(dx < 0 ? -dx : dx)) : 0.0f;
  const float tDY = dy ? (float)(1.0 / (double)(dy < 0 ? -dy : dy)) : 0.0f;
  const float tDZ = dz ? (float)(1.0 / (double)(dz < 0 ? -dz : dz)) : 0.0f;
  int budget = (dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy) + (dz < 0 ? -dz : dz);
  for (;;) {
    if (tMaxX < tMaxY) {
      if (tMaxX < tMaxZ) { x += sx; tMaxX += tDX; }
      else               { z += sz; tMaxZ += tDZ; }
    } else {
      if (tMaxY < tMaxZ) { y += sy; tMaxY += tDY; }
      else               { z += sz; tMaxZ += tDZ; }
    }
    if (x == ex && y == ey && z == ez) return 0;
    // each axis makes exactly |d| steps before the end voxel is reached; a ray that float rounding made miss its
    // end would leave the map (the reference then reads outside its arrays): cut off, reported as clear
    if (--budget < 0) return 0;
    const unsigned f = rd(x, y, z);
    if (f & F_INSIDE) return 0;
    if (f & F_OBS) return 1;
  }
}
// The walk as k_convex runs it - the same float sequence, cheaper around it: 1 / |d| from a table (built once per handle
// by the double division above; intbound_half(d) = 0.5 / |d| is exactly half of it: tested for every |d| < 2048), ONE
// linear voxel index that moves by the axis' stride (a voxel has one index, so "end reached" is one compare; no axis makes
// more than |d| steps before the end is reached - the tMax of a finished axis is above 1, those of the others below - so
// the index never leaves the box of the two ends and cannot alias another voxel's).
__global__ void k_inv_table(float* inv) {
  for (int d = threadIdx.x; d < kDimLimit; d += blockDim.x) inv[d] = d ? (float)(1.0 / (double)d) : 0.0f;
}
__device__ __forceinline__ int ray_walk_lin(const Dev& D, const uint8_t* fl, const float* inv, int cx, int cy, int cz, int target) {
  const int dx = px(target) - cx, dy = py(target) - cy, dz = pz(target) - cz;
  const int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy, az = dz < 0 ? -dz : dz;
  const int ix = dx < 0 ? -D.max_yz : D.max_yz, iy = dy < 0 ? -D.max_z : D.max_z, iz = dz < 0 ? -1 : 1;  // never taken for d == 0
  const float tDX = inv[ax], tDY = inv[ay], tDZ = inv[az];
  float tMaxX = dx ? 0.5f * tDX : INFINITY, tMaxY = dy ? 0.5f * tDY : INFINITY, tMaxZ = dz ? 0.5f * tDZ : INFINITY;
  int id = cx * D.max_yz + cy * D.max_z + cz;
  const int eid = px(target) * D.max_yz + py(target) * D.max_z + pz(target);
  int budget = ax + ay + az;
  for (;;) {
    if (tMaxX < tMaxY) {
      if (tMaxX < tMaxZ) { id += ix; tMaxX += tDX; }
      else               { id += iz; tMaxZ += tDZ; }
    } else {
      if (tMaxY < tMaxZ) { id += iy; tMaxY += tDY; }
      else               { id += iz; tMaxZ += tDZ; }
    }
    if (id == eid) return 0;
    if (--budget < 0) return 0;
    const unsigned f = fl[id];
    if (f & F_INSIDE) return 0;
    if (f & F_OBS) return 1;
  }
}
// The same walk with the loads taken off its critical path.  Which voxel comes next never depends on what the voxels
// hold - only WHEN the walk ends does - so a trip of the loop advances kWalkDepth voxels (selects, no branch: a lane whose
// walk has reached its end voxel or used up its budget stands still on a valid voxel), issues their loads together and then
// judges them in step order; steps taken past a hit are thrown away.  The loop is uniform over the wave (it runs while any
// lane is undecided), so there is no exec-mask bookkeeping for the four exits of the serial form either.  Same float
// sequence, same visiting order, same answer as ray_walk_lin bit for bit (tests/test_gpu_cluster.py).
#ifndef CLUSTER_WALK_DEPTH
#define CLUSTER_WALK_DEPTH 4
#endif
constexpr int kWalkDepth = CLUSTER_WALK_DEPTH;
__device__ __forceinline__ int ray_walk_pipe(const Dev& D, const uint8_t* fl, const float* inv, int cx, int cy, int cz, int target, int valid) {
  const int tg = valid ? target : 0;
  const int ex = valid ? px(tg) : cx, ey = valid ? py(tg) : cy, ez = valid ? pz(tg) : cz;
  const int dx = ex - cx, dy = ey - cy, dz = ez - cz;
  const int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy, az = dz < 0 ? -dz : dz;
  const int ix = dx < 0 ? -D.max_yz : D.max_yz, iy = dy < 0 ? -D.max_z : D.max_z, iz = dz < 0 ? -1 : 1;
  const float tDX = inv[(unsigned)ax], tDY = inv[(unsigned)ay], tDZ = inv[(unsigned)az];
  float tMaxX = dx ? 0.5f * tDX : INFINITY, tMaxY = dy ? 0.5f * tDY : INFINITY, tMaxZ = dz ? 0.5f * tDZ : INFINITY;
  const int cid = cx * D.max_yz + cy * D.max_z + cz;
  int id = cid;
  const int eid = ex * D.max_yz + ey * D.max_z + ez;
  int budget = ax + ay + az;
  int res = valid ? -1 : 0;  // -1: walking
  while (__any(res < 0)) {
    bool live = res < 0;
    unsigned f[kWalkDepth];
    bool lv[kWalkDepth];
#pragma unroll
    for (int u = 0; u < kWalkDepth; u++) {
      const bool xy = tMaxX < tMaxY, xz = tMaxX < tMaxZ, yz = tMaxY < tMaxZ;
      const bool sx = xy & xz, sy = !xy & yz;  // (bitwise on purpose: three lane masks, no selects of booleans)
      const bool sz = !(sx | sy);
      const int nid = id + (sx ? ix : (sy ? iy : iz));
      const float nx = tMaxX + tDX, ny = tMaxY + tDY, nz = tMaxZ + tDZ;
      tMaxX = sx ? nx : tMaxX;  // (what a lane that stands still does to its tMax is never looked at again)
      tMaxY = sy ? ny : tMaxY;
      tMaxZ = sz ? nz : tMaxZ;
      budget -= 1;
      live = live & (nid != eid) & (budget >= 0);
      id = live ? nid : cid;  // a walk that is over stands on the candidate's own voxel: the SAME cache line for every such lane
      lv[u] = live;           // of the wave (k_convex runs at the L1's line rate: a lane parked on a line of its own costs a lookup)
      f[u] = fl[(unsigned)id];  // (a voxel index is never negative: base + 32-bit offset addressing)
    }
    // The four flag bytes side by side, a finished step standing in as "inside" (-> 0): the lowest set bit of the word is the
    // first step that ends the walk, and within a byte F_INSIDE (bit 2) comes before F_OBS (bit 3) as in the serial walk.
    static_assert(kWalkDepth <= 4 && F_INSIDE == 4 && F_OBS == 8, "one byte per step");
    unsigned P = 0;
#pragma unroll
    for (int u = 0; u < kWalkDepth; u++) P |= (lv[u] ? f[u] : (unsigned)F_INSIDE) << (8 * u);
    P &= 0x0c0c0c0cu;
    const int pos = __builtin_ctz(P | 0x80000000u);
    const int r = P == 0u ? -1 : (((pos & 7) == 3) ? 1 : 0);
    res = res >= 0 ? res : r;
  }
  return res;
}
// the tests in front of a walk as k_convex runs them: the reference's two, then the box test
// (any_target: the kernel-level entry point takes arbitrary targets; inside polygonGeneration a target is a cluster voxel or
// a candidate, and neither lies inside the cube: CS:491-494, 301-353)
// Inside polygonGeneration inside_data is the strict interior of the inflated cube and nothing else (k_inflate writes it
// once, CS:441-494; no later kernel touches the bit): the midpoint test is three range compares on the cube's corners instead
// of a scattered byte load per ray - a third of the loads in front of the walks (12 G rays per 64-seed call).  The kernel-level
// entry point (FULL) takes an arbitrary inside_data array and keeps the load.
struct CubeBox {
  int x0, x1, y0, y1, z0, z1;  // inside <=> x0 < x < x1 && y0 < y < y1 && z0 < z < z1
};
template <bool FULL>
__device__ __forceinline__ int ray_needs_walk(const Dev& D, const uint8_t* fl, const CubeBox& cb, int cx, int cy, int cz, int target) {
  const int ex = px(target), ey = py(target), ez = pz(target);
  if (FULL && (fl[(unsigned)(ex * D.max_yz + ey * D.max_z + ez)] & F_INSIDE)) return 0;
  const int mx = cx / 2 + (ex >> 1), my = cy / 2 + (ey >> 1), mz = cz / 2 + (ez >> 1);
  if (FULL) {
    if (fl[(unsigned)(mx * D.max_yz + my * D.max_z + mz)] & F_INSIDE) return 0;
  } else if ((mx > cb.x0) & (mx < cb.x1) & (my > cb.y0) & (my < cb.y1) & (mz > cb.z0) & (mz < cb.z1)) {
    return 0;
  }
  return box_obstacles(D, cx < ex ? cx : ex, cy < ey ? cy : ey, cz < ez ? cz : ez, cx < ex ? ex : cx, cy < ey ? ey : cy,
                       cz < ez ? ez : cz) != 0;
}
__device__ __forceinline__ int ray_blocked(const Dev& D, const uint8_t* fl, int cx, int cy, int cz, int target) {
  return ray_blocked_t([&](int x, int y, int z) -> unsigned { return fl[x * D.max_yz + y * D.max_z + z]; }, cx, cy, cz, target);
}

// bounding boxes of the cluster in chunks of 256 voxels (cluster order is spatially coherent: the cube's surface layer by
// layer, then every round's shell): k_convex tests a candidate against a whole chunk's box before it looks at its rays
__global__ __launch_bounds__(256) void k_chunk_box(Dev D) {
  __shared__ int red[4][6];
  const int e = blockIdx.y, m = blockIdx.x, tid = threadIdx.x;
  const Elem* E = &D.el[e];
  if (!E->live) return;
  const int n = E->n_cluster;
  if (m * 256 >= n) return;
  const int t = m * 256 + tid;
  int v[6] = {kDimLimit, kDimLimit, kDimLimit, -1, -1, -1};
  if (t < n) {
    const int p = D.cluster[(size_t)e * D.ccap + t];
    v[0] = v[3] = px(p); v[1] = v[4] = py(p); v[2] = v[5] = pz(p);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1)
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const int lo = __shfl_xor(v[a], d), hi = __shfl_xor(v[a + 3], d);
      v[a] = lo < v[a] ? lo : v[a];
      v[a + 3] = hi > v[a + 3] ? hi : v[a + 3];
    }
  if ((tid & 63) == 0)
    for (int a = 0; a < 6; a++) red[tid >> 6][a] = v[a];
  __syncthreads();
  if (tid < 6) {
    int r = red[0][tid];
    for (int w = 1; w < 4; w++) r = tid < 3 ? (red[w][tid] < r ? red[w][tid] : r) : (red[w][tid] > r ? red[w][tid] : r);
    D.cbox[((size_t)e * D.nchunk + m) * 6 + tid] = r;
  }
}

// one workgroup per candidate.  full = 1 (kernel-level parity entry point): no early exit, every row is complete
template <bool FULL>
__device__ __forceinline__ void convex_one(const Dev& D, const Elem* E, int e, int i);
// A workgroup takes the candidates blockIdx.x, blockIdx.x + gridDim.x, ... of its seed: the grid is sized for the
// typical shell (a few thousand candidates), not for the candidate CAPACITY - one workgroup per capacity slot meant
// 640 k workgroups per round for 64 seeds, nine in ten of which only looked at n_cand and left (44 M wave launches per
// call, ~10 % of the kernel's time).
template <bool FULL>
__global__ __launch_bounds__(256) void k_convex(Dev D) {
  const int e = blockIdx.y;
  const Elem* E = &D.el[e];
  if (!E->live) return;
  const int n_cand = E->n_cand;
  for (int i = blockIdx.x; i < n_cand; i += gridDim.x) {
    convex_one<FULL>(D, E, e, i);
    __syncthreads();  // the next candidate reuses the workgroup's LDS
  }
}
#if defined(CLUSTER_WALK_SERIAL)  // A/B builds: the branching one-voxel-per-trip walk
#define WALK(D, fl, inv, cx, cy, cz, q, valid) ((valid) ? ray_walk_lin(D, fl, inv, cx, cy, cz, q) : 0)
#else
#define WALK ray_walk_pipe
#endif
template <bool FULL>
__device__ __forceinline__ void convex_one(const Dev& D, const Elem* E, int e, int i) {
  constexpr int full = FULL ? 1 : 0;
  const int tid = threadIdx.x;
  // the inflated cube (k_inflate: E->vertex; corner 7 is its low, corner 1 / 9 / 17 its high end per axis)
  CubeBox cb;
  cb.x0 = E->vertex[7]; cb.x1 = E->vertex[1]; cb.y0 = E->vertex[15]; cb.y1 = E->vertex[9]; cb.z0 = E->vertex[23]; cb.z1 = E->vertex[17];
  const uint8_t* fl = D.flags + (size_t)e * D.G;
  const int* cl = D.cluster + (size_t)e * D.ccap;
  const int* cd = D.cand + (size_t)e * D.kcap;
  const int c = cd[i], cx = px(c), cy = py(c), cz = pz(c), n_clu = E->n_cluster;
  int bad = 0;
  // About half of a candidate's rays towards the cluster end at the two cheap tests (the midpoint lies inside the
  // inflated cube) and most of the rest at the box test: left in place they idle through the walks of their wave.  The
  // rays that have to be walked are queued (in order) and walked 64 at a time, every lane busy; the result is an AND, so
  // the order is free.  Every WAVE keeps its own queue and takes every fourth run of 64 targets: no workgroup barrier
  // inside the loops (two per 256 rays cost more than the tests between them), a flag in LDS carries the early exit.
  __shared__ int queue[4][128];
  __shared__ int s_bad;
  __shared__ unsigned char skip[256];
  const float* inv = D.inv;
  const int lane = tid & 63, wv = tid >> 6;
  int* wq = queue[wv];
  int head = 0, count = 0;  // uniform over the wave
  // whole chunks first: a chunk whose box, joined with the candidate, holds no obstacle cannot block it (k_chunk_box)
  const int nch = (n_clu + 255) >> 8;
  if (tid == 0) s_bad = 0;
  if (tid < nch) {
    const int* bx = D.cbox + ((size_t)e * D.nchunk + tid) * 6;
    skip[tid] = box_obstacles(D, cx < bx[0] ? cx : bx[0], cy < bx[1] ? cy : bx[1], cz < bx[2] ? cz : bx[2], cx > bx[3] ? cx : bx[3],
                              cy > bx[4] ? cy : bx[4], cz > bx[5] ? cz : bx[5]) == 0;
  }
  __syncthreads();
  // the wave's ordered append / take: LDS executes a wave's accesses in order, the barrier only stops the compiler
  auto push = [&](int need, int val) {
    const unsigned long long bal = __ballot(need);
    if (need) wq[(head + count + __popcll(bal & ((1ull << lane) - 1ull))) & 127] = val;
    count += __popcll(bal);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  auto take = [&](int n) {  // the lane's queued value (-1: none)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int v = lane < n ? wq[(head + lane) & 127] : -1;
    head = (head + n) & 127;
    count -= n;
    __builtin_amdgcn_wave_barrier();
    return v;
  };
  for (int m = nch - 1; m >= 0; m--) {
    // newest cluster voxels first, like the reference's loop (cluster_engine_cpu.cpp:41): they lie next to the
    // candidate shell and are the likeliest to reject, so the early exit below comes sooner
    if (m < 256 && skip[m]) continue;
    const int t = m * 256 + tid;
    int tgt = 0, need = 0;
    if (t < n_clu) {
      tgt = cl[t];
      need = ray_needs_walk<FULL>(D, fl, cb, cx, cy, cz, tgt);
    }
    push(need, tgt);
    if (count >= 64) {
      const int q = take(64);
      bad |= WALK(D, fl, inv, cx, cy, cz, q, 1);
      if (!full && __any(bad)) s_bad = 1;  // a rejected candidate's rays towards other candidates are never consulted
    }
    if (!full && *(volatile int*)&s_bad) break;
  }
  if (count > 0 && !(!full && *(volatile int*)&s_bad)) {
    const int q = take(count);
    bad |= WALK(D, fl, inv, cx, cy, cz, q, q >= 0);
  }
  bad = __syncthreads_or(bad);
  if (tid == 0) D.can_clu[(size_t)e * D.kcap + i] = bad ? 0 : 1;
  if (bad && !full) return;
  unsigned long long* row = D.blocked + ((size_t)e * D.kcap + i) * D.kwords;
  constexpr int kRowWords = 256;  // candidate capacities up to 16384 take the queued form
  __shared__ unsigned long long srow[kRowWords];
  if (D.kwords > kRowWords) {
    for (int base = 0; base < i; base += 256) {  // rays towards the candidates before this one
      const int j = base + tid;
      const int b = j < i ? ray_blocked(D, fl, cx, cy, cz, cd[j]) : 0;
      const unsigned long long m = __ballot(b);
      if ((tid & 63) == 0 && base + (tid & ~63) < i) row[(base + tid) >> 6] = m;
    }
    return;
  }
  // the same queues for the rays towards the earlier candidates: the row of the bit matrix is collected in LDS (a blocked
  // ray is rare: one ds_or per hit) and written once
  const int nw = (i + 63) / 64;
  for (int w = tid; w < nw; w += 256) srow[w] = 0ull;
  head = 0;
  count = 0;
  __syncthreads();
  for (int base = 0; base < i; base += 256) {
    const int j = base + tid;
    push(j < i ? ray_needs_walk<FULL>(D, fl, cb, cx, cy, cz, cd[j]) : 0, j);
    if (count >= 64) {
      const int jq = take(64);
      if (WALK(D, fl, inv, cx, cy, cz, cd[jq], 1)) atomicOr(&srow[jq >> 6], 1ull << (jq & 63));
    }
  }
  if (count > 0) {
    const int jq = take(count);
    if (WALK(D, fl, inv, cx, cy, cz, jq >= 0 ? cd[jq] : 0, jq >= 0)) atomicOr(&srow[jq >> 6], 1ull << (jq & 63));
  }
  __syncthreads();
  int nz = 0;
  for (int w = tid; w < nw; w += 256) {
    row[w] = srow[w];
    nz |= srow[w] != 0ull;
  }
  // A candidate that sees the cluster AND every earlier candidate joins whatever the accept loop decides about the
  // others (CS:360-384: it is only ever compared with candidates it cannot see): marked 2, and k_resolve_fast takes it
  // without a place in its sequential chain - which then only holds the few candidates with a blocked ray
  nz = __syncthreads_or(nz);
  if (!full && !nz && tid == 0) D.can_clu[(size_t)e * D.kcap + i] = 2;
}

// the sequential accept loop (CS:360-384) for one seed: one wave.  dry = 1: only accept[] is written.
__global__ __launch_bounds__(64) void k_resolve(Dev D, int dry) {
  extern __shared__ unsigned long long acc[];  // accepted-candidate bitset, kwords words
  const int e = blockIdx.x, lane = threadIdx.x;
  Elem* E = &D.el[e];
  if (!E->live) return;
  const int n_cand = E->n_cand, n_clu = E->n_cluster;
  for (int w = lane; w < D.kwords; w += 64) acc[w] = 0ull;
  __syncthreads();
  int count = 0, overflow = 0;
  const int* cd = D.cand + (size_t)e * D.kcap;
  int* cl = D.cluster + (size_t)e * D.ccap;
  int* ac = D.active + (size_t)e * D.ccap;
  uint8_t* fl = D.flags + (size_t)e * D.G;
  for (int i = 0; i < n_cand; i++) {
    int ok = D.can_clu[(size_t)e * D.kcap + i] != 0;
    if (ok) {
      const unsigned long long* row = D.blocked + ((size_t)e * D.kcap + i) * D.kwords;
      int hit = 0;
      for (int w = lane; w < (i + 63) / 64; w += 64) hit |= (row[w] & acc[w]) != 0ull;
      ok = !__any(hit);
    }
    if (lane == 0) {
      D.accept[(size_t)e * D.kcap + i] = (uint8_t)ok;
      if (ok) acc[i >> 6] |= 1ull << (i & 63);
      if (!dry) {
        const int p = cd[i];
        if (ok) {
          if (n_clu + count < D.ccap) { cl[n_clu + count] = p; ac[count] = p; }
          else overflow = 1;
        } else {
          fl[px(p) * D.max_yz + py(p) * D.max_z + pz(p)] |= F_INVALID;  // CS:380-383
        }
      }
    }
    count += ok;
    __syncthreads();
  }
  if (lane == 0 && !dry) {
    if (overflow) { E->rtn = DIRECT_CLUSTER_OVERFLOW; E->live = 0; E->n_cluster = D.ccap; E->n_active = 0; }
    else {
      E->n_cluster = n_clu + count;
      E->n_active = count;
      if (count == 0) E->live = 0;  // CS:386-387
      else E->iters += 1;           // CS:389
    }
  }
}

// The sequential accept loop (CS:360-384) of one round, one wave per seed, with the candidate's row of the bit matrix
// fetched ONE CANDIDATE AHEAD: the loop is a chain of dependent global loads otherwise (1.5 us per candidate).
__global__ __launch_bounds__(64) void k_resolve_pipe(Dev D) {
  extern __shared__ unsigned long long acc[];  // accepted-candidate bitset, kwords words
  const int e = blockIdx.x, lane = threadIdx.x;
  Elem* E = &D.el[e];
  if (!E->live) return;
  const int n_cand = E->n_cand, n_clu = E->n_cluster;
  if (n_cand == 0) return;
  for (int w = lane; w < D.kwords; w += 64) acc[w] = 0ull;
  __syncthreads();
  int count = 0, overflow = 0;
  const int* cd = D.cand + (size_t)e * D.kcap;
  const uint8_t* cc = D.can_clu + (size_t)e * D.kcap;
  int* cl = D.cluster + (size_t)e * D.ccap;
  int* ac = D.active + (size_t)e * D.ccap;
  uint8_t* fl = D.flags + (size_t)e * D.G;
  constexpr int kW = 4;  // row words per lane held in registers: candidates up to 64 * 64 * kW
  auto load_row = [&](int i, unsigned long long* r, int& okc, int& pc) {
    okc = cc[i] != 0;
    pc = cd[i];
    const unsigned long long* row = D.blocked + ((size_t)e * D.kcap + i) * D.kwords;
    const int nw = (i + 63) / 64;
#pragma unroll
    for (int q = 0; q < kW; q++) {
      const int w = lane + 64 * q;
      r[q] = (okc && w < nw) ? row[w] : 0ull;  // rows of rejected candidates were never written
    }
  };
  unsigned long long rn[kW];
  int okn, pn;
  load_row(0, rn, okn, pn);
  for (int i = 0; i < n_cand; i++) {
    unsigned long long r[kW];
#pragma unroll
    for (int q = 0; q < kW; q++) r[q] = rn[q];
    int ok = okn;
    const int p = pn;
    if (i + 1 < n_cand) load_row(i + 1, rn, okn, pn);
    if (ok) {
      int hit = 0;
#pragma unroll
      for (int q = 0; q < kW; q++) {
        const int w = lane + 64 * q;
        if (w < D.kwords) hit |= (r[q] & acc[w]) != 0ull;
      }
      for (int w = lane + 64 * kW; w < (i + 63) / 64; w += 64)  // beyond the registers (more than 16384 candidates)
        hit |= (D.blocked[((size_t)e * D.kcap + i) * D.kwords + w] & acc[w]) != 0ull;
      ok = !__any(hit);
    }
    if (lane == 0) {
      D.accept[(size_t)e * D.kcap + i] = (uint8_t)ok;
      if (ok) {
        acc[i >> 6] |= 1ull << (i & 63);
        if (n_clu + count < D.ccap) { cl[n_clu + count] = p; ac[count] = p; }
        else overflow = 1;
      } else {
        fl[px(p) * D.max_yz + py(p) * D.max_z + pz(p)] |= F_INVALID;  // CS:380-383
      }
    }
    count += ok;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  overflow = __any(overflow);
  if (lane == 0) {
    if (overflow) { E->rtn = DIRECT_CLUSTER_OVERFLOW; E->live = 0; E->n_cluster = D.ccap; E->n_active = 0; }
    else {
      E->n_cluster = n_clu + count;
      E->n_active = count;
      if (count == 0) E->live = 0;  // CS:386-387
      else E->iters += 1;           // CS:389
    }
  }
}

// candidates [i0, i1) of the accept loop with KW row words per lane (see k_resolve_fast)
template <int KW>
__device__ __forceinline__ void resolve_range(const Dev& D, int e, int lane, const unsigned long long* ccm, unsigned long long* acc, int i0, int i1) {
  constexpr int KG = 32 / KW;
  unsigned long long bufA[KG][KW], bufB[KG][KW];
  int okA[KG], okB[KG];
  auto load_group = [&](int g0, unsigned long long (*buf)[KW], int* okv) {
    // a group lies inside one 64-candidate word of the cluster-test mask (KG divides 64, i0 is a multiple of 4096)
    const int gw = (g0 < i1 ? g0 : i0) >> 6;
    unsigned long long gm = 0ull;
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (q == (gw >> 6)) gm = __shfl(ccm[q], gw & 63);
#pragma unroll
    for (int j = 0; j < KG; j++) {
      const int i = g0 + j;
      const int ic = i < i1 ? i : i0;
      const int okc = i < i1 ? (int)((gm >> (i & 63)) & 1ull) : 0;
      okv[j] = okc;
      const unsigned long long* row = D.blocked + ((size_t)e * D.kcap + ic) * D.kwords;
      const int nw = (i + 63) / 64;
#pragma unroll
      for (int q = 0; q < KW; q++) {
        const int w = lane + 64 * q;
        buf[j][q] = (okc && w < nw) ? row[w] : 0ull;  // rows of rejected candidates were never written
      }
    }
  };
  auto eval_group = [&](int g0, unsigned long long (*buf)[KW], const int* okv) {
#pragma unroll
    for (int j = 0; j < KG; j++) {
      const int i = g0 + j;
      if (i >= i1 || !okv[j]) continue;  // uniform
      int hit = 0;
#pragma unroll
      for (int q = 0; q < KW; q++) hit |= (buf[j][q] & acc[q]) != 0ull;
      if (__any(hit)) continue;
      const int w = i >> 6;
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (q == (w >> 6) && lane == (w & 63)) acc[q] |= 1ull << (i & 63);
    }
  };
  load_group(i0, bufA, okA);
  for (int g0 = i0; g0 < i1; g0 += 2 * KG) {
    load_group(g0 + KG, bufB, okB);
    eval_group(g0, bufA, okA);
    load_group(g0 + 2 * KG, bufA, okA);
    eval_group(g0 + KG, bufB, okB);
  }
}

// The accept loop without a memory round trip on its serial chain (candidate capacities up to 64 * 64 * 4 = 16384; k_resolve_pipe
// above stays for larger ones).  The accepted-candidate bitset lives in REGISTERS (word w in lane w % 64, slot w / 64),
// the rows of the bit matrix arrive in groups of up to 32 candidates, two groups in flight, and nothing is written inside the
// loop: k_resolve_pipe's per-candidate stores needed a release fence, i.e. a wait for every outstanding load - the
// prefetched row included -, which made every candidate cost one global-memory latency (~1 us).  The decisions are the
// same sequence (candidate i joins iff it sees the old cluster and every earlier candidate that has joined); cluster /
// active appends, accept bytes and invalid flags are written afterwards by k_apply, one thread per candidate, from the bitset.
__global__ __launch_bounds__(64) void k_resolve_fast(Dev D) {
  const int e = blockIdx.x, lane = threadIdx.x;
  Elem* E = &D.el[e];
  if (!E->live) return;
  const int n_cand = E->n_cand, n_clu = E->n_cluster;
  if (n_cand == 0) return;
  const int* cd = D.cand + (size_t)e * D.kcap;
  const uint8_t* cc = D.can_clu + (size_t)e * D.kcap;
  constexpr int kW = 4;
  unsigned long long acc[kW] = {0ull, 0ull, 0ull, 0ull};
  // the cluster-test results as a bitset: word w (candidates 64 w ..) in lane w % 64, slot w / 64
  unsigned long long ccm[kW] = {0ull, 0ull, 0ull, 0ull};
  const int nwords = (n_cand + 63) / 64;
  for (int w0 = 0; w0 < nwords; w0 += 8) {
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int i = (w0 + u) * 64 + lane;
      v[u] = (w0 + u < nwords && i < n_cand) ? (int)cc[i] : 0;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      // 1: joins unless an earlier accepted candidate is hidden from it (the sequential chain below); 2: sees everything
      // before it (k_convex): accepted at once - its bit can stand in acc from the start, rows only hold EARLIER candidates
      const unsigned long long m = __ballot(v[u] == 1), m2 = __ballot(v[u] == 2);
      const int w = w0 + u;
#pragma unroll
      for (int q = 0; q < kW; q++)
        if (q == (w >> 6) && lane == (w & 63)) { ccm[q] = m; acc[q] = m2; }
    }
  }
  // candidates below 4096 only meet row words below 64 (one per lane), below 8192 two per lane, ...: the narrower the
  // rows, the more candidates fit into the two groups in flight (32 / 16 / 8 per group)
  resolve_range<1>(D, e, lane, ccm, acc, 0, n_cand < 4096 ? n_cand : 4096);
  if (n_cand > 4096) resolve_range<2>(D, e, lane, ccm, acc, 4096, n_cand < 8192 ? n_cand : 8192);
  if (n_cand > 8192) resolve_range<4>(D, e, lane, ccm, acc, 8192, n_cand);
  // the bitset and its running counts for k_apply
  int count = 0;
#pragma unroll
  for (int q = 0; q < kW; q++) {
    const int c = __popcll(acc[q]);
    int inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(inc, d);
      if (lane >= d) inc += t;
    }
    const int w = q * 64 + lane;
    if (w < D.kwords) {
      D.accbits[(size_t)e * D.kwords + w] = acc[q];
      D.accpre[(size_t)e * D.kwords + w] = count + inc - c;
    }
    count += __shfl(inc, 63);
  }
  if (lane == 0) {
    E->pad0 = n_clu;  // where this round's voxels go (k_apply)
    // overflow: k_apply skips a dead element, so only the n_clu voxels of the earlier rounds are in the cluster array -
    // that valid prefix is what stays (slots beyond it were never written)
    if (n_clu + count > D.ccap) { E->rtn = DIRECT_CLUSTER_OVERFLOW; E->live = 0; E->n_cluster = n_clu; E->n_active = 0; }
    else {
      E->n_cluster = n_clu + count;
      E->n_active = count;
      if (count == 0) E->live = 0;  // CS:386-387
      else E->iters += 1;           // CS:389
    }
  }
}

// what the accept loop decided, applied in parallel: accepted candidates join the cluster and become the active set of
// the next round in candidate order (CS:371-379), the others are marked invalid (CS:380-383)
__global__ __launch_bounds__(256) void k_apply(Dev D) {
  const int e = blockIdx.y;
  const Elem* E = &D.el[e];
  if (!E->live) return;  // finished before this round, finished in it (nothing accepted), or overflowed (result not usable)
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= E->n_cand) return;
  const int w = i >> 6, b = i & 63;
  const unsigned long long word = D.accbits[(size_t)e * D.kwords + w];
  const int ok = (int)((word >> b) & 1ull);
  const int p = D.cand[(size_t)e * D.kcap + i];
  D.accept[(size_t)e * D.kcap + i] = (uint8_t)ok;
  if (ok) {
    const int r = D.accpre[(size_t)e * D.kwords + w] + __popcll(word & ((1ull << b) - 1ull));
    D.cluster[(size_t)e * D.ccap + E->pad0 + r] = p;
    D.active[(size_t)e * D.ccap + r] = p;
  } else {
    D.flags[(size_t)e * D.G + px(p) * D.max_yz + py(p) * D.max_z + pz(p)] |= F_INVALID;
  }
}

__global__ void k_emit(Dev D, int batch, int32_t* vertex_idx, int32_t* cluster_xyz, int32_t* cluster_num, int32_t* iters,
                       int32_t* rtn) {
  const int e = blockIdx.y;
  const Elem* E = &D.el[e];
  const int n = E->rtn == DIRECT_CLUSTER_BAD_SEED ? 0 : E->n_cluster;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) {
    if (cluster_num) cluster_num[e] = n;
    if (iters) iters[e] = E->iters;
    if (rtn) rtn[e] = E->rtn;
  }
  if (vertex_idx && t < 24) vertex_idx[e * 24 + t] = E->vertex[t];
  if (cluster_xyz)
    for (int q = t; q < n; q += gridDim.x * blockDim.x) {
      const int p = D.cluster[(size_t)e * D.ccap + q];
      int32_t* o = cluster_xyz + ((size_t)e * D.ccap + q) * 3;
      o[0] = px(p); o[1] = py(p); o[2] = pz(p);
    }
  (void)batch;
}

#include "hull_kernels.h"

thread_local std::string g_cerr;
direct_status_t cfail(direct_status_t st, const std::string& msg) {
  g_cerr = msg;
  return st;
}
#define CHIP_TRY(expr)                                                                             \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return cfail(DIRECT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));          \
  } while (0)

}  // namespace

struct direct_cluster_handle_s {
  direct_cluster_config_t cfg = {};
  Dev D = {};
  uint8_t* map = nullptr;
  uint8_t* inside_tmp = nullptr;
  int* seeds = nullptr;
  int32_t *st_vertex = nullptr, *st_xyz = nullptr, *st_num = nullptr, *st_iters = nullptr, *st_rtn = nullptr;  // device staging of host outputs
  bool have_map = false;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  std::vector<void*> allocs;
  HullDev H = {};          // scratch of hull_planes_batch, allocated by its first call
  bool have_hull = false;
  int convex_grid = 2048;    // workgroups per seed of k_convex (DIRECT_CLUSTER_CONVEX_GRID: experiments)
  int resident_batch = 0;    // seeds of the last polygon_generation_batch whose clusters are still in D.cluster / D.el (0: none)
  void* hull_out = nullptr;  // device staging of its host outputs
  size_t hull_out_bytes = 0;
};

extern "C" {

const char* direct_cluster_last_error(void) { return g_cerr.c_str(); }

direct_status_t direct_cluster_create(const direct_cluster_config_t* cfg, direct_cluster_handle_t* out) {
  if (!cfg || !out) return cfail(DIRECT_ERR_INVALID, "null argument");
  if (cfg->max_x <= 0 || cfg->max_y <= 0 || cfg->max_z <= 0 || cfg->max_batch <= 0 || cfg->cluster_capacity <= 0 ||
      cfg->candidate_capacity <= 0)
    return cfail(DIRECT_ERR_INVALID, "bad sizes");
  if (cfg->max_x > kDimLimit || cfg->max_y > kDimLimit || cfg->max_z > kDimLimit)
    return cfail(DIRECT_ERR_UNSUPPORTED, "map dimensions above 1024 voxels per axis");
  if ((long long)cfg->max_x * cfg->max_y * cfg->max_z > 0x7fffffffLL / 2) return cfail(DIRECT_ERR_UNSUPPORTED, "map too large");
  // the summed-area table has (X+1)(Y+1)(Z+1) int entries and box_obstacles addresses it with unsigned 32-bit BYTE offsets
  if ((long long)(cfg->max_x + 1) * (cfg->max_y + 1) * (cfg->max_z + 1) >= (1LL << 30))
    return cfail(DIRECT_ERR_UNSUPPORTED, "map too large: the summed-area table would exceed 4 GiB");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return cfail(DIRECT_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return cfail(DIRECT_ERR_INVALID, "bad device ordinal");
  CHIP_TRY(hipSetDevice(cfg->device));
  hipDeviceProp_t prop;
  CHIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return cfail(DIRECT_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
  direct_cluster_handle_t h = new direct_cluster_handle_s();
  h->cfg = *cfg;
  if (const char* ev = getenv("DIRECT_CLUSTER_CONVEX_GRID")) h->convex_grid = atoi(ev) > 0 ? atoi(ev) : h->convex_grid;
  Dev& D = h->D;
  D.max_x = cfg->max_x; D.max_y = cfg->max_y; D.max_z = cfg->max_z; D.max_yz = cfg->max_y * cfg->max_z;
  D.G = cfg->max_x * cfg->max_y * cfg->max_z;
  D.ccap = cfg->cluster_capacity; D.kcap = cfg->candidate_capacity; D.kwords = (cfg->candidate_capacity + 63) / 64;
  const size_t B = cfg->max_batch, G = D.G;
  direct_status_t st = DIRECT_OK;
  auto A = [&](auto pp, size_t bytes) {
    if (st != DIRECT_OK) return;
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, bytes ? bytes : 16);
    if (e != hipSuccess) { st = cfail(DIRECT_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e)); return; }
    h->allocs.push_back(q);
    *pp = (typename std::remove_pointer<decltype(pp)>::type)q;
  };
  D.sat_z = cfg->max_z + 1; D.sat_yz = (cfg->max_y + 1) * D.sat_z;
  A(&D.sat, (size_t)(cfg->max_x + 1) * D.sat_yz * sizeof(int));
  D.nchunk = (cfg->cluster_capacity + 255) / 256;
  float* invp = nullptr;
  A(&invp, kDimLimit * sizeof(float));
  D.inv = invp;
  A(&D.cbox, B * (size_t)D.nchunk * 6 * sizeof(int));
  A(&h->map, G); A(&h->inside_tmp, G); A(&h->seeds, B * 3 * sizeof(int));
  A(&D.flags, B * G); A(&D.key, B * G * sizeof(int));
  A(&D.cluster, B * D.ccap * sizeof(int)); A(&D.active, B * D.ccap * sizeof(int)); A(&D.cand, B * D.kcap * sizeof(int));
  A(&D.can_clu, B * D.kcap); A(&D.accept, B * D.kcap);
  A(&D.blocked, B * D.kcap * (size_t)D.kwords * sizeof(unsigned long long));
  A(&D.accbits, B * (size_t)D.kwords * sizeof(unsigned long long)); A(&D.accpre, B * (size_t)D.kwords * sizeof(int));
  A(&D.el, B * sizeof(Elem));
  A(&D.ccnt, B * kCompactBlocks * sizeof(int)); A(&D.csnap, B * 2 * sizeof(int));
  A(&h->st_vertex, B * 24 * 4); A(&h->st_xyz, B * (size_t)D.ccap * 12); A(&h->st_num, B * 4); A(&h->st_iters, B * 4); A(&h->st_rtn, B * 4);
  D.map = h->map;
  if (st == DIRECT_OK) {
    hipLaunchKernelGGL(k_inv_table, dim3(1), dim3(256), 0, nullptr, invp);
    if (hipDeviceSynchronize() != hipSuccess) st = cfail(DIRECT_ERR_DEVICE, "k_inv_table failed");
  }
  if (st == DIRECT_OK && (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess))
    st = cfail(DIRECT_ERR_DEVICE, "hipEventCreate failed");
  if (st != DIRECT_OK) {
    direct_cluster_destroy(h);
    return st;
  }
  *out = h;
  return DIRECT_OK;
}

direct_status_t direct_cluster_destroy(direct_cluster_handle_t h) {
  if (!h) return DIRECT_OK;
  (void)hipSetDevice(h->cfg.device);
  (void)hipDeviceSynchronize();
  for (void* p : h->allocs) (void)hipFree(p);
  if (h->hull_out) (void)hipFree(h->hull_out);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  delete h;
  return DIRECT_OK;
}

direct_status_t direct_cluster_set_map(direct_cluster_handle_t h, int32_t mem, const uint8_t* map_data) {
  if (!h || !map_data) return cfail(DIRECT_ERR_INVALID, "null argument");
  CHIP_TRY(hipSetDevice(h->cfg.device));
  CHIP_TRY(hipMemcpyAsync(h->map, map_data, (size_t)h->D.G, mem == DIRECT_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                          h->stream));
  CHIP_TRY(hipStreamSynchronize(h->stream));
  {  // the obstacles' summed-area table for k_convex's box test
    const Dev& D = h->D;
    hipLaunchKernelGGL(k_sat_fill, dim3(1024), dim3(256), 0, h->stream, D);
    for (int axis = 2; axis >= 0; axis--) hipLaunchKernelGGL(k_sat_scan, dim3(256), dim3(256), 0, h->stream, D, axis);
    CHIP_TRY(hipGetLastError());
    CHIP_TRY(hipStreamSynchronize(h->stream));
  }
  h->have_map = true;
  return DIRECT_OK;
}

direct_status_t direct_cluster_polygon_generation_batch(direct_cluster_handle_t h, int32_t batch, const int32_t* seeds,
                                                        int32_t itr_inflate_max, int32_t itr_cluster_max, int32_t mem,
                                                        int32_t* vertex_idx, int32_t* cluster_xyz, int32_t* cluster_num,
                                                        int32_t* cluster_iters, int32_t* rtn) {
  if (!h || !seeds) return cfail(DIRECT_ERR_INVALID, "null argument");
  if (batch <= 0 || batch > h->cfg.max_batch) return cfail(DIRECT_ERR_INVALID, "batch exceeds the handle's max_batch");
  if (itr_inflate_max < 0 || itr_cluster_max < 0) return cfail(DIRECT_ERR_INVALID, "negative iteration limit");
  if (!h->have_map) return cfail(DIRECT_ERR_INVALID, "direct_cluster_set_map has not been called");
  CHIP_TRY(hipSetDevice(h->cfg.device));
  Dev& D = h->D;
  h->resident_batch = 0;
  CHIP_TRY(hipMemcpyAsync(h->seeds, seeds, (size_t)batch * 3 * sizeof(int), hipMemcpyHostToDevice, h->stream));
  CHIP_TRY(hipEventRecord(h->ev0, h->stream));
  const int gblocks = std::min((D.G + 255) / 256, 4096);
  hipLaunchKernelGGL(k_flags_init, dim3(gblocks, batch), dim3(256), 0, h->stream, D, (const uint8_t*)nullptr);  // flagClear CS:20-26
  hipLaunchKernelGGL(k_inflate, dim3(batch), dim3(256), 0, h->stream, D, (const int*)h->seeds, itr_inflate_max);
  CHIP_TRY(hipGetLastError());
  std::vector<Elem> el(batch);
  auto fetch = [&]() -> hipError_t {
    hipError_t e = hipMemcpyAsync(el.data(), D.el, (size_t)batch * sizeof(Elem), hipMemcpyDeviceToHost, h->stream);
    return e != hipSuccess ? e : hipStreamSynchronize(h->stream);
  };
  CHIP_TRY(fetch());
  // polytopeCluster_cpu (CS:295-392).  No launch of a round depends on a count the round produces (grid-stride or
  // capacity-sized kernels over device-resident counters; workgroups of finished seeds return at once), so the host enqueues rounds
  // blindly, kRoundsPerCheck at a time, and only then looks whether any seed is still growing: one read-back per
  // kRoundsPerCheck rounds instead of three per round (the reference's CUDA twin, cluster_server.cu:628-685, copies
  // per round as well).
  constexpr int kRoundsPerCheck = 4;
  for (int round = 0; round < itr_cluster_max;) {
    int live = 0;
    for (const Elem& E : el) live += E.live ? 1 : 0;
    if (!live) break;
    const int n = std::min(kRoundsPerCheck, itr_cluster_max - round);
    for (int r = 0; r < n; r++) {
      hipLaunchKernelGGL(k_mark, dim3(128, batch), dim3(256), 0, h->stream, D);
      hipLaunchKernelGGL(k_compact_count, dim3(kCompactBlocks, batch), dim3(256), 0, h->stream, D);
      hipLaunchKernelGGL(k_compact_write, dim3(kCompactBlocks, batch), dim3(256), 0, h->stream, D);
      // one workgroup per candidate SLOT: the slots beyond a seed's candidate count return at once (640 k empty
      // workgroups cost ~0.1 ms; a persistent ticket kernel with the rays' voxels staged in LDS was built and measured
      // 1.6 - 2 x slower: thousands of short workgroups balance the seeds' very different loads better, and the flag
      // bytes of a round's rays live in L2 anyway)
      hipLaunchKernelGGL(k_chunk_box, dim3(D.nchunk, batch), dim3(256), 0, h->stream, D);
      hipLaunchKernelGGL(k_convex<false>, dim3(std::min(D.kcap, h->convex_grid), batch), dim3(256), 0, h->stream, D);
      if (D.kwords <= 256) {
        hipLaunchKernelGGL(k_resolve_fast, dim3(batch), dim3(64), 0, h->stream, D);
        hipLaunchKernelGGL(k_apply, dim3((D.kcap + 255) / 256, batch), dim3(256), 0, h->stream, D);
      }
      else hipLaunchKernelGGL(k_resolve_pipe, dim3(batch), dim3(64), (size_t)D.kwords * 8, h->stream, D);
    }
    CHIP_TRY(hipGetLastError());
    round += n;
    CHIP_TRY(fetch());
  }
  int32_t *dv = vertex_idx, *dc = cluster_xyz, *dn = cluster_num, *di = cluster_iters, *dr = rtn;
  auto cleanup = [&]() {};
  if (mem == DIRECT_MEM_HOST) {  // host outputs are staged in buffers the handle owns (allocated once, in create)
    dv = vertex_idx ? h->st_vertex : nullptr; dc = cluster_xyz ? h->st_xyz : nullptr; dn = cluster_num ? h->st_num : nullptr;
    di = cluster_iters ? h->st_iters : nullptr; dr = rtn ? h->st_rtn : nullptr;
  }
  hipLaunchKernelGGL(k_emit, dim3(64, batch), dim3(256), 0, h->stream, D, batch, dv, dc, dn, di, dr);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipEventRecord(h->ev1, h->stream);
  h->timed = e == hipSuccess;
  if (e == hipSuccess && mem == DIRECT_MEM_HOST) {
    auto dn_ = [&](void* dst, const void* src, size_t bytes) {
      return dst ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream) : hipSuccess;
    };
    if (e == hipSuccess) e = dn_(vertex_idx, dv, (size_t)batch * 24 * 4);
    if (e == hipSuccess && cluster_xyz)  // only the filled prefix of every row
      for (int b = 0; b < batch && e == hipSuccess; b++) {
        const int n = el[b].rtn == DIRECT_CLUSTER_BAD_SEED ? 0 : el[b].n_cluster;
        if (n > 0) e = hipMemcpyAsync(cluster_xyz + (size_t)b * D.ccap * 3, dc + (size_t)b * D.ccap * 3, (size_t)n * 12,
                                      hipMemcpyDeviceToHost, h->stream);
      }
    if (e == hipSuccess) e = dn_(cluster_num, dn, (size_t)batch * 4);
    if (e == hipSuccess) e = dn_(cluster_iters, di, (size_t)batch * 4);
    if (e == hipSuccess) e = dn_(rtn, dr, (size_t)batch * 4);
  }
  hipError_t e2 = hipStreamSynchronize(h->stream);
  cleanup();
  if (e != hipSuccess || e2 != hipSuccess)
    return cfail(DIRECT_ERR_DEVICE, std::string("polygon_generation_batch: ") + hipGetErrorString(e != hipSuccess ? e : e2));
  h->resident_batch = batch;
  return DIRECT_OK;
}

direct_status_t direct_cluster_convex_test(direct_cluster_handle_t h, const uint8_t* inside_data, int32_t n_candidate,
                                           const int32_t* candidate_xyz, int32_t n_cluster, const int32_t* cluster_xyz,
                                           uint8_t* can_clu, uint8_t* can_can, uint8_t* accept) {
  if (!h || !inside_data || !candidate_xyz || (!cluster_xyz && n_cluster > 0) || !can_clu)
    return cfail(DIRECT_ERR_INVALID, "null argument");
  if (!h->have_map) return cfail(DIRECT_ERR_INVALID, "direct_cluster_set_map has not been called");
  Dev& D = h->D;
  if (n_candidate <= 0 || n_candidate > D.kcap || n_cluster < 0 || n_cluster > D.ccap)
    return cfail(DIRECT_ERR_INVALID, "candidate / cluster count exceeds the handle's capacity");
  CHIP_TRY(hipSetDevice(h->cfg.device));
  auto pack = [&](const int32_t* xyz, int n, std::vector<int>& out) -> bool {
    out.resize(n);
    for (int i = 0; i < n; i++) {
      const int x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
      if (x < 0 || x >= D.max_x || y < 0 || y >= D.max_y || z < 0 || z >= D.max_z) return false;
      out[i] = (x << 20) | (y << 10) | z;
    }
    return true;
  };
  std::vector<int> pc, pk;
  if (!pack(candidate_xyz, n_candidate, pc) || !pack(cluster_xyz, n_cluster, pk)) return cfail(DIRECT_ERR_INVALID, "voxel outside the map");
  Elem E = {};
  E.n_cluster = n_cluster; E.n_cand = n_candidate; E.live = 1;
  h->resident_batch = 0;  // element 0's cluster, candidates and flags are overwritten below: the last generation's clusters are gone
  CHIP_TRY(hipMemcpyAsync(h->inside_tmp, inside_data, (size_t)D.G, hipMemcpyHostToDevice, h->stream));
  CHIP_TRY(hipMemcpyAsync(D.cand, pc.data(), (size_t)n_candidate * 4, hipMemcpyHostToDevice, h->stream));
  if (n_cluster) CHIP_TRY(hipMemcpyAsync(D.cluster, pk.data(), (size_t)n_cluster * 4, hipMemcpyHostToDevice, h->stream));
  CHIP_TRY(hipMemcpyAsync(D.el, &E, sizeof E, hipMemcpyHostToDevice, h->stream));
  CHIP_TRY(hipMemsetAsync(D.blocked, 0, (size_t)n_candidate * D.kwords * 8, h->stream));
  CHIP_TRY(hipEventRecord(h->ev0, h->stream));
  hipLaunchKernelGGL(k_flags_init, dim3(std::min((D.G + 255) / 256, 4096), 1), dim3(256), 0, h->stream, D, (const uint8_t*)h->inside_tmp);
  hipLaunchKernelGGL(k_chunk_box, dim3(D.nchunk, 1), dim3(256), 0, h->stream, D);
  hipLaunchKernelGGL(k_convex<true>, dim3(n_candidate, 1), dim3(256), 0, h->stream, D);
  hipLaunchKernelGGL(k_resolve, dim3(1), dim3(64), (size_t)D.kwords * 8, h->stream, D, 1);
  CHIP_TRY(hipGetLastError());
  CHIP_TRY(hipEventRecord(h->ev1, h->stream));
  h->timed = true;
  std::vector<unsigned long long> rows(can_can ? (size_t)n_candidate * D.kwords : 0);
  CHIP_TRY(hipMemcpyAsync(can_clu, D.can_clu, (size_t)n_candidate, hipMemcpyDeviceToHost, h->stream));
  if (accept) CHIP_TRY(hipMemcpyAsync(accept, D.accept, (size_t)n_candidate, hipMemcpyDeviceToHost, h->stream));
  if (can_can) CHIP_TRY(hipMemcpyAsync(rows.data(), D.blocked, rows.size() * 8, hipMemcpyDeviceToHost, h->stream));
  CHIP_TRY(hipStreamSynchronize(h->stream));
  if (can_can)
    for (int i = 0; i < n_candidate; i++)
      for (int j = 0; j < i; j++)
        can_can[(size_t)i * (i - 1) / 2 + j] = ((rows[(size_t)i * D.kwords + (j >> 6)] >> (j & 63)) & 1ull) ? 0 : 1;
  return DIRECT_OK;
}

direct_status_t direct_cluster_hull_planes_batch(direct_cluster_handle_t h, int32_t batch, int32_t mem_in,
                                                 const int32_t* cluster_xyz, const int32_t* cluster_num, double resolution,
                                                 const double* map_lower, int32_t plane_capacity, int32_t vertex_capacity,
                                                 int32_t mem, double* planes, int64_t* plane_int, int32_t* n_planes,
                                                 double* vertices, int32_t* n_vertices, double* center, int32_t* degenerate,
                                                 int32_t* rtn) {
  if (!h || !map_lower) return cfail(DIRECT_ERR_INVALID, "null argument");
  if (batch <= 0 || batch > h->cfg.max_batch) return cfail(DIRECT_ERR_INVALID, "batch exceeds the handle's max_batch");
  if (cluster_xyz && !cluster_num) return cfail(DIRECT_ERR_INVALID, "cluster_xyz without cluster_num");
  if (plane_capacity <= 0 || vertex_capacity <= 0 || !(resolution > 0)) return cfail(DIRECT_ERR_INVALID, "bad capacity / resolution");
  if (!cluster_xyz && batch > h->resident_batch)
    return cfail(DIRECT_ERR_INVALID, "no resident clusters for this batch: call direct_cluster_polygon_generation_batch with at least "
                                     "`batch` seeds first (caller-provided voxels replace the resident clusters)");
  if (cluster_xyz) h->resident_batch = 0;  // the voxels are packed into the handle's cluster storage: the generation's clusters are gone
  CHIP_TRY(hipSetDevice(h->cfg.device));
  Dev& D = h->D;
  HullDev& H = h->H;
  const size_t B = h->cfg.max_batch;
  if (!h->have_hull) {
    H.QX = 2 * D.max_x + 1; H.QY = 2 * D.max_y + 1; H.QZ = 2 * D.max_z + 1;
    H.half_words = (size_t)H.QY * H.QZ + (size_t)H.QX * H.QZ + (size_t)H.QX * H.QY;
    H.line_words = 2 * H.half_words;
    // all or nothing: a failed allocation frees what this call has allocated, so that a retry starts clean
    std::vector<void*> mine;
    hipError_t ae = hipSuccess;
    auto A = [&](auto pp, size_t bytes) {
      if (ae != hipSuccess) return;
      void* q = nullptr;
      ae = hipMalloc(&q, bytes);
      if (ae != hipSuccess) return;
      mine.push_back(q);
      *pp = (typename std::remove_pointer<decltype(pp)>::type)q;
    };
    A(&H.he, B * sizeof(HullElem));
    A(&H.lines, B * H.line_words * sizeof(int));
    A(&H.cand, B * 3 * hull::kCandCap * sizeof(int));
    A(&H.first, B * hull::kCandCap * sizeof(int));
    A(&H.isv, B * hull::kCandCap * sizeof(int));
    A(&H.raw, B * hull::kRawCap * 4 * sizeof(hull::i64));
    A(&H.sorted, B * hull::kRawCap * 4 * sizeof(hull::i64));
    A(&H.vq, B * hull::kCandCap * 3 * sizeof(int));
    if (ae != hipSuccess) {
      for (void* q : mine) (void)hipFree(q);
      H = HullDev{};
      return cfail(DIRECT_ERR_DEVICE, std::string("hull_planes_batch: scratch allocation: ") + hipGetErrorString(ae));
    }
    h->allocs.insert(h->allocs.end(), mine.begin(), mine.end());
    h->have_hull = true;
  }
  // device views of the outputs: the caller's pointers, or one staging block for host outputs
  const size_t nb = (size_t)batch;
  const size_t sz[8] = {nb * plane_capacity * 4 * sizeof(double), nb * plane_capacity * 4 * sizeof(long long), nb * sizeof(int32_t),
                        nb * vertex_capacity * 3 * sizeof(double), nb * sizeof(int32_t), nb * 3 * sizeof(double),
                        nb * sizeof(int32_t), nb * sizeof(int32_t)};
  void* user[8] = {planes, plane_int, n_planes, vertices, n_vertices, center, degenerate, rtn};
  void* dev[8];
  if (mem == DIRECT_MEM_HOST) {
    size_t total = 0, off[8];
    for (int i = 0; i < 8; i++) { off[i] = total; total += (sz[i] + 255) & ~(size_t)255; }
    if (total > h->hull_out_bytes) {
      if (h->hull_out) {
        CHIP_TRY(hipStreamSynchronize(h->stream));
        (void)hipFree(h->hull_out);
        h->hull_out = nullptr; h->hull_out_bytes = 0;
      }
      CHIP_TRY(hipMalloc(&h->hull_out, total));
      h->hull_out_bytes = total;
    }
    for (int i = 0; i < 8; i++) dev[i] = user[i] ? (char*)h->hull_out + off[i] : nullptr;
  } else {
    for (int i = 0; i < 8; i++) dev[i] = user[i];
  }
  const int32_t *sx = cluster_xyz, *sn = cluster_num;
  if (cluster_xyz && mem_in == DIRECT_MEM_HOST) {  // caller-provided voxels from the host: the filled prefix of every row,
                                                   // through the generation's own staging blocks (st_xyz, st_num)
    for (int b = 0; b < batch; b++) {
      const int n = cluster_num[b] < 0 ? 0 : (cluster_num[b] > D.ccap ? D.ccap : cluster_num[b]);
      if (n > 0)
        CHIP_TRY(hipMemcpyAsync(h->st_xyz + (size_t)b * D.ccap * 3, cluster_xyz + (size_t)b * D.ccap * 3, (size_t)n * 12,
                                hipMemcpyHostToDevice, h->stream));
    }
    CHIP_TRY(hipMemcpyAsync(h->st_num, cluster_num, nb * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    sx = h->st_xyz; sn = h->st_num;
  }
  CHIP_TRY(hipEventRecord(h->ev0, h->stream));
  CHIP_TRY(hipMemset2DAsync(H.lines, H.line_words * sizeof(int), 0x7f, H.half_words * sizeof(int), nb, h->stream));
  CHIP_TRY(hipMemset2DAsync(H.lines + H.half_words, H.line_words * sizeof(int), 0x80, H.half_words * sizeof(int), nb, h->stream));
  CHIP_TRY(hipMemsetAsync(H.he, 0, nb * sizeof(HullElem), h->stream));
  hipLaunchKernelGGL(k_hull_src, dim3(32, batch), dim3(256), 0, h->stream, D, H, batch, sx, sn);
  hipLaunchKernelGGL(k_hull_lines, dim3(batch), dim3(256), 0, h->stream, D, H);
  hipLaunchKernelGGL(k_hull_cand, dim3(batch), dim3(256), 0, h->stream, D, H);
  hipLaunchKernelGGL(k_hull_edges, dim3(64, batch), dim3(256), 0, h->stream, H);
  hipLaunchKernelGGL(k_hull_finish, dim3(batch), dim3(256), 0, h->stream, H, resolution, map_lower[0], map_lower[1], map_lower[2],
                     plane_capacity, vertex_capacity, (double*)dev[0], (long long*)dev[1], (int32_t*)dev[2], (double*)dev[3],
                     (int32_t*)dev[4], (double*)dev[5], (int32_t*)dev[6], (int32_t*)dev[7]);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipEventRecord(h->ev1, h->stream);
  h->timed = e == hipSuccess;
  if (mem == DIRECT_MEM_HOST)
    for (int i = 0; i < 8 && e == hipSuccess; i++)
      if (user[i]) e = hipMemcpyAsync(user[i], dev[i], sz[i], hipMemcpyDeviceToHost, h->stream);
  hipError_t e2 = hipStreamSynchronize(h->stream);
  if (e != hipSuccess || e2 != hipSuccess)
    return cfail(DIRECT_ERR_DEVICE, std::string("hull_planes_batch: ") + hipGetErrorString(e != hipSuccess ? e : e2));
  return DIRECT_OK;
}

direct_status_t direct_cluster_set_stream(direct_cluster_handle_t h, void* hip_stream) {
  if (!h) return DIRECT_ERR_INVALID;
  h->stream = (hipStream_t)hip_stream;  // every copy, launch and event of the handle is enqueued on it from now on
  return DIRECT_OK;
}

direct_status_t direct_cluster_last_ms(direct_cluster_handle_t h, float* ms) {
  if (!h || !ms) return cfail(DIRECT_ERR_INVALID, "null argument");
  if (!h->timed) return cfail(DIRECT_ERR_INVALID, "nothing has been timed yet");
  CHIP_TRY(hipEventSynchronize(h->ev1));
  CHIP_TRY(hipEventElapsedTime(ms, h->ev0, h->ev1));
  return DIRECT_OK;
}

}  // extern "C"
