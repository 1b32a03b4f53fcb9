// Output sampling: the step right after the hot path.  Batched restatement of the
// sampling loops of the caller's visualisation / audit helpers (teach_repeat_planner.cpp:1380-1394,
// 1440-1455, 1493-1508, 1551-1566) on top of Bernstein::getPosFromBezier / getVel / getAcc
// (global_planner/include/global_planner/utils/bezier_base.h:77-127):
//
//   for every segment i:  for (double t = 0.0; t < 1.0; t += dt / T_i)
//       pos = T_i * sum_j C(5,j) c_ij t^j (1-t)^(5-j)                   c_ij: time-scaled control points
//       vel =        sum_j C(4,j) 5 (c_i,j+1 - c_ij) t^j (1-t)^(4-j)
//       acc = 1/T_i  sum_j C(3,j) 20 (c_i,j+2 - 2 c_i,j+1 + c_ij) t^j (1-t)^(3-j)
//       traj_len += |pos - previous pos|
//
// One wavefront per trajectory; lane l of a chunk evaluates sample l.  The NUMBER of samples per segment
// is the reference's: its loop accumulates t += step in floating point, so the count is decided by the
// rounded recurrence, not by ceil(1/step).  The recurrence drifts from k*step by at most k ulp, so when
// no k*step lies within that margin of 1.0 the count follows from a multiplication (fast path, sample
// times k*step); otherwise the segment replays the exact recurrence (lane l performs the l sequential
// additions).  The kernel streams (bez, T) in and 9 words per sample out: it is HBM-write bound.
#pragma once
#include <hip/hip_runtime.h>

namespace direct {

template <typename St>
struct SampleArgs {
  int batch, nmax, capacity, derivs;
  const int32_t* n_seg;
  const St* bez;
  const St* T;
  double dt;
  int32_t* count;
  int32_t* seg_first;
  St* pos;
  St* vel;
  St* acc;
  St* length;
  St* vmax;
  St* amax;
  // containment audit (optional)
  int pmax;
  const int32_t* n_planes;
  const St* planes;
  St* cmax;
};

// one 12- / 24-byte store per lane: consecutive lanes write consecutive bytes (global_store_dwordx3 / x4 + x2)
template <typename St> struct Vec3;
template <> struct Vec3<float> { typedef float3 type; };
template <> struct Vec3<double> { typedef double3 type; };
template <typename St>
__device__ __forceinline__ void store3(St* dst, double x, double y, double z) {
  typename Vec3<St>::type v;
  v.x = (St)x; v.y = (St)y; v.z = (St)z;
  *reinterpret_cast<typename Vec3<St>::type*>(dst) = v;
}

__device__ __forceinline__ double sample_readlane(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

template <typename St>
__global__ __launch_bounds__(64) void k_sample(SampleArgs<St> A) {
  const int b = blockIdx.x, lane = threadIdx.x;
  // n_seg is clamped to the array extent: with device-resident inputs nothing has validated it on the host
  const int N = min(max(A.n_seg[b], 0), A.nmax);
  const St* Tb = A.T + (size_t)b * A.nmax;
  // a negative duration aborts the reference's loop before anything is published (TRP:1552-1555)
  int neg = 0;
  for (int i = lane; i < N; i += 64) neg |= (Tb[i] < (St)0) ? 1 : 0;
  if (__any(neg)) {
    if (lane == 0) {
      A.count[b] = -1;
      if (A.length) A.length[b] = (St)0;
      if (A.vmax) A.vmax[b] = (St)0;
      if (A.amax) A.amax[b] = (St)0;
      if (A.cmax) A.cmax[b] = (St)0;
    }
    return;
  }
  int base = 0;
  double len = 0.0, vm = 0.0, am = 0.0, cm = -1.0e300;
  const bool audit = A.cmax != nullptr;
  double px = 0.0, py = 0.0, pz = 0.0;  // last point of the previous chunk
  const size_t ob = (size_t)b * A.capacity;
  for (int i = 0; i < N; i++) {
    const double Ti = (double)Tb[i];
    double step = A.dt / Ti;
    if (!(step > 0.0)) step = 2.0;  // the reference would never leave its loop (step 0 / NaN): one sample instead
    const St* c = A.bez + ((size_t)b * A.nmax + i) * 18;
    double cf[18];
#pragma unroll
    for (int q = 0; q < 18; q++) cf[q] = (double)c[q];
    if (A.seg_first && lane == 0) A.seg_first[(size_t)b * A.nmax + i] = base;
    // fast path: count = the smallest k with k * step >= 1, provided no candidate is within k ulp of 1.0
    int nfast = -1;
    if (step < 1.0 && step > 1.0e-7) {
      const double kf = floor(1.0 / step);
      int cnt = -1, ambiguous = 0;
      for (int c = -1; c <= 2; c++) {
        const double k = kf + (double)c;
        if (k < 1.0) continue;
        const double p = k * step, margin = (k + 4.0) * 2.220446049250313e-16;
        if (fabs(p - 1.0) <= margin) ambiguous = 1;
        if (cnt < 0 && p >= 1.0) cnt = (int)k;
      }
      if (!ambiguous && cnt > 0) nfast = cnt;
    } else if (step >= 1.0) {
      nfast = 1;
    }
    const double invT = 1.0 / Ti;
    double t_carry = 0.0;
    int kbase = 0;
    while (true) {
      double t;
      bool valid;
      if (nfast >= 0) {
        t = (double)(kbase + lane) * step;
        valid = kbase + lane < nfast;
      } else {
        t = t_carry;
#pragma unroll 8
        for (int q = 0; q < 63; q++) t = (q < lane) ? t + step : t;
        valid = t < 1.0;
      }
      const int n = __popcll(__ballot(valid));
      const double u = 1.0 - t;
      double tp[6], up[6];
      tp[0] = 1.0; up[0] = 1.0;
#pragma unroll
      for (int j = 1; j < 6; j++) { tp[j] = tp[j - 1] * t; up[j] = up[j - 1] * u; }
      const double C5[6] = {1, 5, 10, 10, 5, 1}, C4[5] = {1, 4, 6, 4, 1}, C3[4] = {1, 3, 3, 1};
      double p[3], v[3] = {0, 0, 0}, a[3] = {0, 0, 0};
#pragma unroll
      for (int d = 0; d < 3; d++) {
        double r = 0.0;
#pragma unroll
        for (int j = 0; j < 6; j++) r += C5[j] * cf[d * 6 + j] * tp[j] * up[5 - j];
        p[d] = Ti * r;
        if (A.derivs >= 1) {
          double rv = 0.0;
#pragma unroll
          for (int j = 0; j < 5; j++) rv += C4[j] * 5.0 * (cf[d * 6 + j + 1] - cf[d * 6 + j]) * tp[j] * up[4 - j];
          v[d] = rv;
        }
        if (A.derivs >= 2) {
          double ra = 0.0;
#pragma unroll
          for (int j = 0; j < 4; j++)
            ra += C3[j] * 5.0 * 4.0 * (cf[d * 6 + j + 2] - 2.0 * cf[d * 6 + j + 1] + cf[d * 6 + j]) * tp[j] * up[3 - j];
          a[d] = ra * invT;
        }
      }
      // distance to the previous sample (lane - 1, or the carried point for lane 0)
      double qx = __shfl_up(p[0], 1, 64), qy = __shfl_up(p[1], 1, 64), qz = __shfl_up(p[2], 1, 64);
      if (lane == 0) { qx = px; qy = py; qz = pz; }
      const double dx = qx - p[0], dy = qy - p[1], dz = qz - p[2];
      double dist = sqrt(dx * dx + dy * dy + dz * dz);
      if (!valid || (base == 0 && lane == 0)) dist = 0.0;  // the first point of the trajectory has no predecessor
      for (int o = 32; o > 0; o >>= 1) dist += __shfl_xor(dist, o, 64);
      len += dist;
      const int idx = base + lane;
      if (audit && valid) {  // the planes of a segment are the same for every lane: broadcast loads
        const int np = min(max(A.n_planes[(size_t)b * A.nmax + i], 0), A.pmax);
        const St* pq = A.planes + ((size_t)b * A.nmax + i) * A.pmax * 4;
        for (int q = 0; q < np; q++)
          cm = fmax(cm, (double)pq[4 * q] * p[0] + (double)pq[4 * q + 1] * p[1] + (double)pq[4 * q + 2] * p[2] + (double)pq[4 * q + 3]);
      }
      if (valid) {
        if (A.derivs >= 1) vm = fmax(vm, fmax(fabs(v[0]), fmax(fabs(v[1]), fabs(v[2]))));
        if (A.derivs >= 2) am = fmax(am, fmax(fabs(a[0]), fmax(fabs(a[1]), fabs(a[2]))));
        if (idx < A.capacity) {
          store3(A.pos + (ob + idx) * 3, p[0], p[1], p[2]);
          if (A.derivs >= 1 && A.vel) store3(A.vel + (ob + idx) * 3, v[0], v[1], v[2]);
          if (A.derivs >= 2 && A.acc) store3(A.acc + (ob + idx) * 3, a[0], a[1], a[2]);
        }
      }
      if (n > 0) {
        px = sample_readlane(p[0], n - 1); py = sample_readlane(p[1], n - 1); pz = sample_readlane(p[2], n - 1);
      }
      base += n;
      kbase += 64;
      if (n < 64) break;
      t_carry = sample_readlane(t, 63) + step;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    vm = fmax(vm, __shfl_xor(vm, o, 64));
    am = fmax(am, __shfl_xor(am, o, 64));
    cm = fmax(cm, __shfl_xor(cm, o, 64));
  }
  if (lane == 0) {
    A.count[b] = base;
    if (A.length) A.length[b] = (St)len;
    if (A.vmax) A.vmax[b] = (St)vm;
    if (A.amax) A.amax[b] = (St)am;
    if (A.cmax) A.cmax[b] = (St)cm;
  }
}

}  // namespace direct
