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
// additions).  The kernel streams (bez, T) in and 9 words per sample out: it is HBM-write bound by design;
// the evaluation itself is 36 fused multiply-adds per sample (monomial form, see below).
#pragma once
#include <hip/hip_runtime.h>

namespace direct {

template <typename St>
struct SampleArgs {
  int batch, nmax, capacity, derivs;
  const int32_t* n_seg;
  const St* bez;
  const St* T;
  double dt, inv_dt;  // inv_dt = 1 / dt (host): the sample-count estimate T / dt then needs no division per segment
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

// lane l receives lane l - 1's value, lane 0 receives `first`: a whole-wave shift on the VALU's DPP path (two moves per
// double) instead of __shfl_up's two ds_bpermute round trips through the LDS pipe
__device__ __forceinline__ double sample_shift_up(double v, double first) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(first), __double2loint(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(first), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
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
  // Everything a segment reads is the same for the 64 lanes (its duration, its 18 control points): read through the
  // CONSTANT address space, i.e. by scalar loads.  Vector loads would share the in-order vmcnt counter with the sample
  // stores: the wait for a segment's coefficients then also waits until every store of the previous segment has drained
  // to HBM - the kernel ran at 58 % of its own instruction count's time and 64 % of the pure-write rate of its store pattern
  // (tools/hbm_calib: 5.6 TB/s).  (The arrays are written by earlier kernels only.)
  typedef __attribute__((address_space(4))) const St CSt;
  CSt* Tb = (CSt*)(A.T + (size_t)b * A.nmax);
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
    CSt* c = (CSt*)(A.bez + ((size_t)b * A.nmax + i) * 18);
    double cf[18];
#pragma unroll
    for (int q = 0; q < 18; q++) cf[q] = (double)c[q];
    if (A.seg_first && lane == 0) A.seg_first[(size_t)b * A.nmax + i] = base;
    // fast path: count = the smallest k with k * step >= 1, provided no candidate is within k ulp of 1.0
    int nfast = -1;
    if (step < 1.0 && step > 1.0e-7) {
      const double kf = floor(Ti * A.inv_dt);  // ~ 1 / step: the candidates around it are examined below
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
    const double invT = frcp(Ti);  // (1-2 ulp: scales the acceleration samples only)
    // The segment's three Bernstein polynomials in the monomial basis, once per segment: with D_i the i-th forward
    // difference of the control points at 0, the degree-5 position polynomial is sum_i C(5,i) D_i t^i, the reference's
    // velocity polynomial (control points 5 (c_j+1 - c_j), degree 4) is sum_i 5 C(4,i) D_i+1 t^i and its acceleration
    // polynomial (control points 20 (c_j+2 - 2 c_j+1 + c_j), degree 3) is sum_i 20 C(3,i) D_i+2 t^i - ONE difference
    // table (15 subtractions per axis) serves all three, and a sample is 5 + 4 + 3 fused multiply-adds per axis (Horner on
    // t in [0, 1]) plus twelve products of t instead of 15 basis products and ~100 multiply-adds.  The kernel was bound by exactly that arithmetic
    // (480 VALU instructions per chunk of 64 samples, 39 % of the HBM roofline).  Against the reference's Bernstein
    // sums this differs at rounding level (measured < 1e-13 of the largest value; the parity tests ask for 1e-12).
    // Only the difference table is kept per segment (18 doubles): the binomial factors ride on the sample's t
    // (s_k = D_k + (C_k+1 / C_k) t s_k+1, value = C_0 s_0), so the kernel stays below 128 VGPRs (four waves per SIMD).
    double D[3][6];
#pragma unroll
    for (int d = 0; d < 3; d++) {
      double w[6];
#pragma unroll
      for (int j = 0; j < 6; j++) w[j] = cf[d * 6 + j];
      D[d][0] = w[0];
#pragma unroll
      for (int lvl = 1; lvl < 6; lvl++) {
#pragma unroll
        for (int j = 0; j + lvl < 6; j++) w[j] = w[j + 1] - w[j];
        D[d][lvl] = w[0];
      }
    }
    const double vs = 5.0, as = 20.0 * invT;
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
      // t times the ratios of consecutive binomials: C(5,.) = 1 5 10 10 5 1, C(4,.) = 1 4 6 4 1, C(3,.) = 1 3 3 1
      const double p5[5] = {5.0 * t, 2.0 * t, t, 0.5 * t, 0.2 * t};
      const double p4[4] = {4.0 * t, 1.5 * t, (2.0 / 3.0) * t, 0.25 * t};
      const double p3[3] = {3.0 * t, t, (1.0 / 3.0) * t};
      double p[3], v[3] = {0, 0, 0}, a[3] = {0, 0, 0};
#pragma unroll
      for (int d = 0; d < 3; d++) {
        double r = D[d][5];
#pragma unroll
        for (int q = 4; q >= 0; q--) r = fma(r, p5[q], D[d][q]);
        p[d] = Ti * r;
        if (A.derivs >= 1) {
          double rv = D[d][5];
#pragma unroll
          for (int q = 3; q >= 0; q--) rv = fma(rv, p4[q], D[d][q + 1]);
          v[d] = vs * rv;
        }
        if (A.derivs >= 2) {
          double ra = D[d][5];
#pragma unroll
          for (int q = 2; q >= 0; q--) ra = fma(ra, p3[q], D[d][q + 2]);
          a[d] = as * ra;
        }
      }
      // distance to the previous sample (lane - 1, or the carried point for lane 0)
      const double qx = sample_shift_up(p[0], px), qy = sample_shift_up(p[1], py), qz = sample_shift_up(p[2], pz);
      const double dx = qx - p[0], dy = qy - p[1], dz = qz - p[2];
      const double d2 = dx * dx + dy * dy + dz * dz;
      double dist = d2 > 0.0 ? d2 * frsq(d2) : 0.0;  // sqrt by the hardware seed + two Newton steps (1-2 ulp)
      if (!valid || (base == 0 && lane == 0)) dist = 0.0;  // the first point of the trajectory has no predecessor
      len += dist;  // per lane; ONE wave sum per trajectory at the end (a sum per chunk was 15 % of the chunk's instructions)
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
  len = wave_sum_d(len);  // DPP reduction on the VALU (ddp_wave.h)
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
