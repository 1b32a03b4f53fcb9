// BASELINE-label model on gfx950: batched Gauss-Newton iLQR for a 12-state / 4-control quadrotor
// (include/direct_quad.h).  NO REFERENCE COUNTERPART: ntu-caokun/DIRECT has no such optimiser;
// the outer-loop conventions are the reference's where they carry over (ddp_optimizer.cpp:297-310, 452-474, 666-670).
//
// One 64-lane wavefront (one workgroup) per trajectory, as on the main path.  Per knot everything lives in LDS:
// V (12x12), A = I + dt f_x (12x12), B = dt f_u (12x4), VA, VB, the Q-function blocks and the gains; the 144-entry
// products run one entry per lane in three passes of 12 FMAs.  Arithmetic is double for both storage types (the
// value recursion over 100 knots), like the main path.  HBM per knot-iteration: x (12) + u (4) in, gains K (48) +
// k (4) out and in again, the accepted x, u out: 152 words.  The sweeps are latency-bound (seven
// LDS hand-offs per backward knot); with 608 B per knot-iteration nothing here is near the HBM roofline.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/direct_quad.h"

namespace {

constexpr int NX = DIRECT_QUAD_NX, NU = DIRECT_QUAD_NU;

// Phase boundary inside a one-wave workgroup: LDS operations of one wave execute in order, so all that is needed is
// that the compiler keeps them in order (same reasoning as WSYNC in ddp_wave.h).  __syncthreads() would add an
// s_barrier and s_waitcnt vmcnt(0), i.e. wait for every outstanding HBM access at every phase.
#define QSYNC()                      \
  do {                               \
    asm volatile("" ::: "memory");   \
    __builtin_amdgcn_wave_barrier(); \
  } while (0)

// 1 / sqrt(x) and 1 / x: hardware seed + two Newton steps (full double accuracy), a dozen instructions
// instead of the ~40 of an IEEE division or square root
__device__ __forceinline__ double q_rsq(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}
__device__ __forceinline__ double q_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

struct QConst {
  double m, g, J[3], dt, q[NX], r[NU], qf[NX], reg_base, tol;
  double inv_m, inv_J[3], kJ[3];  // 1 / m, 1 / J, (J2-J1)/J0, (J0-J2)/J1, (J1-J0)/J2
  int iter_max, fixed_iters;
};

struct QState {  // per trajectory
  double cost;
  int reg, step, fp_failed, bp_failed, iter, done, fwd_passes, cur;
};

template <typename St>
struct QBatch {
  int B, N;
  const St* x0;  // [B][12]
  const St* xg;  // [B][12]
  St* X[2];      // [B][N+1][12]
  St* U[2];      // [B][N][4]
  St* K;         // [B][N][4][12]
  St* kf;        // [B][N][4]
  double* Tg[2]; // [B][N][8]: sin / cos of the Euler angles and 1 / cos(theta) of X[.][k], written by whoever rolled the knot
  QState* st;
  QConst c;
};

struct QLds {
  double V[144], VA[144], Qxx[144];
  double VB[48], Qux[48], Kk[48], QuuK[48];
  double Quu[16], Vx[12], Qx[12], xk[12], xn[12], dx[12], xnext[12];
  double Qu[4], kk[4], Quuk[4], uk[4], un[4], trig[8], S[60], part[16];
  QState st;
};

__device__ __forceinline__ double wsum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// entries of f_x / f_u that are not constants, from trig[] = {s_phi, c_phi, s_th, c_th, s_psi, c_psi}:
// S[0..8] d vdot / d euler, S[9..17] d eulerdot / d euler, S[18..26] W, S[27..35] d omegadot / d omega, S[36..38] R e3 / m
__device__ __forceinline__ double special_entry(const QConst& c, const double* x, const double* u, const double* t, int e) {
  const double sph = t[0], cph = t[1], sth = t[2], cth = t[3], sps = t[4], cps = t[5];
  const double icth = t[6];  // 1 / cos(theta), once per knot (trig_lanes)
  const double a = u[0] * c.inv_m, tth = sth * icth, sec2 = icth * icth;
  const double k0 = c.kJ[0], k1 = c.kJ[1], k2 = c.kJ[2];
  switch (e) {
    case 0: return a * (-sph * sth * cps + cph * sps);
    case 1: return a * (cph * cth * cps);
    case 2: return a * (-cph * sth * sps + sph * cps);
    case 3: return a * (-sph * sth * sps - cph * cps);
    case 4: return a * (cph * cth * sps);
    case 5: return a * (cph * sth * cps + sph * sps);
    case 6: return a * (-sph * cth);
    case 7: return a * (-cph * sth);
    case 8: return 0.0;
    case 9: return cph * tth * x[10] - sph * tth * x[11];
    case 10: return (sph * x[10] + cph * x[11]) * sec2;
    case 11: return 0.0;
    case 12: return -sph * x[10] - cph * x[11];
    case 13: return 0.0;
    case 14: return 0.0;
    case 15: return (cph * x[10] - sph * x[11]) * icth;
    case 16: return (sph * x[10] + cph * x[11]) * sth * sec2;
    case 17: return 0.0;
    case 18: return 1.0;
    case 19: return sph * tth;
    case 20: return cph * tth;
    case 21: return 0.0;
    case 22: return cph;
    case 23: return -sph;
    case 24: return 0.0;
    case 25: return sph * icth;
    case 26: return cph * icth;
    case 27: return 0.0;
    case 28: return -k0 * x[11];
    case 29: return -k0 * x[10];
    case 30: return -k1 * x[11];
    case 31: return 0.0;
    case 32: return -k1 * x[9];
    case 33: return -k2 * x[10];
    case 34: return -k2 * x[9];
    case 35: return 0.0;
    case 36: return (cph * sth * cps + sph * sps) * c.inv_m;
    case 37: return (cph * sth * sps - sph * cps) * c.inv_m;
    default: return (cph * cth) * c.inv_m;
  }
}

// component l of f(x, u) from trig[]
__device__ __forceinline__ double dyn_entry(const QConst& c, const double* x, const double* u, const double* t, int l) {
  const double sph = t[0], cph = t[1], sth = t[2], cth = t[3], sps = t[4], cps = t[5];
  const double icth = t[6];
  const double a = u[0] * c.inv_m, tth = sth * icth;
  switch (l) {
    case 0: return x[3];
    case 1: return x[4];
    case 2: return x[5];
    case 3: return a * (cph * sth * cps + sph * sps);
    case 4: return a * (cph * sth * sps - sph * cps);
    case 5: return a * (cph * cth) - c.g;
    case 6: return x[9] + sph * tth * x[10] + cph * tth * x[11];
    case 7: return cph * x[10] - sph * x[11];
    case 8: return (sph * x[10] + cph * x[11]) * icth;
    case 9: return (u[1] - (c.J[2] - c.J[1]) * x[10] * x[11]) * c.inv_J[0];
    case 10: return (u[2] - (c.J[0] - c.J[2]) * x[9] * x[11]) * c.inv_J[1];
    default: return (u[3] - (c.J[1] - c.J[0]) * x[9] * x[10]) * c.inv_J[2];
  }
}

// sin and cos in double without the library's general-purpose range reduction (a quarter of the sweep time went
// into sincos()): Cody-Waite reduction by pi/2 (exact enough for |x| < 1e5 rad; Euler angles are O(1)) and the
// fdlibm kernel polynomials on [-pi/4, pi/4]; 1-2 ulp, ~30 FMAs for both.
__device__ __forceinline__ void sincos_fast(double x, double* sn, double* cs) {
  const double k = rint(x * 6.36619772367581382433e-01);  // 2 / pi
  double r = fma(-k, 1.57079632673412561417e+00, x);     // pi/2, high part (33 bits)
  r = fma(-k, 6.07710050650619224932e-11, r);            // pi/2, low part
  const double z = r * r;
  const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                    z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
  const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                    z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
  const double s0 = fma(r * z, ps, r);
  const double c0 = fma(z * z, pc, fma(-0.5, z, 1.0));
  const int q = (int)k & 3;
  const double s1 = (q & 1) ? c0 : s0, c1 = (q & 1) ? s0 : c0;
  *sn = (q & 2) ? -s1 : s1;
  *cs = ((q + 1) & 2) ? -c1 : c1;
}

// sin / cos of the three Euler angles of x[] into trig[0..5]: one evaluation per angle, on three lanes
__device__ __forceinline__ void trig_lanes(const double* x, double* trig, int lane) {
  if (lane < 3) {
    double s, c;
    sincos_fast(x[6 + lane], &s, &c);
    trig[2 * lane] = s;
    trig[2 * lane + 1] = c;
    if (lane == 1) trig[6] = q_rcp(c);  // 1 / cos(theta)
  }
}

// one knot of a roll-out from LDS state L.xn with input L.un: cost contribution (wave-uniform) and L.xnext
__device__ __forceinline__ double roll_knot(const QConst& c, QLds& L, const double* xg, int lane) {
  double part = 0.0;
  if (lane < NX) {
    const double d = L.xn[lane] - xg[lane];
    part = c.q[lane] * d * d;
    L.xnext[lane] = L.xn[lane] + c.dt * dyn_entry(c, L.xn, L.un, L.trig, lane);
  } else if (lane < NX + NU) {
    const int i = lane - NX;
    const double d = L.un[i] - (i == 0 ? c.m * c.g : 0.0);
    part = c.r[i] * d * d;
  }
  return 0.5 * c.dt * wsum(part);
}

template <typename St>
__device__ void q_begin(const QBatch<St>& Q, QLds& L, int b, int lane) {
  const int N = Q.N;
  __shared__ double xg[NX];
  if (lane < NX) {
    xg[lane] = (double)Q.xg[(size_t)b * NX + lane];
    L.xn[lane] = (double)Q.x0[(size_t)b * NX + lane];
  }
  if (lane < NU) L.un[lane] = lane == 0 ? Q.c.m * Q.c.g : 0.0;  // the hover input
  QSYNC();
  double cost = 0.0;
  St* X = Q.X[0] + (size_t)b * (N + 1) * NX;
  St* U = Q.U[0] + (size_t)b * N * NU;
  for (int k = 0; k < N; k++) {
    if (lane < NX) L.xn[lane] = (double)(St)L.xn[lane];  // the stored iterate is the iterate
    QSYNC();
    trig_lanes(L.xn, L.trig, lane);
    QSYNC();
    if (lane < 7) Q.Tg[0][((size_t)b * N + k) * 8 + lane] = L.trig[lane];
    cost += roll_knot(Q.c, L, xg, lane);
    if (lane < NX) X[(size_t)k * NX + lane] = (St)L.xn[lane];
    if (lane < NU) U[(size_t)k * NU + lane] = (St)L.un[lane];
    QSYNC();
    if (lane < NX) L.xn[lane] = L.xnext[lane];
    QSYNC();
  }
  double part = 0.0;
  if (lane < NX) {
    const double xs = (double)(St)L.xn[lane];
    X[(size_t)N * NX + lane] = (St)xs;
    part = Q.c.qf[lane] * (xs - xg[lane]) * (xs - xg[lane]);
  }
  cost += 0.5 * wsum(part);
  if (lane == 0) {
    QState s;
    s.cost = cost; s.reg = 0; s.step = 0; s.fp_failed = 0; s.bp_failed = 0; s.iter = 0; s.done = 0; s.fwd_passes = 0; s.cur = 0;
    Q.st[b] = s;
  }
}

// backward sweep; returns 1 on success, 0 when a 4x4 LLT failed (the regulariser then grows, reference :297-310)
template <typename St>
__device__ int q_backward(const QBatch<St>& Q, QLds& L, int b, int lane, const double* xg) {
  const QConst& c = Q.c;
  const int N = Q.N, cur = L.st.cur;
  if (lane == 0) {
    int reg = L.st.reg;
    if (L.st.fp_failed || L.st.bp_failed) reg += 1;
    else if (L.st.step == 0) reg -= 1;
    else if (L.st.step > 3) reg += 1;
    L.st.reg = reg < 0 ? 0 : (reg > 24 ? 24 : reg);
  }
  QSYNC();
  double lam = 1.0;
  for (int q = 0; q < L.st.reg; q++) lam *= c.reg_base;
  lam -= 1.0;
  const St* X = Q.X[cur] + (size_t)b * (N + 1) * NX;
  const St* U = Q.U[cur] + (size_t)b * N * NU;
  for (int e = lane; e < 144; e += 64) L.V[e] = (e / 12 == e % 12) ? c.qf[e / 12] : 0.0;
  if (lane < 18) L.S[40 + lane] = (lane >= 9 && (lane - 9) / 3 == (lane - 9) % 3) ? 1.0 : 0.0;  // zero block, identity block
  if (lane < NX) L.Vx[lane] = c.qf[lane] * ((double)X[(size_t)N * NX + lane] - xg[lane]);
  QSYNC();
  // x_k / u_k are loaded one knot ahead into a register: with QSYNC nothing waits for HBM but the use of the value
  const int lxu = lane < NX + NU ? lane : NX + NU - 1;
  const St* src = lxu < NX ? X + lxu : U + (lxu - NX);
  const int stride = lxu < NX ? NX : NU;
  St pre = src[(size_t)(N - 1) * stride];
  // sin / cos of x_k's Euler angles were computed when the knot was rolled (same stored x_k, same function: same bits)
  const double* tsrc = Q.Tg[cur] + (size_t)b * N * 8 + (lane < 7 ? lane : 6);
  double tpre = tsrc[(size_t)(N - 1) * 8];
  for (int k = N - 1; k >= 0; k--) {
    if (lane < NX) L.xk[lane] = (double)pre;
    else if (lane < NX + NU) L.uk[lane - NX] = (double)pre;
    if (lane < 7) L.trig[lane] = tpre;
    pre = src[(size_t)(k > 0 ? k - 1 : 0) * stride];
    tpre = tsrc[(size_t)(k > 0 ? k - 1 : 0) * 8];
    QSYNC();
    if (lane < 39) L.S[lane] = special_entry(c, L.xk, L.uk, L.trig, lane);
    QSYNC();
    // A = I + dt F and B = dt G are never formed.  F = f_x has four 3x3 blocks of non-constant entries (S[0..35]) and an
    // identity block (rows 0..2, columns 3..5); G = f_u has the column S[36..38] (rows 3..5) and 1 / J (rows 9..11).
    // With S[40..48] = 0 and S[49..57] = I3 every product against F is "two 3x3 blocks per column group":
    //   (V F)[r][3g + c]  = sum_j V[r][la(g) + j] S[sa(g) + 3j + c] + sum_j V[r][lb(g) + j] S[sb(g) + 3j + c]
    //   (F' M)[3g + c][q] = sum_j S[sa(g) + 3j + c] M[la(g) + j][q] + sum_j S[sb(g) + 3j + c] M[lb(g) + j][q]
    // six terms instead of twelve, and no 144-entry matrix to build per knot.
    for (int e = lane; e < 144; e += 64) {  // VA = V A = V + dt V F
      const int r = e / 12, cc = e % 12, g = cc / 3, c3 = cc % 3;
      const int la = g == 2 ? 3 : (g == 3 ? 6 : 0), lb = g == 2 ? 6 : (g == 3 ? 9 : 0);
      const int sa = g == 0 ? 40 : (g == 1 ? 49 : (g == 2 ? 0 : 18)), sb = g == 2 ? 9 : (g == 3 ? 27 : 40);
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 3; j++) acc += L.V[r * 12 + la + j] * L.S[sa + 3 * j + c3] + L.V[r * 12 + lb + j] * L.S[sb + 3 * j + c3];
      L.VA[e] = L.V[e] + c.dt * acc;
    }
    if (lane < 48) {  // VB = V B
      const int r = lane / 4, cc = lane % 4;
      double acc;
      if (cc == 0) acc = L.V[r * 12 + 3] * L.S[36] + L.V[r * 12 + 4] * L.S[37] + L.V[r * 12 + 5] * L.S[38];
      else acc = L.V[r * 12 + 8 + cc] * c.inv_J[cc - 1];
      L.VB[lane] = c.dt * acc;
    }
    QSYNC();
    for (int e = lane; e < 144; e += 64) {  // Qxx = lxx + A' VA = lxx + VA + dt F' VA
      const int r = e / 12, cc = e % 12, g = r / 3, r3 = r % 3;
      const int la = g == 2 ? 3 : (g == 3 ? 6 : 0), lb = g == 2 ? 6 : (g == 3 ? 9 : 0);
      const int sa = g == 0 ? 40 : (g == 1 ? 49 : (g == 2 ? 0 : 18)), sb = g == 2 ? 9 : (g == 3 ? 27 : 40);
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 3; j++) acc += L.S[sa + 3 * j + r3] * L.VA[(la + j) * 12 + cc] + L.S[sb + 3 * j + r3] * L.VA[(lb + j) * 12 + cc];
      L.Qxx[e] = L.VA[e] + c.dt * acc + (r == cc ? c.dt * c.q[r] : 0.0);
    }
    if (lane < 48) {  // Qux = B' VA
      const int i = lane / 12, j = lane % 12;
      double acc;
      if (i == 0) acc = L.S[36] * L.VA[3 * 12 + j] + L.S[37] * L.VA[4 * 12 + j] + L.S[38] * L.VA[5 * 12 + j];
      else acc = c.inv_J[i - 1] * L.VA[(8 + i) * 12 + j];
      L.Qux[lane] = c.dt * acc;
    } else {  // Quu = luu + B' VB
      const int e = lane - 48, i = e / 4, j = e % 4;
      double acc;
      if (i == 0) acc = L.S[36] * L.VB[3 * 4 + j] + L.S[37] * L.VB[4 * 4 + j] + L.S[38] * L.VB[5 * 4 + j];
      else acc = c.inv_J[i - 1] * L.VB[(8 + i) * 4 + j];
      L.Quu[e] = c.dt * acc + (i == j ? c.dt * c.r[i] : 0.0);
    }
    if (lane < NX) {  // Qx = lx + A' Vx = lx + Vx + dt F' Vx
      const int g = lane / 3, r3 = lane % 3;
      const int la = g == 2 ? 3 : (g == 3 ? 6 : 0), lb = g == 2 ? 6 : (g == 3 ? 9 : 0);
      const int sa = g == 0 ? 40 : (g == 1 ? 49 : (g == 2 ? 0 : 18)), sb = g == 2 ? 9 : (g == 3 ? 27 : 40);
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 3; j++) acc += L.S[sa + 3 * j + r3] * L.Vx[la + j] + L.S[sb + 3 * j + r3] * L.Vx[lb + j];
      L.Qx[lane] = c.dt * c.q[lane] * (L.xk[lane] - xg[lane]) + L.Vx[lane] + c.dt * acc;
    } else if (lane < NX + NU) {  // Qu = lu + B' Vx
      const int i = lane - NX;
      double acc;
      if (i == 0) acc = L.S[36] * L.Vx[3] + L.S[37] * L.Vx[4] + L.S[38] * L.Vx[5];
      else acc = c.inv_J[i - 1] * L.Vx[8 + i];
      L.Qu[i] = c.dt * c.r[i] * (L.uk[i] - (i == 0 ? c.m * c.g : 0.0)) + c.dt * acc;
    }
    QSYNC();
    // LLT of Quu + lam I: computed by every lane (wave-uniform), then one right-hand side per lane
    double Lm[NU][NU], ri[NU];  // ri = 1 / L_jj: the factor and both solves multiply instead of dividing
    int ok = 1;
#pragma unroll
    for (int j = 0; j < NU; j++) {
      double d = L.Quu[j * NU + j] + lam;
#pragma unroll
      for (int l = 0; l < j; l++) d -= Lm[j][l] * Lm[j][l];
      if (d <= 0.0) ok = 0;
      ri[j] = q_rsq(d);
#pragma unroll
      for (int i = j + 1; i < NU; i++) {
        double v = L.Quu[i * NU + j];
#pragma unroll
        for (int l = 0; l < j; l++) v -= Lm[i][l] * Lm[j][l];
        Lm[i][j] = v * ri[j];
      }
    }
    if (!ok) {
      if (lane == 0) L.st.bp_failed = 1;
      QSYNC();
      return 0;
    }
    if (lane <= NX) {  // column 0: Qu -> k, columns 1..12: Qux[:, c-1] -> K[:, c-1]
      double y[NU], z[NU];
#pragma unroll
      for (int i = 0; i < NU; i++) {
        double v = lane == 0 ? L.Qu[i] : L.Qux[i * NX + lane - 1];
#pragma unroll
        for (int l = 0; l < i; l++) v -= Lm[i][l] * y[l];
        y[i] = v * ri[i];
      }
#pragma unroll
      for (int i = NU - 1; i >= 0; i--) {
        double v = y[i];
#pragma unroll
        for (int l = i + 1; l < NU; l++) v -= Lm[l][i] * z[l];
        z[i] = v * ri[i];
      }
#pragma unroll
      for (int i = 0; i < NU; i++) {
        if (lane == 0) L.kk[i] = -z[i];
        else L.Kk[i * NX + lane - 1] = -z[i];
      }
    }
    QSYNC();
    if (lane < 48) {  // gains to HBM; W = Quu K + 2 Qux
      Q.K[((size_t)b * N + k) * 48 + lane] = (St)L.Kk[lane];
      const int i = lane / 12, j = lane % 12;
      double acc = 2.0 * L.Qux[lane];
#pragma unroll
      for (int l = 0; l < NU; l++) acc += L.Quu[i * NU + l] * L.Kk[l * NX + j];
      L.QuuK[lane] = acc;
    } else if (lane < 52) {  // w = Quu k + Qu
      const int i = lane - 48;
      Q.kf[((size_t)b * N + k) * NU + i] = (St)L.kk[i];
      double acc = L.Qu[i];
#pragma unroll
      for (int l = 0; l < NU; l++) acc += L.Quu[i * NU + l] * L.kk[l];
      L.Quuk[i] = acc;
    }
    QSYNC();
    // Value update with the UNREGULARISED Quu (reference :626-628), symmetrised:
    //   Vn = Qxx + K' Quu K + K' Qux + Qux' K,  V = (Vn + Vn') / 2 = (Qxx + Qxx') / 2 + (K' W + W' K) / 2,  W = Quu K + 2 Qux
    // (Quu is symmetric up to rounding) - eight products per entry, written straight into V: nothing reads V between
    // the V A product at the top of the knot and here.
    for (int e = lane; e < 144; e += 64) {
      const int i = e / 12, j = e % 12;
      double acc = L.Qxx[e] + L.Qxx[j * 12 + i];
#pragma unroll
      for (int l = 0; l < NU; l++) acc += L.Kk[l * NX + i] * L.QuuK[l * NX + j] + L.Kk[l * NX + j] * L.QuuK[l * NX + i];
      L.V[e] = 0.5 * acc;
    }
    if (lane < NX) {  // Vx = Qx + K' (Quu k + Qu) + Qux' k; Vx was last read for Qx / Qu above
      double acc = L.Qx[lane];
#pragma unroll
      for (int l = 0; l < NU; l++) acc += L.Kk[l * NX + lane] * L.Quuk[l] + L.Qux[l * NX + lane] * L.kk[l];
      L.Vx[lane] = acc;
    }
    QSYNC();
  }
  if (lane == 0) L.st.bp_failed = 0;
  QSYNC();
  return 1;
}

template <typename St>
__device__ void q_forward(const QBatch<St>& Q, QLds& L, int b, int lane, const double* xg) {
  const QConst& c = Q.c;
  const int N = Q.N, cur = L.st.cur, nxt = 1 - cur;
  const St* X = Q.X[cur] + (size_t)b * (N + 1) * NX;
  const St* U = Q.U[cur] + (size_t)b * N * NU;
  St* Xt = Q.X[nxt] + (size_t)b * (N + 1) * NX;
  St* Ut = Q.U[nxt] + (size_t)b * N * NU;
  const double cost_old = L.st.cost;
  for (int step = 0; step < 11; step++) {
    double alpha = 1.0;
    for (int q = 0; q < step; q++) alpha *= 0.5;
    if (lane < NX) L.xn[lane] = (double)X[lane];
    QSYNC();
    double cost = 0.0;
    // the old iterate and the gains of knot k + 1 are loaded while knot k is evaluated (clamped indices: no branches)
    const int l12 = lane < NX ? lane : NX - 1, l52 = lane < 52 ? lane : 51, l4 = lane >= 48 && lane < 52 ? lane - 48 : 0;
    const St* gsrc = l52 < 48 ? Q.K + (size_t)b * N * 48 + l52 : Q.kf + (size_t)b * N * NU + (l52 - 48);
    const int gstride = l52 < 48 ? 48 : NU;
    St px = X[l12], pg = gsrc[0], pu = U[l4];
    for (int k = 0; k < N; k++) {
      if (lane < NX) L.dx[lane] = L.xn[lane] - (double)px;
      if (lane < 48) L.Kk[lane] = (double)pg;
      else if (lane < 52) {
        L.kk[lane - 48] = (double)pg;
        L.uk[lane - 48] = (double)pu;
      }
      {
        const size_t kn = (size_t)(k + 1 < N ? k + 1 : k);
        px = X[kn * NX + l12];
        pg = gsrc[kn * gstride];
        pu = U[kn * NU + l4];
      }
      trig_lanes(L.xn, L.trig, lane);
      QSYNC();
      if (lane < NU) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < NX; j++) acc += L.Kk[lane * NX + j] * L.dx[j];
        // rounded to the storage type before use: the recorded cost belongs to the iterate that is stored
        L.un[lane] = (double)(St)(L.uk[lane] + alpha * L.kk[lane] + acc);
      }
      QSYNC();
      cost += roll_knot(c, L, xg, lane);
      if (lane < 7) Q.Tg[nxt][((size_t)b * N + k) * 8 + lane] = L.trig[lane];
      if (lane < NX) Xt[(size_t)k * NX + lane] = (St)L.xn[lane];
      if (lane < NU) Ut[(size_t)k * NU + lane] = (St)L.un[lane];
      QSYNC();
      if (lane < NX) L.xn[lane] = (double)(St)L.xnext[lane];
      QSYNC();
    }
    double part = 0.0;
    if (lane < NX) {
      Xt[(size_t)N * NX + lane] = (St)L.xn[lane];
      part = c.qf[lane] * (L.xn[lane] - xg[lane]) * (L.xn[lane] - xg[lane]);
    }
    cost += 0.5 * wsum(part);
    if (cost < cost_old) {  // strict decrease; NaN is rejected
      if (lane == 0) {
        L.st.cost = cost; L.st.step = step; L.st.fp_failed = 0; L.st.cur = nxt;
      }
      QSYNC();
      return;
    }
    QSYNC();
  }
  if (lane == 0) L.st.fp_failed = 1;
  QSYNC();
}

template <typename St>
__global__ __launch_bounds__(64) void k_quad_begin(QBatch<St> Q) {
  __shared__ QLds L;
  q_begin(Q, L, blockIdx.x, threadIdx.x);
}

template <typename St>
__global__ __launch_bounds__(64, 4) void k_quad_iterate(QBatch<St> Q, int n_iters) {  // 4 waves per SIMD: 4096 trajectories resident at once
  __shared__ QLds L;
  __shared__ double xg[NX];
  const int b = blockIdx.x, lane = threadIdx.x;
  if (lane == 0) L.st = Q.st[b];
  if (lane < NX) xg[lane] = (double)Q.xg[(size_t)b * NX + lane];
  __syncthreads();
  for (int it = 0; it < n_iters; it++) {
    if (L.st.done || L.st.iter >= Q.c.iter_max) {
      if (lane == 0) L.st.done = 1;
      break;
    }
    int tries = 0;
    while (!q_backward(Q, L, b, lane, xg)) {
      if (++tries > 30) break;
    }
    const double prev = L.st.cost;
    q_forward(Q, L, b, lane, xg);
    if (lane == 0) {
      L.st.fwd_passes++;
      L.st.iter++;
      if (!Q.c.fixed_iters && !L.st.fp_failed && prev - L.st.cost <= Q.c.tol * prev) L.st.done = 1;
      if (L.st.iter >= Q.c.iter_max) L.st.done = 1;
    }
    __syncthreads();  // a real barrier: the next backward sweep reads U on other lanes than the forward pass wrote it
  }
  __syncthreads();
  if (lane == 0) Q.st[b] = L.st;
}

template <typename St>
__global__ void k_quad_emit(QBatch<St> Q, St* cost, int32_t* iters, St* x, St* u) {
  const int b = blockIdx.x, N = Q.N;
  const QState s = Q.st[b];
  if (threadIdx.x == 0) {
    if (cost) cost[b] = (St)s.cost;
    if (iters) iters[b] = s.fwd_passes;
  }
  if (x)
    for (int e = threadIdx.x; e < (N + 1) * NX; e += blockDim.x) x[(size_t)b * (N + 1) * NX + e] = Q.X[s.cur][(size_t)b * (N + 1) * NX + e];
  if (u)
    for (int e = threadIdx.x; e < N * NU; e += blockDim.x) u[(size_t)b * N * NU + e] = Q.U[s.cur][(size_t)b * N * NU + e];
}

thread_local std::string g_qerr;
direct_status_t qfail(direct_status_t st, const std::string& msg) {
  g_qerr = msg;
  return st;
}
#define QHIP_TRY(expr)                                                                             \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return qfail(DIRECT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));          \
  } while (0)

}  // namespace

struct direct_quad_handle_s {
  int dtype = 0, device = 0, max_batch = 0, N = 0, B = 0;
  size_t rsz = 4;
  bool begun = false, timed = false;
  void *x0 = nullptr, *xg = nullptr, *X[2] = {nullptr, nullptr}, *U[2] = {nullptr, nullptr}, *K = nullptr, *kf = nullptr, *Tg[2] = {nullptr, nullptr};
  void *o_cost = nullptr, *o_x = nullptr, *o_u = nullptr;
  int32_t* o_iters = nullptr;
  QState* st = nullptr;
  QConst c = {};
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<void*> allocs;
};

namespace {
template <typename St>
QBatch<St> make_q(direct_quad_handle_t h) {
  QBatch<St> Q;
  Q.B = h->B; Q.N = h->N; Q.x0 = (const St*)h->x0; Q.xg = (const St*)h->xg;
  for (int i = 0; i < 2; i++) { Q.X[i] = (St*)h->X[i]; Q.U[i] = (St*)h->U[i]; Q.Tg[i] = (double*)h->Tg[i]; }
  Q.K = (St*)h->K; Q.kf = (St*)h->kf; Q.st = h->st; Q.c = h->c;
  return Q;
}
direct_status_t set_params(direct_quad_handle_t h, const direct_quad_params_t* p) {
  if (!p) return qfail(DIRECT_ERR_INVALID, "null params");
  if (!(p->mass > 0) || !(p->dt > 0) || !(p->inertia[0] > 0) || !(p->inertia[1] > 0) || !(p->inertia[2] > 0) || p->iter_max < 0 ||
      !(p->reg_base > 1.0))
    return qfail(DIRECT_ERR_INVALID, "bad model parameters");
  QConst& c = h->c;
  c.m = p->mass; c.g = p->gravity; c.dt = p->dt; c.reg_base = p->reg_base; c.tol = p->tol;
  c.iter_max = p->iter_max; c.fixed_iters = p->fixed_iters;
  for (int i = 0; i < 3; i++) {
    c.J[i] = p->inertia[i];
    c.inv_J[i] = 1.0 / p->inertia[i];
    c.q[i] = p->q_pos; c.q[3 + i] = p->q_vel; c.q[6 + i] = p->q_ang; c.q[9 + i] = p->q_rate;
    c.qf[i] = p->qf_pos; c.qf[3 + i] = p->qf_vel; c.qf[6 + i] = p->qf_ang; c.qf[9 + i] = p->qf_rate;
  }
  c.r[0] = p->r_thrust; c.r[1] = c.r[2] = c.r[3] = p->r_torque;
  c.inv_m = 1.0 / c.m;
  c.kJ[0] = (c.J[2] - c.J[1]) / c.J[0]; c.kJ[1] = (c.J[0] - c.J[2]) / c.J[1]; c.kJ[2] = (c.J[1] - c.J[0]) / c.J[2];
  return DIRECT_OK;
}
}  // namespace

extern "C" {

const char* direct_quad_last_error(void) { return g_qerr.c_str(); }

void direct_quad_default_params(direct_quad_params_t* p) {
  if (!p) return;
  p->mass = 0.98; p->gravity = 9.81;  // Quadrotor.cpp:16-17
  p->inertia[0] = 2.64e-3; p->inertia[1] = 2.64e-3; p->inertia[2] = 4.96e-3;  // Quadrotor.cpp:18
  p->dt = 0.05;
  p->q_pos = 1.0; p->q_vel = 0.1; p->q_ang = 1.0; p->q_rate = 0.05;
  p->r_thrust = 0.05; p->r_torque = 50.0;
  p->qf_pos = 1000.0; p->qf_vel = 500.0; p->qf_ang = 500.0; p->qf_rate = 100.0;
  p->reg_base = 4.0; p->tol = 1.0e-6; p->iter_max = 50; p->fixed_iters = 0;
}

direct_status_t direct_quad_create(int32_t dtype, int32_t device, int32_t max_batch, int32_t n_knots, direct_quad_handle_t* out) {
  if (!out || max_batch <= 0 || n_knots <= 0 || (dtype != DIRECT_F32 && dtype != DIRECT_F64)) return qfail(DIRECT_ERR_INVALID, "bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return qfail(DIRECT_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return qfail(DIRECT_ERR_INVALID, "bad device ordinal");
  QHIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  QHIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return qfail(DIRECT_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
  direct_quad_handle_t h = new direct_quad_handle_s();
  h->dtype = dtype; h->device = device; h->max_batch = max_batch; h->N = n_knots; h->rsz = dtype == DIRECT_F64 ? 8 : 4;
  const size_t B = max_batch, N = n_knots, r = h->rsz;
  direct_status_t st = DIRECT_OK;
  auto A = [&](void** pp, size_t bytes) {
    if (st != DIRECT_OK) return;
    hipError_t e = hipMalloc(pp, bytes ? bytes : 16);
    if (e != hipSuccess) { st = qfail(DIRECT_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e)); return; }
    h->allocs.push_back(*pp);
  };
  A(&h->x0, B * NX * r); A(&h->xg, B * NX * r);
  for (int i = 0; i < 2; i++) { A(&h->X[i], B * (N + 1) * NX * r); A(&h->U[i], B * N * NU * r); A(&h->Tg[i], B * N * 8 * sizeof(double)); }
  A(&h->K, B * N * 48 * r); A(&h->kf, B * N * NU * r); A((void**)&h->st, B * sizeof(QState));
  A(&h->o_cost, B * r); A((void**)&h->o_iters, B * 4); A(&h->o_x, B * (N + 1) * NX * r); A(&h->o_u, B * N * NU * r);
  if (st == DIRECT_OK && (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess))
    st = qfail(DIRECT_ERR_DEVICE, "hipEventCreate failed");
  if (st != DIRECT_OK) {
    direct_quad_destroy(h);
    return st;
  }
  *out = h;
  return DIRECT_OK;
}

direct_status_t direct_quad_destroy(direct_quad_handle_t h) {
  if (!h) return DIRECT_OK;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  for (void* p : h->allocs) (void)hipFree(p);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  delete h;
  return DIRECT_OK;
}

direct_status_t direct_quad_begin(direct_quad_handle_t h, const direct_quad_params_t* p, int32_t batch, int32_t mem, const void* x0,
                                  const void* xg) {
  if (!h || !x0 || !xg) return qfail(DIRECT_ERR_INVALID, "null argument");
  if (batch <= 0 || batch > h->max_batch) return qfail(DIRECT_ERR_INVALID, "batch exceeds the handle's max_batch");
  direct_status_t s = set_params(h, p);
  if (s != DIRECT_OK) return s;
  QHIP_TRY(hipSetDevice(h->device));
  const hipMemcpyKind kind = mem == DIRECT_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  QHIP_TRY(hipMemcpyAsync(h->x0, x0, (size_t)batch * NX * h->rsz, kind, h->stream));
  QHIP_TRY(hipMemcpyAsync(h->xg, xg, (size_t)batch * NX * h->rsz, kind, h->stream));
  h->B = batch;
  if (h->dtype == DIRECT_F64) hipLaunchKernelGGL(k_quad_begin<double>, dim3(batch), dim3(64), 0, h->stream, make_q<double>(h));
  else hipLaunchKernelGGL(k_quad_begin<float>, dim3(batch), dim3(64), 0, h->stream, make_q<float>(h));
  QHIP_TRY(hipGetLastError());
  h->begun = true;
  return DIRECT_OK;
}

direct_status_t direct_quad_iterate(direct_quad_handle_t h, int32_t n_iters) {
  if (!h) return qfail(DIRECT_ERR_INVALID, "null handle");
  if (!h->begun) return qfail(DIRECT_ERR_INVALID, "direct_quad_begin has not been called");
  QHIP_TRY(hipSetDevice(h->device));
  QHIP_TRY(hipEventRecord(h->ev0, h->stream));
  if (h->dtype == DIRECT_F64) hipLaunchKernelGGL(k_quad_iterate<double>, dim3(h->B), dim3(64), 0, h->stream, make_q<double>(h), n_iters);
  else hipLaunchKernelGGL(k_quad_iterate<float>, dim3(h->B), dim3(64), 0, h->stream, make_q<float>(h), n_iters);
  QHIP_TRY(hipGetLastError());
  QHIP_TRY(hipEventRecord(h->ev1, h->stream));
  h->timed = true;
  return DIRECT_OK;
}

direct_status_t direct_quad_solve_batch(direct_quad_handle_t h, const direct_quad_params_t* p, int32_t batch, int32_t mem, const void* x0,
                                        const void* xg, void* cost, int32_t* iters, void* x, void* u) {
  direct_status_t s = direct_quad_begin(h, p, batch, mem, x0, xg);
  if (s != DIRECT_OK) return s;
  s = direct_quad_iterate(h, p->iter_max);
  if (s != DIRECT_OK) return s;
  const bool host = mem == DIRECT_MEM_HOST;
  const size_t B = batch, N = h->N, r = h->rsz;
  void* dc = host ? (cost ? h->o_cost : nullptr) : cost;
  int32_t* di = host ? (iters ? h->o_iters : nullptr) : iters;
  void* dx = host ? (x ? h->o_x : nullptr) : x;
  void* du = host ? (u ? h->o_u : nullptr) : u;
  if (h->dtype == DIRECT_F64)
    hipLaunchKernelGGL(k_quad_emit<double>, dim3(batch), dim3(256), 0, h->stream, make_q<double>(h), (double*)dc, di, (double*)dx, (double*)du);
  else
    hipLaunchKernelGGL(k_quad_emit<float>, dim3(batch), dim3(256), 0, h->stream, make_q<float>(h), (float*)dc, di, (float*)dx, (float*)du);
  QHIP_TRY(hipGetLastError());
  if (host) {
    if (cost) QHIP_TRY(hipMemcpyAsync(cost, dc, B * r, hipMemcpyDeviceToHost, h->stream));
    if (iters) QHIP_TRY(hipMemcpyAsync(iters, di, B * 4, hipMemcpyDeviceToHost, h->stream));
    if (x) QHIP_TRY(hipMemcpyAsync(x, dx, B * (N + 1) * NX * r, hipMemcpyDeviceToHost, h->stream));
    if (u) QHIP_TRY(hipMemcpyAsync(u, du, B * N * NU * r, hipMemcpyDeviceToHost, h->stream));
    QHIP_TRY(hipStreamSynchronize(h->stream));
  }
  return DIRECT_OK;
}

direct_status_t direct_quad_get(direct_quad_handle_t h, void* x, void* u, void* K, void* kf, double* scalars) {
  if (!h) return qfail(DIRECT_ERR_INVALID, "null handle");
  if (!h->begun) return qfail(DIRECT_ERR_INVALID, "direct_quad_begin has not been called");
  QHIP_TRY(hipSetDevice(h->device));
  const size_t B = h->B, N = h->N, r = h->rsz;
  std::vector<QState> st(B);
  QHIP_TRY(hipMemcpyAsync(st.data(), h->st, B * sizeof(QState), hipMemcpyDeviceToHost, h->stream));
  QHIP_TRY(hipStreamSynchronize(h->stream));
  for (size_t b = 0; b < B; b++) {
    const int cur = st[b].cur;
    if (x) QHIP_TRY(hipMemcpyAsync((char*)x + b * (N + 1) * NX * r, (char*)h->X[cur] + b * (N + 1) * NX * r, (N + 1) * NX * r, hipMemcpyDeviceToHost, h->stream));
    if (u) QHIP_TRY(hipMemcpyAsync((char*)u + b * N * NU * r, (char*)h->U[cur] + b * N * NU * r, N * NU * r, hipMemcpyDeviceToHost, h->stream));
    if (scalars) {
      double* o = scalars + b * 8;
      o[0] = st[b].cost; o[1] = st[b].reg; o[2] = st[b].step; o[3] = st[b].fp_failed; o[4] = st[b].bp_failed;
      o[5] = st[b].iter; o[6] = st[b].done; o[7] = st[b].fwd_passes;
    }
  }
  if (K) QHIP_TRY(hipMemcpyAsync(K, h->K, B * N * 48 * r, hipMemcpyDeviceToHost, h->stream));
  if (kf) QHIP_TRY(hipMemcpyAsync(kf, h->kf, B * N * NU * r, hipMemcpyDeviceToHost, h->stream));
  QHIP_TRY(hipStreamSynchronize(h->stream));
  return DIRECT_OK;
}

direct_status_t direct_quad_set_stream(direct_quad_handle_t h, void* hip_stream) {
  if (!h) return DIRECT_ERR_INVALID;
  h->stream = (hipStream_t)hip_stream;  // every copy, launch and event of the handle is enqueued on it from now on
  return DIRECT_OK;
}

direct_status_t direct_quad_last_kernel_ms(direct_quad_handle_t h, double* ms) {
  if (!h || !ms) return qfail(DIRECT_ERR_INVALID, "null argument");
  if (!h->timed) return qfail(DIRECT_ERR_INVALID, "nothing has been timed yet");
  QHIP_TRY(hipEventSynchronize(h->ev1));
  float t = 0.f;
  QHIP_TRY(hipEventElapsedTime(&t, h->ev0, h->ev1));
  *ms = (double)t;
  return DIRECT_OK;
}

}  // extern "C"
