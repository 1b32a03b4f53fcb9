/*
 * quad_ref.c -- TEST INFRASTRUCTURE: dense CPU iLQR for the BASELINE-label model (12-state / 4-control quadrotor,
 * unconstrained, N knots, explicit-Euler step dt), the checker of direct_amd/csrc/quad_wave.h.
 *
 * NO REFERENCE COUNTERPART.  BASELINE.json's metric string names a "12-state/4-ctrl quad" that does not exist in
 * ntu-caokun/DIRECT (the reference's optimiser has 9 states / 10 controls; the only 4-input
 * quadrotor is the simulator plant).  SURVEY.md 8(d) config 2 asks for this model as a second policy, reported
 * separately.  The MODEL is specified here and in include/direct_quad.h; its physical constants are the simulator
 * plant's (simulation/so3_quadrotor_simulator/src/dynamics/Quadrotor.cpp:16-20: g 9.81, mass 0.98, J = diag(2.64e-3,
 * 2.64e-3, 4.96e-3)).  The SOLVER keeps the conventions of the reference's outer loop where they carry over to an
 * unconstrained problem: Gauss-Newton Q-function (no second-order dynamics terms, ddp_optimizer.cpp:513-520),
 * regulariser lam = base^reg - 1 with the schedule of ddp_optimizer.cpp:452-474 (reg in 0..24), LLT failure ->
 * retry with a larger regulariser (:297-310), 11 step sizes 2^0..2^-10 (:666-670), gains from the regularised Quu
 * and the value update from the unregularised one (:529, 626-628).  A trial is accepted on a strict cost decrease
 * (no barrier, no filter: there are no constraints).  PARITY: against this file only (unpinned by construction).
 *
 *   x = [p(3), v(3), euler(phi, theta, psi), omega(3)],  u = [thrust T, torque(3)]
 *   pdot = v;  vdot = (T/m) R(euler) e3 - g e3;  eulerdot = W(euler) omega;  omegadot = J^-1 (tau - omega x J omega)
 *   x+ = x + dt f(x, u)
 *   cost = sum_k dt/2 [ (x_k - xg)' Q (x_k - xg) + (u_k - uh)' R (u_k - uh) ] + 1/2 (x_N - xg)' Qf (x_N - xg),
 *   uh = (m g, 0, 0, 0), Q / R / Qf diagonal.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/direct_quad.h"

#define NX 12
#define NU 4

typedef struct {
  double m, g, J[3], dt, q[NX], r[NU], qf[NX], reg_base;
} qparams;

static void qp_from(const direct_quad_params_t* p, qparams* o) {
  o->m = p->mass; o->g = p->gravity; o->dt = p->dt; o->reg_base = p->reg_base;
  for (int i = 0; i < 3; i++) o->J[i] = p->inertia[i];
  for (int i = 0; i < 3; i++) {
    o->q[i] = p->q_pos; o->q[3 + i] = p->q_vel; o->q[6 + i] = p->q_ang; o->q[9 + i] = p->q_rate;
    o->qf[i] = p->qf_pos; o->qf[3 + i] = p->qf_vel; o->qf[6 + i] = p->qf_ang; o->qf[9 + i] = p->qf_rate;
  }
  o->r[0] = p->r_thrust; o->r[1] = o->r[2] = o->r[3] = p->r_torque;
}

/* continuous dynamics f(x, u) */
static void qdyn(const qparams* P, const double* x, const double* u, double* f) {
  const double sph = sin(x[6]), cph = cos(x[6]), sth = sin(x[7]), cth = cos(x[7]), sps = sin(x[8]), cps = cos(x[8]);
  const double a = u[0] / P->m;
  f[0] = x[3]; f[1] = x[4]; f[2] = x[5];
  f[3] = a * (cph * sth * cps + sph * sps);
  f[4] = a * (cph * sth * sps - sph * cps);
  f[5] = a * (cph * cth) - P->g;
  const double tth = sth / cth;
  f[6] = x[9] + sph * tth * x[10] + cph * tth * x[11];
  f[7] = cph * x[10] - sph * x[11];
  f[8] = (sph * x[10] + cph * x[11]) / cth;
  f[9] = (u[1] - (P->J[2] - P->J[1]) * x[10] * x[11]) / P->J[0];
  f[10] = (u[2] - (P->J[0] - P->J[2]) * x[9] * x[11]) / P->J[1];
  f[11] = (u[3] - (P->J[1] - P->J[0]) * x[9] * x[10]) / P->J[2];
}

/* A = I + dt df/dx (row-major 12x12), B = dt df/du (12x4) */
void quad_ref_jacobians(const direct_quad_params_t* pp, const double* x, const double* u, double* A, double* B) {
  qparams P;
  qp_from(pp, &P);
  const double dt = P.dt;
  const double sph = sin(x[6]), cph = cos(x[6]), sth = sin(x[7]), cth = cos(x[7]), sps = sin(x[8]), cps = cos(x[8]);
  const double a = u[0] / P.m, tth = sth / cth, sec2 = 1.0 / (cth * cth);
  memset(A, 0, sizeof(double) * NX * NX);
  memset(B, 0, sizeof(double) * NX * NU);
  for (int i = 0; i < NX; i++) A[i * NX + i] = 1.0;
  for (int i = 0; i < 3; i++) A[i * NX + 3 + i] = dt;
  /* d vdot / d euler */
  A[3 * NX + 6] = dt * a * (-sph * sth * cps + cph * sps);
  A[3 * NX + 7] = dt * a * (cph * cth * cps);
  A[3 * NX + 8] = dt * a * (-cph * sth * sps + sph * cps);
  A[4 * NX + 6] = dt * a * (-sph * sth * sps - cph * cps);
  A[4 * NX + 7] = dt * a * (cph * cth * sps);
  A[4 * NX + 8] = dt * a * (cph * sth * cps + sph * sps);
  A[5 * NX + 6] = dt * a * (-sph * cth);
  A[5 * NX + 7] = dt * a * (-cph * sth);
  /* d eulerdot / d euler */
  A[6 * NX + 6] += dt * (cph * tth * x[10] - sph * tth * x[11]);
  A[6 * NX + 7] = dt * (sph * x[10] + cph * x[11]) * sec2;
  A[7 * NX + 6] = dt * (-sph * x[10] - cph * x[11]);
  A[8 * NX + 6] = dt * (cph * x[10] - sph * x[11]) / cth;
  A[8 * NX + 7] = dt * (sph * x[10] + cph * x[11]) * sth * sec2;
  /* d eulerdot / d omega = W */
  A[6 * NX + 9] = dt; A[6 * NX + 10] = dt * sph * tth; A[6 * NX + 11] = dt * cph * tth;
  A[7 * NX + 10] = dt * cph; A[7 * NX + 11] = -dt * sph;
  A[8 * NX + 10] = dt * sph / cth; A[8 * NX + 11] = dt * cph / cth;
  /* d omegadot / d omega */
  const double k0 = (P.J[2] - P.J[1]) / P.J[0], k1 = (P.J[0] - P.J[2]) / P.J[1], k2 = (P.J[1] - P.J[0]) / P.J[2];
  A[9 * NX + 10] = -dt * k0 * x[11]; A[9 * NX + 11] = -dt * k0 * x[10];
  A[10 * NX + 9] = -dt * k1 * x[11]; A[10 * NX + 11] = -dt * k1 * x[9];
  A[11 * NX + 9] = -dt * k2 * x[10]; A[11 * NX + 10] = -dt * k2 * x[9];
  /* B */
  B[3 * NU + 0] = dt * (cph * sth * cps + sph * sps) / P.m;
  B[4 * NU + 0] = dt * (cph * sth * sps - sph * cps) / P.m;
  B[5 * NU + 0] = dt * (cph * cth) / P.m;
  B[9 * NU + 1] = dt / P.J[0]; B[10 * NU + 2] = dt / P.J[1]; B[11 * NU + 3] = dt / P.J[2];
}

void quad_ref_step(const direct_quad_params_t* pp, const double* x, const double* u, double* xn) {
  qparams P;
  qp_from(pp, &P);
  double f[NX];
  qdyn(&P, x, u, f);
  for (int i = 0; i < NX; i++) xn[i] = x[i] + P.dt * f[i];
}

typedef struct {
  int N;
  qparams P;
  direct_quad_params_t pp;
  double xg[NX], uh[NU];
  double *x, *u, *xt, *ut, *K, *kf; /* [N+1][12], [N][4], trial buffers, [N][4][12], [N][4] */
  double cost;
  int reg, step, fp_failed, bp_failed, iter, done, fwd_passes;
} qsolver;

static double knot_cost(const qsolver* s, const double* x, const double* u) {
  double c = 0;
  for (int i = 0; i < NX; i++) c += s->P.q[i] * (x[i] - s->xg[i]) * (x[i] - s->xg[i]);
  for (int i = 0; i < NU; i++) c += s->P.r[i] * (u[i] - s->uh[i]) * (u[i] - s->uh[i]);
  return 0.5 * s->P.dt * c;
}
static double term_cost(const qsolver* s, const double* x) {
  double c = 0;
  for (int i = 0; i < NX; i++) c += s->P.qf[i] * (x[i] - s->xg[i]) * (x[i] - s->xg[i]);
  return 0.5 * c;
}

void* quad_ref_begin(const direct_quad_params_t* pp, int N, const double* x0, const double* xg) {
  qsolver* s = (qsolver*)calloc(1, sizeof(qsolver));
  s->N = N; s->pp = *pp;
  qp_from(pp, &s->P);
  memcpy(s->xg, xg, sizeof(double) * NX);
  s->uh[0] = s->P.m * s->P.g;
  s->x = (double*)calloc((size_t)(N + 1) * NX, 8); s->xt = (double*)calloc((size_t)(N + 1) * NX, 8);
  s->u = (double*)calloc((size_t)N * NU, 8); s->ut = (double*)calloc((size_t)N * NU, 8);
  s->K = (double*)calloc((size_t)N * NU * NX, 8); s->kf = (double*)calloc((size_t)N * NU, 8);
  memcpy(s->x, x0, sizeof(double) * NX);
  double c = 0;
  for (int k = 0; k < N; k++) { /* initial roll from the hover input */
    memcpy(s->u + (size_t)k * NU, s->uh, sizeof(double) * NU);
    c += knot_cost(s, s->x + (size_t)k * NX, s->u + (size_t)k * NU);
    quad_ref_step(pp, s->x + (size_t)k * NX, s->u + (size_t)k * NU, s->x + (size_t)(k + 1) * NX);
  }
  s->cost = c + term_cost(s, s->x + (size_t)N * NX);
  return s;
}
void quad_ref_end(void* h) {
  qsolver* s = (qsolver*)h;
  free(s->x); free(s->xt); free(s->u); free(s->ut); free(s->K); free(s->kf); free(s);
}

/* one backward sweep; returns 1 on success, 0 when the 4x4 LLT failed */
static int backward(qsolver* s) {
  {
    int reg = s->reg;
    if (s->fp_failed || s->bp_failed) reg += 1;
    else if (s->step == 0) reg -= 1;
    else if (s->step > 3) reg += 1;
    s->reg = reg < 0 ? 0 : (reg > 24 ? 24 : reg);
  }
  double lam = 1.0;
  for (int q = 0; q < s->reg; q++) lam *= s->P.reg_base;
  lam -= 1.0;
  const int N = s->N;
  double V[NX * NX] = {0}, Vx[NX];
  for (int i = 0; i < NX; i++) {
    V[i * NX + i] = s->P.qf[i];
    Vx[i] = s->P.qf[i] * (s->x[(size_t)N * NX + i] - s->xg[i]);
  }
  for (int k = N - 1; k >= 0; k--) {
    const double *x = s->x + (size_t)k * NX, *u = s->u + (size_t)k * NU;
    double A[NX * NX], B[NX * NU], VA[NX * NX], VB[NX * NU], Qxx[NX * NX], Qux[NU * NX], Quu[NU * NU], Qx[NX], Qu[NU];
    quad_ref_jacobians(&s->pp, x, u, A, B);
    for (int i = 0; i < NX; i++)
      for (int j = 0; j < NX; j++) {
        double acc = 0;
        for (int l = 0; l < NX; l++) acc += V[i * NX + l] * A[l * NX + j];
        VA[i * NX + j] = acc;
      }
    for (int i = 0; i < NX; i++)
      for (int j = 0; j < NU; j++) {
        double acc = 0;
        for (int l = 0; l < NX; l++) acc += V[i * NX + l] * B[l * NU + j];
        VB[i * NU + j] = acc;
      }
    for (int i = 0; i < NX; i++) {
      double acc = 0;
      for (int l = 0; l < NX; l++) acc += A[l * NX + i] * Vx[l];
      Qx[i] = s->P.dt * s->P.q[i] * (x[i] - s->xg[i]) + acc;
      for (int j = 0; j < NX; j++) {
        double a2 = 0;
        for (int l = 0; l < NX; l++) a2 += A[l * NX + i] * VA[l * NX + j];
        Qxx[i * NX + j] = a2 + (i == j ? s->P.dt * s->P.q[i] : 0.0);
      }
    }
    for (int i = 0; i < NU; i++) {
      double acc = 0;
      for (int l = 0; l < NX; l++) acc += B[l * NU + i] * Vx[l];
      Qu[i] = s->P.dt * s->P.r[i] * (u[i] - s->uh[i]) + acc;
      for (int j = 0; j < NX; j++) {
        double a2 = 0;
        for (int l = 0; l < NX; l++) a2 += B[l * NU + i] * VA[l * NX + j];
        Qux[i * NX + j] = a2;
      }
      for (int j = 0; j < NU; j++) {
        double a2 = 0;
        for (int l = 0; l < NX; l++) a2 += B[l * NU + i] * VB[l * NU + j];
        Quu[i * NU + j] = a2 + (i == j ? s->P.dt * s->P.r[i] : 0.0);
      }
    }
    /* LLT of Quu + lam I (lower triangle), failure iff a pivot <= 0 */
    double Lm[NU * NU] = {0};
    for (int j = 0; j < NU; j++) {
      double d = Quu[j * NU + j] + lam;
      for (int l = 0; l < j; l++) d -= Lm[j * NU + l] * Lm[j * NU + l];
      if (d <= 0.0) { s->bp_failed = 1; return 0; }
      const double dj = sqrt(d);
      Lm[j * NU + j] = dj;
      for (int i = j + 1; i < NU; i++) {
        double v = Quu[i * NU + j];
        for (int l = 0; l < j; l++) v -= Lm[i * NU + l] * Lm[j * NU + l];
        Lm[i * NU + j] = v / dj;
      }
    }
    double* Kk = s->K + (size_t)k * NU * NX;
    double* kk = s->kf + (size_t)k * NU;
    for (int c = 0; c <= NX; c++) { /* column 0: Qu, columns 1..12: Qux[:, c-1] */
      double rhs[NU], y[NU], z[NU];
      for (int i = 0; i < NU; i++) rhs[i] = c == 0 ? Qu[i] : Qux[i * NX + c - 1];
      for (int i = 0; i < NU; i++) {
        double v = rhs[i];
        for (int l = 0; l < i; l++) v -= Lm[i * NU + l] * y[l];
        y[i] = v / Lm[i * NU + i];
      }
      for (int i = NU - 1; i >= 0; i--) {
        double v = y[i];
        for (int l = i + 1; l < NU; l++) v -= Lm[l * NU + i] * z[l];
        z[i] = v / Lm[i * NU + i];
      }
      for (int i = 0; i < NU; i++) {
        if (c == 0) kk[i] = -z[i];
        else Kk[i * NX + c - 1] = -z[i];
      }
    }
    /* value update with the UNREGULARISED Quu (reference convention, ddp_optimizer.cpp:626-628) */
    double QuuK[NU * NX], Quuk[NU];
    for (int i = 0; i < NU; i++) {
      double a1 = 0;
      for (int l = 0; l < NU; l++) a1 += Quu[i * NU + l] * kk[l];
      Quuk[i] = a1;
      for (int j = 0; j < NX; j++) {
        double a2 = 0;
        for (int l = 0; l < NU; l++) a2 += Quu[i * NU + l] * Kk[l * NX + j];
        QuuK[i * NX + j] = a2;
      }
    }
    double Vn[NX * NX];
    for (int i = 0; i < NX; i++) {
      double a1 = Qx[i];
      for (int l = 0; l < NU; l++) a1 += Kk[l * NX + i] * Quuk[l] + Kk[l * NX + i] * Qu[l] + Qux[l * NX + i] * kk[l];
      Vx[i] = a1;
      for (int j = 0; j < NX; j++) {
        double a2 = Qxx[i * NX + j];
        for (int l = 0; l < NU; l++)
          a2 += Kk[l * NX + i] * QuuK[l * NX + j] + Kk[l * NX + i] * Qux[l * NX + j] + Qux[l * NX + i] * Kk[l * NX + j];
        Vn[i * NX + j] = a2;
      }
    }
    for (int i = 0; i < NX; i++)
      for (int j = 0; j < NX; j++) V[i * NX + j] = 0.5 * (Vn[i * NX + j] + Vn[j * NX + i]);
  }
  s->bp_failed = 0;
  return 1;
}

static void forward(qsolver* s) {
  const int N = s->N;
  for (int step = 0; step < 11; step++) {
    double alpha = 1.0;
    for (int q = 0; q < step; q++) alpha *= 0.5;
    memcpy(s->xt, s->x, sizeof(double) * NX);
    double c = 0;
    for (int k = 0; k < N; k++) {
      const double* xo = s->x + (size_t)k * NX;
      double* xn = s->xt + (size_t)k * NX;
      double* un = s->ut + (size_t)k * NU;
      for (int i = 0; i < NU; i++) {
        double a = 0;
        for (int j = 0; j < NX; j++) a += s->K[((size_t)k * NU + i) * NX + j] * (xn[j] - xo[j]);
        un[i] = s->u[(size_t)k * NU + i] + alpha * s->kf[(size_t)k * NU + i] + a;
      }
      c += knot_cost(s, xn, un);
      quad_ref_step(&s->pp, xn, un, s->xt + (size_t)(k + 1) * NX);
    }
    c += term_cost(s, s->xt + (size_t)N * NX);
    if (c < s->cost) { /* NaN fails the comparison and is rejected */
      memcpy(s->x, s->xt, sizeof(double) * (size_t)(N + 1) * NX);
      memcpy(s->u, s->ut, sizeof(double) * (size_t)N * NU);
      s->cost = c; s->step = step; s->fp_failed = 0;
      return;
    }
  }
  s->fp_failed = 1;
}

/* n trips of the outer loop; exits like the device: iter_max, or |dJ| <= tol * J unless fixed_iters */
int quad_ref_iterate(void* h, int n) {
  qsolver* s = (qsolver*)h;
  for (int it = 0; it < n; it++) {
    if (s->done || s->iter >= s->pp.iter_max) { s->done = 1; break; }
    int tries = 0;
    while (!backward(s)) {
      if (++tries > 30) break;
    }
    const double prev = s->cost;
    forward(s);
    s->fwd_passes++;
    s->iter++;
    if (!s->pp.fixed_iters && !s->fp_failed && prev - s->cost <= s->pp.tol * prev) s->done = 1;
    if (s->iter >= s->pp.iter_max) s->done = 1;
  }
  return s->done;
}

void quad_ref_get(void* h, double* x, double* u, double* K, double* kf, double* scalars) {
  qsolver* s = (qsolver*)h;
  if (x) memcpy(x, s->x, sizeof(double) * (size_t)(s->N + 1) * NX);
  if (u) memcpy(u, s->u, sizeof(double) * (size_t)s->N * NU);
  if (K) memcpy(K, s->K, sizeof(double) * (size_t)s->N * NU * NX);
  if (kf) memcpy(kf, s->kf, sizeof(double) * (size_t)s->N * NU);
  if (scalars) {
    scalars[0] = s->cost; scalars[1] = s->reg; scalars[2] = s->step; scalars[3] = s->fp_failed;
    scalars[4] = s->bp_failed; scalars[5] = s->iter; scalars[6] = s->done; scalars[7] = s->fwd_passes;
  }
}

/* whole solves for a batch (OpenMP), for the CPU baseline of the label model */
int quad_ref_solve_batch(const direct_quad_params_t* pp, int batch, int N, const double* x0, const double* xg, double* cost,
                         int32_t* iters, double* x_out, double* u_out, int n_threads) {
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
  for (int b = 0; b < batch; b++) {
    void* h = quad_ref_begin(pp, N, x0 + (size_t)b * NX, xg + (size_t)b * NX);
    quad_ref_iterate(h, pp->iter_max);
    qsolver* s = (qsolver*)h;
    if (cost) cost[b] = s->cost;
    if (iters) iters[b] = s->fwd_passes;
    if (x_out) memcpy(x_out + (size_t)b * (N + 1) * NX, s->x, sizeof(double) * (size_t)(N + 1) * NX);
    if (u_out) memcpy(u_out + (size_t)b * N * NU, s->u, sizeof(double) * (size_t)N * NU);
    quad_ref_end(h);
  }
  (void)n_threads;
  return 0;
}
