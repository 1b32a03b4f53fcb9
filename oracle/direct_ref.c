/*
 * direct_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C99, fp64, dense, one problem at a time) of the IPDDP trajectory
 * optimiser of ntu-caokun/DIRECT:
 *     global_planner/src/ddp_optimizer.cpp                   ("DDP" below)
 *     global_planner/include/global_planner/ddp_optimizer.h  ("DDPH")
 *     global_planner/src/teach_repeat_planner.cpp            ("TRP", time allocation only)
 * Every function cites the reference lines it follows.  It deliberately keeps the dense
 * formulation of the reference (cx, cu materialised, dense products, LLT of the lower
 * triangle) so that it is an independent check of the structured HIP kernels.
 *
 * PARITY UNPINNED: the reference ships no test, golden vector or fixture for this path
 *  and cannot be built here (Eigen3, ROS and OOQP headers are absent
 * and may not be stubbed).  The arithmetic that lives in the un-vendored, un-pinned Eigen 3
 * (README implies 3.3.4 / 3.2.92) -- dense products, LLT<MatrixXd> (lower triangle,
 * NumericalIssue iff a pivot <= 0), .inverse(), lpNorm -- is restated from its published
 * semantics.  This file is cross-checked against an independent NumPy
 * restatement (oracle/ddp_numpy.py) and analytic identities (tests/test_oracle_*.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#include "../include/direct_ddp.h"

#include <malloc.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NX 9
#define NU 10
#define NCTRL 6

/* ---- constant tables -------------------------------------------------------------------- */
/* DDP:63-77 (MINVO value tables) */
static const double MINVO6[6][6] = {
    {1.0, -0.06471861202, -0.03728008486, -0.02577637794, -0.02027573243, -0.01678273037},
    {1.0, 0.03314986096, -0.06548114211, -0.05530463802, -0.04362718953, -0.03671639115},
    {1.0, 0.3375528997, 0.05836232552, -0.02920033165, -0.04690387913, -0.04376447947},
    {1.0, 0.6624471003, 0.3832565261, 0.1916286091, 0.06985980172, -0.002892843108},
    {1.0, 0.966850139, 0.868219136, 0.7594116288, 0.6521050661, 0.5510660979},
    {1.0, 1.064718612, 1.092157139, 1.108091959, 1.118023718, 1.123960059}};
static const double MINVO_V6[5][6] = {
    {0, 1.0, -0.1423379297, -0.1332742327, -0.1242105357, -0.126304257},
    {0, 1.0, 0.1887439858, -0.1831318297, -0.2466606848, -0.2321393311},
    {0, 1.0, 1.0, 0.5411016575, 0.08220331498, -0.2433474658},
    {0, 1.0, 1.811256014, 2.250636213, 2.381669451, 2.282405938},
    {0, 1.0, 2.14233793, 3.293739556, 4.445141183, 5.585385392}};
static const double MINVO_A6[4][6] = {{0, 0, 2.0, -0.4472869252, -0.6133793313, -0.6406553622},
                                      {0, 0, 2.0, 1.223711659, -0.5552714346, -1.854618819},
                                      {0, 0, 2.0, 4.776288341, 6.54988193, 6.841145057},
                                      {0, 0, 2.0, 6.447286925, 13.17576837, 22.04662796}};
/* DDP:79-95 (Bezier value tables) */
static const double BEZ6[6][6] = {{1.0, 0, 0, 0, 0, 0},       {1.0, 0.2, 0, 0, 0, 0},
                                  {1.0, 0.4, 0.1, 0, 0, 0},   {1.0, 0.6, 0.3, 0.1, 0, 0},
                                  {1.0, 0.8, 0.6, 0.4, 0.2, 0}, {1.0, 1.0, 1.0, 1.0, 1.0, 1.0}};
static const double BEZ_V6[5][6] = {{0, 1.0, 0, 0, 0, 0},
                                    {0, 1.0, 0.5, 0, 0, 0},
                                    {0, 1.0, 1.0, 0.5, 0, 0},
                                    {0, 1.0, 1.5, 1.5, 1.0, 0},
                                    {0, 1.0, 2.0, 3.0, 4.0, 5.0}};
static const double BEZ_A6[4][6] = {
    {0, 0, 2.0, 0, 0, 0}, {0, 0, 2.0, 2.0, 0, 0}, {0, 0, 2.0, 4.0, 4.0, 0}, {0, 0, 2.0, 6.0, 12.0, 20.0}};
/* DDP:1544-1560 (d/dT tables: coefficient of T^(col-1), T^(col-2), T^(col-3)). Always MINVO: quirk Q1. */
static const double MINVO6_DT[6][6] = {
    {0, -0.06471861202, -0.07456016972, -0.07732913382, -0.08110292972, -0.08391365186},
    {0, 0.03314986096, -0.1309622842, -0.1659139141, -0.1745087581, -0.1835819558},
    {0, 0.3375528997, 0.116724651, -0.08760099494, -0.1876155165, -0.2188223973},
    {0, 0.6624471003, 0.7665130522, 0.5748858272, 0.2794392069, -0.01446421554},
    {0, 0.966850139, 1.736438272, 2.278234886, 2.608420264, 2.755330489},
    {0, 1.064718612, 2.184314278, 3.324275878, 4.472094873, 5.619800295}};
static const double MINVO_V6_DT[5][6] = {{0, 0, -0.1423379297, -0.2665484655, -0.3726316072, -0.5052170278},
                                         {0, 0, 0.1887439858, -0.3662636595, -0.7399820545, -0.9285573245},
                                         {0, 0, 1.0, 1.082203315, 0.2466099449, -0.9733898632},
                                         {0, 0, 1.811256014, 4.501272426, 7.145008354, 9.129623752},
                                         {0, 0, 2.14233793, 6.587479113, 13.33542355, 22.34154157}};
static const double MINVO_A6_DT[4][6] = {{0, 0, 0, -0.4472869252, -1.226758663, -1.921966087},
                                         {0, 0, 0, 1.223711659, -1.110542869, -5.563856457},
                                         {0, 0, 0, 4.776288341, 13.09976386, 20.52343517},
                                         {0, 0, 0, 6.447286925, 26.35153674, 66.13988387}};
/* DDP:1050-1055 */
static const double TEMPM[6][6] = {{1, 0, 0, 0, 0, 0},      {-5, 5, 0, 0, 0, 0},
                                   {10, -20, 10, 0, 0, 0},  {-10, 30, -30, 10, 0, 0},
                                   {5, -20, 30, -20, 5, 0}, {-1, 5, -10, 10, -5, 1}};
static const double EK_INV[3] = {1.0, 1.0, 0.5}; /* DDP:101-103 */

/* ---- solver state (DDPH:18-28 algParam, 30-210 fwdPass, 212-236 bwdPass) ------------------ */
typedef struct {
  int N, pmax, ncmax;
  /* algParam */
  double mu, tol;
  int maxiter, infeas;
  /* fwdPass scalars */
  double maxVel, maxAcc, w_snap, Rtime, w_term, reg_exp_base;
  int time_power, minvo, zero_init, line_init, fixed_iters, exact_dt;
  double M6[6][6], Mv[5][6], Ma[4][6];
  double xd[NX];
  const int* np;
  int* nc;
  const double* planes; /* [N][pmax][4] */
  const double* seeds;  /* [N][3] or NULL */
  double *x, *u, *c, *s, *y, *q;
  double p, px[NX], pxx[NX * NX];
  double *fx, *fu, *qu, *quu, *cx, *cu;
  double cost, costq, logcost, err, stepsize;
  int step, fp_failed;
  double* filter; /* [2][filter_cap] stored as pairs */
  int nfilter, filter_cap;
  /* bwdPass */
  double *ku, *Ku, *ks, *ky, *Ks, *Ky;
  double reg, opterr, dV[2];
  int bp_failed;
  /* outer loop */
  int iter, rtn, fwd_passes, line_failed, infeas_ref;
  int bp_no_upd_count, no_upd_count;
  double prev_cost, prev_costq;
  /* scratch for forwardpass / backwardpass */
  double *xn, *un, *cn, *sn, *yn, *qn, *bw_scratch;
  /* optional per-iteration trace: cost, costq, logcost, err, mu, reg, step, opterr, stepsize, fp_failed, n_sweeps, bp_failed */
  double* trace;
  int trace_cap;
} ref_t;

#define TRACE_W 12

/* ---- small dense helpers (row-major) ---------------------------------------------------- */
static void mat_zero(double* a, int n) { memset(a, 0, sizeof(double) * (size_t)n); }

/* C(m x n) = A(m x k) * B(k x n) */
static void mm(double* C, const double* A, const double* B, int m, int k, int n) {
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) {
      double acc = 0.0;
      for (int l = 0; l < k; l++) acc += A[i * k + l] * B[l * n + j];
      C[i * n + j] = acc;
    }
}
/* C(k x n) = A(m x k)^T * B(m x n) */
static void mtm(double* C, const double* A, const double* B, int m, int k, int n) {
  for (int i = 0; i < k; i++)
    for (int j = 0; j < n; j++) {
      double acc = 0.0;
      for (int l = 0; l < m; l++) acc += A[l * k + i] * B[l * n + j];
      C[i * n + j] = acc;
    }
}

/* Eigen::LLT<MatrixXd> semantics: reads the lower triangle, unblocked,
 * NumericalIssue iff a pivot x = A_kk - sum L_kj^2 is <= 0 (NaN does not trip it). */
static int llt_lower(double* L, const double* A, int n) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) L[i * n + j] = (j <= i) ? A[i * n + j] : 0.0;
  for (int k = 0; k < n; k++) {
    double x = L[k * n + k];
    for (int j = 0; j < k; j++) x -= L[k * n + j] * L[k * n + j];
    if (x <= 0.0) return 0;
    x = sqrt(x);
    L[k * n + k] = x;
    for (int i = k + 1; i < n; i++) {
      double v = L[i * n + k];
      for (int j = 0; j < k; j++) v -= L[i * n + j] * L[k * n + j];
      L[i * n + k] = v / x;
    }
  }
  return 1;
}
/* solve L L^T X = B for nrhs columns; B is n x nrhs row-major, overwritten */
static void llt_solve(const double* L, double* B, int n, int nrhs) {
  for (int c = 0; c < nrhs; c++) {
    for (int i = 0; i < n; i++) {
      double v = B[i * nrhs + c];
      for (int j = 0; j < i; j++) v -= L[i * n + j] * B[j * nrhs + c];
      B[i * nrhs + c] = v / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
      double v = B[i * nrhs + c];
      for (int j = i + 1; j < n; j++) v -= L[j * n + i] * B[j * nrhs + c];
      B[i * nrhs + c] = v / L[i * n + i];
    }
  }
}

/* ---- time-dependent tables -------------------------------------------------------------- */
/* DDP:836-890 time2barFkbarGk (sys_order == 3) */
static void time2FG(double Tk, double F[3][3], double G[3][3]) {
  double Tk2 = Tk * Tk, Tk3 = Tk2 * Tk, Tk4 = Tk3 * Tk, Tk5 = Tk4 * Tk;
  double f[3][3] = {{1.0, Tk, Tk2 / 2.0}, {0.0, 1.0, Tk}, {0.0, 0.0, 1.0}};
  double g[3][3] = {{Tk3, Tk4, Tk5}, {3 * Tk2, 4 * Tk3, 5 * Tk4}, {6 * Tk, 12 * Tk2, 20 * Tk3}};
  memcpy(F, f, sizeof f);
  memcpy(G, g, sizeof g);
}
/* DDP:892-962 time2barFkprimebarGkprime (sys_order == 3) */
static void time2FGprime(double Tk, double Fp[3][3], double Gp[3][3]) {
  double Tk2 = Tk * Tk, Tk3 = Tk2 * Tk, Tk4 = Tk3 * Tk;
  double f[3][3] = {{0, 1, Tk}, {0, 0, 1}, {0, 0, 0}};
  double g[3][3] = {{3 * Tk2, 4 * Tk3, 5 * Tk4}, {6 * Tk, 12 * Tk2, 20 * Tk3}, {6, 24 * Tk, 60 * Tk2}};
  memcpy(Fp, f, sizeof f);
  memcpy(Gp, g, sizeof g);
}
/* DDP:964-1015 time2barR (sys_order == 3) */
static void time2R(double Tk, double R[3][3], double Rp[3][3], double Rpp[3][3]) {
  double Tk2 = Tk * Tk, Tk3 = Tk2 * Tk, Tk4 = Tk3 * Tk, Tk5 = Tk4 * Tk;
  double r[3][3] = {{36 * Tk, 72 * Tk2, 120 * Tk3}, {72 * Tk2, 192 * Tk3, 360 * Tk4}, {120 * Tk3, 360 * Tk4, 720 * Tk5}};
  double rp[3][3] = {{36, 144 * Tk, 360 * Tk2}, {144 * Tk, 576 * Tk2, 1440 * Tk3}, {360 * Tk2, 1440 * Tk3, 3600 * Tk4}};
  double rpp[3][3] = {{0, 144, 720 * Tk}, {144, 1152 * Tk, 4320 * Tk2}, {720 * Tk, 4320 * Tk2, 14400 * Tk3}};
  memcpy(R, r, sizeof r);
  memcpy(Rp, rp, sizeof rp);
  memcpy(Rpp, rpp, sizeof rpp);
}
/* kron(A3x3, I3) into a 9x9 row-major matrix: bar(i*3+k, j*3+k) = A(i,j)   (DDP:874-889) */
static void kron3(double* bar, double A[3][3]) {
  mat_zero(bar, 81);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) bar[(i * 3 + k) * 9 + (j * 3 + k)] = A[i][j];
}

/* DDP:1062-1067 computenextx */
static void computenextx(const double* x, const double* u, double* xnext) {
  double F[3][3], G[3][3], bF[81], bG[81];
  time2FG(u[9], F, G);
  kron3(bF, F);
  kron3(bG, G);
  for (int i = 0; i < 9; i++) {
    double a = 0.0, b = 0.0;
    for (int j = i; j < 9; j++) a += bF[i * 9 + j] * x[j]; /* triangularView<Upper> */
    for (int j = 0; j < 9; j++) b += bG[i * 9 + j] * u[j];
    xnext[i] = a + b;
  }
}

/* scaled tables polyt2minvotau (6x6), _v (5x6), _a (4x6)   (DDP:1148-1162, 1203-1210, 1240-1249) */
static void scaled_tables(const ref_t* r, double T, double P6[6][6], double V6[5][6], double A6[4][6]) {
  double Tkv[7];
  Tkv[0] = T;
  for (int i = 1; i < 7; i++) Tkv[i] = Tkv[i - 1] * Tkv[0];
  for (int j = 0; j < 6; j++) {
    P6[j][0] = 1.0;
    for (int i = 1; i < 6; i++) P6[j][i] = r->M6[j][i] * Tkv[i - 1];
  }
  for (int j = 0; j < 5; j++) {
    V6[j][0] = 0.0;
    V6[j][1] = r->Mv[j][1];
    for (int i = 2; i < 6; i++) V6[j][i] = r->Mv[j][i] * Tkv[i - 2];
  }
  for (int j = 0; j < 4; j++) {
    A6[j][0] = 0.0;
    A6[j][1] = 0.0;
    A6[j][2] = r->Ma[j][2];
    for (int i = 3; i < 6; i++) A6[j][i] = r->Ma[j][i] * Tkv[i - 3];
  }
}

/* DDP:1132-1285 computecminvo: c(x,u) for segment k.  Returns the constraint count. */
static int computecminvo(const ref_t* r, const double* x, const double* u, int k, double* c) {
  double polyCoeff[6][3];
  for (int j = 0; j < 3; j++) {
    for (int i = 0; i < 3; i++) polyCoeff[i][j] = x[i * 3 + j] * EK_INV[i];
    for (int i = 3; i < 6; i++) polyCoeff[i][j] = u[(i - 3) * 3 + j];
  }
  double P6[6][6], V6[5][6], A6[4][6];
  scaled_tables(r, u[9], P6, V6, A6);
  double posCoeff[6][3];
  for (int j = 0; j < 6; j++)
    for (int d = 0; d < 3; d++) {
      double a = 0.0;
      for (int i = 0; i < 6; i++) a += P6[j][i] * polyCoeff[i][d];
      posCoeff[j][d] = a;
    }
  int P = r->np[k];
  const double* pl = r->planes + (size_t)k * r->pmax * 4;
  int n = 0;
  for (int j = 0; j < 6; j++)
    for (int q = 0; q < P; q++)
      c[j * P + q] = pl[q * 4 + 0] * posCoeff[j][0] + pl[q * 4 + 1] * posCoeff[j][1] +
                     pl[q * 4 + 2] * posCoeff[j][2] + pl[q * 4 + 3];
  n = 6 * P;
  double cv[15], ca[12];
  for (int j = 0; j < 5; j++)
    for (int d = 0; d < 3; d++) {
      double a = 0.0;
      for (int i = 1; i < 6; i++) a += V6[j][i] * polyCoeff[i][d];
      cv[j * 3 + d] = a;
    }
  for (int i = 0; i < 15; i++) c[n + i] = cv[i] - r->maxVel;
  for (int i = 0; i < 15; i++) c[n + 15 + i] = -cv[i] - r->maxVel;
  n += 30;
  for (int j = 0; j < 4; j++)
    for (int d = 0; d < 3; d++) {
      double a = 0.0;
      for (int i = 2; i < 6; i++) a += A6[j][i] * polyCoeff[i][d];
      ca[j * 3 + d] = a;
    }
  for (int i = 0; i < 12; i++) c[n + i] = ca[i] - r->maxAcc;
  for (int i = 0; i < 12; i++) c[n + 12 + i] = -ca[i] - r->maxAcc;
  n += 24;
  c[n] = -u[9] + 0.3;
  n += 1;
  if (!r->minvo)
    for (int i = 0; i < n; i++) c[i] = c[i] - 2.0e-4;
  return n;
}

/* DDP:1289-1292 computep */
static double computep(const ref_t* r, const double* x) {
  double a = 0.0;
  for (int i = 0; i < 9; i++) a += (x[i] - r->xd[i]) * r->w_term * (x[i] - r->xd[i]);
  return 0.5 * a;
}
/* DDP:1294-1305 computeq */
static double computeq(const ref_t* r, const double* u) {
  double R[3][3], Rp[3][3], Rpp[3][3], bR[81];
  time2R(u[9], R, Rp, Rpp);
  kron3(bR, R);
  double a = 0.0;
  for (int i = 0; i < 9; i++) {
    double t = 0.0;
    for (int j = 0; j < 9; j++) t += bR[i * 9 + j] * u[j];
    a += u[i] * t;
  }
  if (r->time_power == 2) return 0.5 * r->w_snap * a + 0.5 * u[9] * r->Rtime * u[9];
  return 0.5 * r->w_snap * a + 0.5 * r->Rtime * u[9];
}

/* DDP:1318-1323 computeprelated */
static void computeprelated(ref_t* r) {
  const double* xN = r->x + (size_t)r->N * 9;
  r->p = computep(r, xN);
  for (int i = 0; i < 9; i++) r->px[i] = r->w_term * (xN[i] - r->xd[i]);
  mat_zero(r->pxx, 81);
  for (int i = 0; i < 9; i++) r->pxx[i * 9 + i] = r->w_term;
}
/* DDP:1325-1336 computefrelated */
static void computefrelated(ref_t* r) {
  for (int i = 0; i < r->N; i++) {
    const double* u = r->u + (size_t)i * 10;
    const double* x = r->x + (size_t)i * 9;
    double F[3][3], G[3][3], Fp[3][3], Gp[3][3], bG[81], bFp[81], bGp[81];
    time2FG(u[9], F, G);
    kron3(r->fx + (size_t)i * 81, F);
    kron3(bG, G);
    time2FGprime(u[9], Fp, Gp);
    kron3(bFp, Fp);
    kron3(bGp, Gp);
    double* fu = r->fu + (size_t)i * 90;
    for (int a = 0; a < 9; a++) {
      double g = 0.0;
      for (int b = a + 1; b < 9; b++) g += bFp[a * 9 + b] * x[b]; /* StrictlyUpper */
      for (int b = 0; b < 9; b++) g += bGp[a * 9 + b] * u[b];
      for (int b = 0; b < 9; b++) fu[a * 10 + b] = bG[a * 9 + b];
      fu[a * 10 + 9] = g;
    }
  }
}
/* DDP:1338-1368 computeqrelated */
static void computeqrelated(ref_t* r) {
  for (int i = 0; i < r->N; i++) {
    const double* u = r->u + (size_t)i * 10;
    double R[3][3], Rp[3][3], Rpp[3][3], bR[81], bRp[81], bRpp[81];
    time2R(u[9], R, Rp, Rpp);
    kron3(bR, R);
    kron3(bRp, Rp);
    kron3(bRpp, Rpp);
    double Ru[9], Rpu[9], Rppu[9];
    for (int a = 0; a < 9; a++) {
      double t0 = 0, t1 = 0, t2 = 0;
      for (int b = 0; b < 9; b++) {
        t0 += bR[a * 9 + b] * u[b];
        t1 += bRp[a * 9 + b] * u[b];
        t2 += bRpp[a * 9 + b] * u[b];
      }
      Ru[a] = t0;
      Rpu[a] = t1;
      Rppu[a] = t2;
    }
    double uRpu = 0, uRppu = 0;
    for (int a = 0; a < 9; a++) {
      uRpu += u[a] * Rpu[a];
      uRppu += u[a] * Rppu[a];
    }
    double* qu = r->qu + (size_t)i * 10;
    double* quu = r->quu + (size_t)i * 100;
    for (int a = 0; a < 9; a++) qu[a] = r->w_snap * Ru[a];
    for (int a = 0; a < 9; a++) {
      for (int b = 0; b < 9; b++) quu[a * 10 + b] = r->w_snap * bR[a * 9 + b];
      quu[a * 10 + 9] = r->w_snap * Rpu[a];
      quu[9 * 10 + a] = r->w_snap * Rpu[a]; /* u' Rp == (Rp u)' : Rp symmetric */
    }
    if (r->time_power == 2) {
      qu[9] = r->Rtime * u[9] + 0.5 * r->w_snap * uRpu;
      quu[99] = r->Rtime + 0.5 * r->w_snap * uRppu;
    } else {
      qu[9] = 0.5 * r->Rtime + 0.5 * r->w_snap * uRpu;
      quu[99] = 0.5 * r->w_snap * uRppu;
    }
  }
}

/* DDP:1455-1604 computecrelatedminvo: c, cx, cu for every segment */
static void computecrelatedminvo(ref_t* r) {
  for (int i = 0; i < r->N; i++) {
    const double* x = r->x + (size_t)i * 9;
    const double* u = r->u + (size_t)i * 10;
    double* c = r->c + (size_t)i * r->ncmax;
    double* cx = r->cx + (size_t)i * r->ncmax * 9;
    double* cu = r->cu + (size_t)i * r->ncmax * 10;
    int P = r->np[i];
    int nc = computecminvo(r, x, u, i, c);
    const double* pl = r->planes + (size_t)i * r->pmax * 4;
    double P6[6][6], V6[5][6], A6[4][6];
    scaled_tables(r, u[9], P6, V6, A6); /* members left behind by computecminvo: quirk Q13 */
    /* tempv = [barEkinv x ; u.head(9)]  (DDP:1489-1491) as C_i (6 x 3) */
    double Cc[6][3];
    for (int d = 0; d < 3; d++) {
      for (int k = 0; k < 3; k++) Cc[k][d] = EK_INV[k] * x[k * 3 + d];
      for (int k = 3; k < 6; k++) Cc[k][d] = u[(k - 3) * 3 + d];
    }
    double Tk = u[9], Tk2 = Tk * Tk, Tk3 = Tk2 * Tk, Tk4 = Tk3 * Tk;
    double tp[5] = {1.0, Tk, Tk2, Tk3, Tk4};
    /* d/dT tables (DDP:1543-1561): always MINVO unless the non-parity flag asks for exact */
    double D6[6][6], DV[5][6], DA[4][6];
    for (int j = 0; j < 6; j++)
      for (int k = 0; k < 6; k++) {
        if (r->exact_dt)
          D6[j][k] = (k >= 1) ? k * r->M6[j][k] * tp[k - 1] : 0.0;
        else
          D6[j][k] = (k >= 1) ? MINVO6_DT[j][k] * tp[k - 1] : 0.0;
      }
    for (int j = 0; j < 5; j++)
      for (int k = 0; k < 6; k++) {
        if (r->exact_dt)
          DV[j][k] = (k >= 2) ? (k - 1) * r->Mv[j][k] * tp[k - 2] : 0.0;
        else
          DV[j][k] = (k >= 2) ? MINVO_V6_DT[j][k] * tp[k - 2] : 0.0;
      }
    for (int j = 0; j < 4; j++)
      for (int k = 0; k < 6; k++) {
        if (r->exact_dt)
          DA[j][k] = (k >= 3) ? (k - 2) * r->Ma[j][k] * tp[k - 3] : 0.0;
        else
          DA[j][k] = (k >= 3) ? MINVO_A6_DT[j][k] * tp[k - 3] : 0.0;
      }
    mat_zero(cx, nc * 9);
    mat_zero(cu, nc * 10);
    /* position rows: hatAbarpoly2minvotau(j*P+ld, k*3+d) = P6(j,k) * plane_ld(d)  (DDP:1468-1475);
     * times temp = blkdiag(barEkinv, I9) (DDP:1485-1487, 1496, 1596) */
    for (int j = 0; j < 6; j++)
      for (int ld = 0; ld < P; ld++) {
        int row = j * P + ld;
        for (int k = 0; k < 3; k++)
          for (int d = 0; d < 3; d++) cx[row * 9 + k * 3 + d] = P6[j][k] * pl[ld * 4 + d] * EK_INV[k];
        for (int k = 3; k < 6; k++)
          for (int d = 0; d < 3; d++) cu[row * 10 + (k - 3) * 3 + d] = P6[j][k] * pl[ld * 4 + d];
        double t = 0.0; /* hatAbarpoly2minvotau_dt * tempv  (DDP:1566-1573, 1596) */
        for (int k = 0; k < 6; k++)
          for (int d = 0; d < 3; d++) t += D6[j][k] * pl[ld * 4 + d] * Cc[k][d];
        cu[row * 10 + 9] = t;
      }
    int base = 6 * P;
    /* velocity rows (DDP:1494, 1591-1592, 1597-1598) */
    for (int j = 0; j < 5; j++)
      for (int d = 0; d < 3; d++) {
        int rp = base + j * 3 + d, rm = base + 15 + j * 3 + d;
        for (int k = 1; k < 3; k++) {
          cx[rp * 9 + k * 3 + d] = V6[j][k] * EK_INV[k];
          cx[rm * 9 + k * 3 + d] = -V6[j][k] * EK_INV[k];
        }
        for (int k = 3; k < 6; k++) {
          cu[rp * 10 + (k - 3) * 3 + d] = V6[j][k];
          cu[rm * 10 + (k - 3) * 3 + d] = -V6[j][k];
        }
        double t = 0.0;
        for (int k = 2; k < 6; k++) t += DV[j][k] * Cc[k][d];
        cu[rp * 10 + 9] = t;
        cu[rm * 10 + 9] = -t;
      }
    base += 30;
    /* acceleration rows (DDP:1495, 1593-1594, 1599-1600) */
    for (int j = 0; j < 4; j++)
      for (int d = 0; d < 3; d++) {
        int rp = base + j * 3 + d, rm = base + 12 + j * 3 + d;
        cx[rp * 9 + 2 * 3 + d] = A6[j][2] * EK_INV[2];
        cx[rm * 9 + 2 * 3 + d] = -A6[j][2] * EK_INV[2];
        for (int k = 3; k < 6; k++) {
          cu[rp * 10 + (k - 3) * 3 + d] = A6[j][k];
          cu[rm * 10 + (k - 3) * 3 + d] = -A6[j][k];
        }
        double t = 0.0;
        for (int k = 3; k < 6; k++) t += DA[j][k] * Cc[k][d];
        cu[rp * 10 + 9] = t;
        cu[rm * 10 + 9] = -t;
      }
    base += 24;
    cu[base * 10 + 9] = -1.0; /* DDP:1601 */
  }
}

/* DDP:1309-1316 computeall */
static void computeall(ref_t* r) {
  computeprelated(r);
  computefrelated(r);
  computeqrelated(r);
  computecrelatedminvo(r);
}

/* DDP:1608-1620 initialroll */
static void initialroll(ref_t* r) {
  double qs = 0.0;
  for (int i = 0; i < r->N; i++) {
    const double* x = r->x + (size_t)i * 9;
    const double* u = r->u + (size_t)i * 10;
    r->nc[i] = computecminvo(r, x, u, i, r->c + (size_t)i * r->ncmax);
    r->q[i] = computeq(r, u);
    computenextx(x, u, r->x + (size_t)(i + 1) * 9);
  }
  for (int i = 0; i < r->N; i++) qs += r->q[i];
  r->cost = qs + computep(r, r->x + (size_t)r->N * 9);
  r->costq = qs;
}

/* DDP:1636-1662 resetfilter */
static void resetfilter(ref_t* r) {
  r->logcost = r->cost;
  r->err = 0.0;
  if (r->infeas) {
    for (int i = 0; i < r->N; i++) {
      const double* y = r->y + (size_t)i * r->ncmax;
      const double* c = r->c + (size_t)i * r->ncmax;
      double sl = 0.0, e = 0.0;
      for (int j = 0; j < r->nc[i]; j++) {
        sl += log(y[j]);
        e += fabs(c[j] + y[j]);
      }
      r->logcost -= r->mu * sl;
      r->err += e;
    }
    if (r->err < r->tol) r->err = 0.0;
  } else {
    for (int i = 0; i < r->N; i++) {
      const double* c = r->c + (size_t)i * r->ncmax;
      double sl = 0.0;
      for (int j = 0; j < r->nc[i]; j++) sl += log(-c[j]);
      r->logcost -= r->mu * sl;
      r->err = 0.0;
    }
  }
  r->nfilter = 1;
  r->filter[0] = r->logcost;
  r->filter[1] = r->err;
  r->step = 0;
  r->fp_failed = 0;
}

/* DDP:440-644 backwardpass */
static void backwardpass(ref_t* r) {
  int N = r->N;
  double dV[2] = {0.0, 0.0};
  double c_err = 0.0, mu_err = 0.0, Qu_err = 0.0;
  /* DDP:452-474 regulariser schedule */
  if (r->fp_failed || r->bp_failed) {
    r->reg = r->reg + 1.0;
  } else {
    if (r->step == 0)
      r->reg = r->reg - 1.0;
    else if (r->step <= 3)
      r->reg = r->reg;
    else
      r->reg = r->reg + 1.0;
  }
  if (r->reg < 0.0)
    r->reg = 0.0;
  else if (r->reg > 24.0)
    r->reg = 24.0;
  if (!r->fp_failed) computeall(r); /* DDP:476-478 */

  double Vx[9], Vxx[81];
  memcpy(Vx, r->px, sizeof Vx);
  memcpy(Vxx, r->pxx, sizeof Vxx);
  int ncm = r->ncmax;
  double* SDcu = r->bw_scratch; /* preallocated in ref_begin: [ncm*10 | ncm*9 | ncm*4 | ncm*9 | ncm] */
  double* SDcx = SDcu + (size_t)ncm * 10;
  double* rr = SDcx + (size_t)ncm * 9;
  double *rhat = rr + ncm, *dinv = rr + 2 * ncm, *tv2 = rr + 3 * ncm;
  double* cxpcuKu = rr + (size_t)ncm * 4;
  double* cuiku = cxpcuKu + (size_t)ncm * 9;

  for (int i = N - 1; i >= 0; i--) {
    int nc = r->nc[i];
    const double* fx = r->fx + (size_t)i * 81;
    const double* fu = r->fu + (size_t)i * 90;
    const double* cx = r->cx + (size_t)i * ncm * 9;
    const double* cu = r->cu + (size_t)i * ncm * 10;
    const double* c = r->c + (size_t)i * ncm;
    const double* s = r->s + (size_t)i * ncm;
    const double* y = r->y + (size_t)i * ncm;
    const double* qu = r->qu + (size_t)i * 10;
    const double* quu = r->quu + (size_t)i * 100;
    double Qx[9], Qu[10], Qxx[81], Qxu[90], Quu[100], fxV[81], fuV[90], t9[9], t10[10];
    /* DDP:508-509 (qx = 0) */
    mtm(Qx, cx, s, nc, 9, 1);
    mtm(t9, fx, Vx, 9, 9, 1);
    for (int a = 0; a < 9; a++) Qx[a] = Qx[a] + t9[a];
    mtm(Qu, cu, s, nc, 10, 1);
    mtm(t10, fu, Vx, 9, 10, 1);
    for (int a = 0; a < 10; a++) Qu[a] = qu[a] + Qu[a] + t10[a];
    /* DDP:517-521 (qxx = qxu = 0, tensor terms zero: quirk Q2) */
    mtm(fxV, fx, Vxx, 9, 9, 9);
    mm(Qxx, fxV, fx, 9, 9, 9);
    mm(Qxu, fxV, fu, 9, 9, 10);
    mtm(fuV, fu, Vxx, 9, 10, 9);
    mm(Quu, fuV, fu, 10, 9, 10);
    for (int a = 0; a < 100; a++) Quu[a] += quu[a];
    for (int a = 0; a < 10; a++) /* exact symmetrisation (Appendix B, quirk Q14) */
      for (int b = 0; b < a; b++) {
        double m = 0.5 * (Quu[a * 10 + b] + Quu[b * 10 + a]);
        Quu[a * 10 + b] = m;
        Quu[b * 10 + a] = m;
      }
    double Quu_reg[100], cDc[100], L[100], tempQux[90], kK[100], Ku[90], ku[10];
    double lam = pow(r->reg_exp_base, r->reg) - 1.0; /* DDP:529 */
    memcpy(Quu_reg, Quu, sizeof Quu);
    for (int a = 0; a < 10; a++) Quu_reg[a * 10 + a] += lam;

    if (r->infeas) { /* DDP:532-579 */
      for (int j = 0; j < nc; j++) {
        rr[j] = s[j] * y[j] - r->mu;
        rhat[j] = s[j] * (c[j] + y[j]) - rr[j];
        dinv[j] = 1.0 / y[j];
      }
      for (int j = 0; j < nc; j++) {
        double d = s[j] * dinv[j];
        for (int a = 0; a < 10; a++) SDcu[j * 10 + a] = d * cu[j * 10 + a];
        for (int a = 0; a < 9; a++) SDcx[j * 9 + a] = d * cx[j * 9 + a];
      }
      mtm(cDc, cu, SDcu, nc, 10, 10);
      double A[100];
      for (int a = 0; a < 100; a++) A[a] = Quu_reg[a] + cDc[a];
      if (!llt_lower(L, A, 10)) {
        r->bp_failed = 1;
        r->opterr = INFINITY;
        goto done;
      }
      for (int j = 0; j < nc; j++) tv2[j] = dinv[j] * rhat[j];
      mtm(t10, cu, tv2, nc, 10, 1);
      for (int a = 0; a < 10; a++) Qu[a] += t10[a];
      mtm(tempQux, cu, SDcx, nc, 10, 9);
      for (int a = 0; a < 10; a++)
        for (int b = 0; b < 9; b++) tempQux[a * 9 + b] += Qxu[b * 10 + a];
      for (int a = 0; a < 10; a++) {
        kK[a * 10] = Qu[a];
        for (int b = 0; b < 9; b++) kK[a * 10 + 1 + b] = tempQux[a * 9 + b];
      }
      llt_solve(L, kK, 10, 10);
      for (int a = 0; a < 10; a++) {
        ku[a] = -kK[a * 10];
        for (int b = 0; b < 9; b++) Ku[a * 9 + b] = -kK[a * 10 + 1 + b];
      }
      mm(cuiku, cu, ku, nc, 10, 1);
      mm(cxpcuKu, cu, Ku, nc, 10, 9);
      for (int j = 0; j < nc * 9; j++) cxpcuKu[j] += cx[j];
      double* ks = r->ks + (size_t)i * ncm;
      double* ky = r->ky + (size_t)i * ncm;
      double* Ks = r->Ks + (size_t)i * ncm * 9;
      double* Ky = r->Ky + (size_t)i * ncm * 9;
      for (int j = 0; j < nc; j++) {
        ks[j] = dinv[j] * (rhat[j] + s[j] * cuiku[j]);
        ky[j] = -(c[j] + y[j]) - cuiku[j];
        double d = s[j] * dinv[j];
        for (int a = 0; a < 9; a++) {
          Ks[j * 9 + a] = d * cxpcuKu[j * 9 + a];
          Ky[j * 9 + a] = -cxpcuKu[j * 9 + a];
        }
      }
      for (int a = 0; a < 100; a++) Quu[a] = Quu[a] + cDc[a];
      for (int a = 0; a < 9; a++)
        for (int b = 0; b < 10; b++) Qxu[a * 10 + b] = tempQux[b * 9 + a];
      double t81[81];
      mtm(t81, cx, SDcx, nc, 9, 9);
      for (int a = 0; a < 81; a++) Qxx[a] += t81[a];
      mtm(t9, cx, tv2, nc, 9, 1);
      for (int a = 0; a < 9; a++) Qx[a] += t9[a];
    } else { /* DDP:581-619 */
      for (int j = 0; j < nc; j++) {
        rr[j] = s[j] * c[j] + r->mu;
        dinv[j] = 1.0 / c[j];
      }
      for (int j = 0; j < nc; j++) {
        double d = s[j] * dinv[j];
        for (int a = 0; a < 10; a++) SDcu[j * 10 + a] = d * cu[j * 10 + a];
        for (int a = 0; a < 9; a++) SDcx[j * 9 + a] = d * cx[j * 9 + a];
      }
      mtm(cDc, cu, SDcu, nc, 10, 10);
      double A[100];
      for (int a = 0; a < 100; a++) A[a] = Quu_reg[a] - cDc[a];
      if (!llt_lower(L, A, 10)) {
        r->bp_failed = 1;
        r->opterr = INFINITY;
        goto done;
      }
      for (int j = 0; j < nc; j++) tv2[j] = dinv[j] * rr[j];
      mtm(t10, cu, tv2, nc, 10, 1);
      for (int a = 0; a < 10; a++) Qu[a] -= t10[a];
      mtm(tempQux, cu, SDcx, nc, 10, 9);
      for (int a = 0; a < 10; a++)
        for (int b = 0; b < 9; b++) tempQux[a * 9 + b] = Qxu[b * 10 + a] - tempQux[a * 9 + b];
      for (int a = 0; a < 10; a++) {
        kK[a * 10] = Qu[a];
        for (int b = 0; b < 9; b++) kK[a * 10 + 1 + b] = tempQux[a * 9 + b];
      }
      llt_solve(L, kK, 10, 10);
      for (int a = 0; a < 10; a++) {
        ku[a] = -kK[a * 10];
        for (int b = 0; b < 9; b++) Ku[a * 9 + b] = -kK[a * 10 + 1 + b];
      }
      mm(cuiku, cu, ku, nc, 10, 1);
      mm(cxpcuKu, cu, Ku, nc, 10, 9);
      for (int j = 0; j < nc * 9; j++) cxpcuKu[j] += cx[j];
      double* ks = r->ks + (size_t)i * ncm;
      double* ky = r->ky + (size_t)i * ncm;
      double* Ks = r->Ks + (size_t)i * ncm * 9;
      double* Ky = r->Ky + (size_t)i * ncm * 9;
      for (int j = 0; j < nc; j++) {
        ks[j] = -(dinv[j] * (rr[j] + s[j] * cuiku[j]));
        ky[j] = 0.0;
        double d = s[j] * dinv[j];
        for (int a = 0; a < 9; a++) {
          Ks[j * 9 + a] = -(d * cxpcuKu[j * 9 + a]);
          Ky[j * 9 + a] = 0.0;
        }
      }
      for (int a = 0; a < 100; a++) Quu[a] = Quu[a] - cDc[a];
      for (int a = 0; a < 9; a++)
        for (int b = 0; b < 10; b++) Qxu[a * 10 + b] = tempQux[b * 9 + a];
      double t81[81];
      mtm(t81, cx, SDcx, nc, 9, 9);
      for (int a = 0; a < 81; a++) Qxx[a] -= t81[a];
      mtm(t9, cx, tv2, nc, 9, 1);
      for (int a = 0; a < 9; a++) Qx[a] -= t9[a];
    }
    /* DDP:620-628 */
    for (int a = 0; a < 10; a++) dV[0] += ku[a] * Qu[a];
    double QxuKu[81], KutQuu[90], Quuku[10];
    mm(QxuKu, Qxu, Ku, 9, 10, 9);
    mtm(KutQuu, Ku, Quu, 10, 9, 10);
    mm(Quuku, Quu, ku, 10, 10, 1);
    for (int a = 0; a < 10; a++) dV[1] += 0.5 * ku[a] * Quuku[a];
    double KutQu[9], KutQuuku[9], Qxuku[9], KutQuuKu[81];
    mtm(KutQu, Ku, Qu, 10, 9, 1);
    mm(KutQuuku, KutQuu, ku, 9, 10, 1);
    mm(Qxuku, Qxu, ku, 9, 10, 1);
    mm(KutQuuKu, KutQuu, Ku, 9, 10, 9);
    for (int a = 0; a < 9; a++) Vx[a] = Qx[a] + KutQu[a] + KutQuuku[a] + Qxuku[a];
    for (int a = 0; a < 9; a++)
      for (int b = 0; b < 9; b++) Vxx[a * 9 + b] = Qxx[a * 9 + b] + QxuKu[b * 9 + a] + QxuKu[a * 9 + b] + KutQuuKu[a * 9 + b];
    for (int a = 0; a < 9; a++)
      for (int b = 0; b < a; b++) {
        double m = 0.5 * (Vxx[a * 9 + b] + Vxx[b * 9 + a]);
        Vxx[a * 9 + b] = m;
        Vxx[b * 9 + a] = m;
      }
    memcpy(r->ku + (size_t)i * 10, ku, sizeof ku);
    memcpy(r->Ku + (size_t)i * 90, Ku, sizeof Ku);
    /* DDP:633-637 */
    for (int a = 0; a < 10; a++) Qu_err = fmax(Qu_err, fabs(Qu[a]));
    for (int j = 0; j < nc; j++) mu_err = fmax(mu_err, fabs(rr[j]));
    if (r->infeas)
      for (int j = 0; j < nc; j++) c_err = fmax(c_err, fabs(c[j] + y[j]));
  }
  r->bp_failed = 0;
  r->opterr = fmax(fmax(Qu_err, c_err), mu_err);
  r->dV[0] = dV[0];
  r->dV[1] = dV[1];
done:
  return;
}

/* DDP:647-778 forwardpass */
static void forwardpass(ref_t* r) {
  int N = r->N, ncm = r->ncmax;
  size_t nx = (size_t)(N + 1) * 9, nu = (size_t)N * 10, ncs = (size_t)N * ncm;
  memcpy(r->xn, r->x, sizeof(double) * nx);
  memcpy(r->un, r->u, sizeof(double) * nu);
  memcpy(r->cn, r->c, sizeof(double) * ncs);
  memcpy(r->yn, r->y, sizeof(double) * ncs);
  memcpy(r->sn, r->s, sizeof(double) * ncs);
  for (int i = 0; i < N; i++) r->qn[i] = 0.0;
  double cost = 0, costq = 0, logcost = 0, stepsize = 0, err = 0;
  double tau = fmax(0.99, 1 - r->mu);
  int step, failed = 0;
  for (step = 0; step < 11; step++) {
    failed = 0;
    stepsize = pow(2.0, -(double)step); /* DDP:670 */
    memcpy(r->xn, r->x, sizeof(double) * 9);
    for (int i = 0; i < N; i++) {
      int nc = r->nc[i];
      const double *xo = r->x + (size_t)i * 9, *uo = r->u + (size_t)i * 10;
      const double *so = r->s + (size_t)i * ncm, *yo = r->y + (size_t)i * ncm, *co = r->c + (size_t)i * ncm;
      double *xni = r->xn + (size_t)i * 9, *uni = r->un + (size_t)i * 10;
      double *sni = r->sn + (size_t)i * ncm, *yni = r->yn + (size_t)i * ncm, *cni = r->cn + (size_t)i * ncm;
      const double *ks = r->ks + (size_t)i * ncm, *ky = r->ky + (size_t)i * ncm;
      const double *Ks = r->Ks + (size_t)i * ncm * 9, *Ky = r->Ky + (size_t)i * ncm * 9;
      const double *ku = r->ku + (size_t)i * 10, *Ku = r->Ku + (size_t)i * 90;
      double dx[9];
      for (int a = 0; a < 9; a++) dx[a] = xni[a] - xo[a];
      if (r->infeas) { /* DDP:679-691 */
        for (int j = 0; j < nc; j++) {
          double ay = 0.0, as = 0.0;
          for (int a = 0; a < 9; a++) {
            ay += Ky[j * 9 + a] * dx[a];
            as += Ks[j * 9 + a] * dx[a];
          }
          yni[j] = yo[j] + stepsize * ky[j] + ay;
          sni[j] = so[j] + stepsize * ks[j] + as;
        }
        for (int j = 0; j < nc; j++)
          if (yni[j] < (1 - tau) * yo[j] || sni[j] < (1 - tau) * so[j]) failed = 1;
        if (failed) break;
      } else { /* DDP:693-706 */
        for (int j = 0; j < nc; j++) {
          double as = 0.0;
          for (int a = 0; a < 9; a++) as += Ks[j * 9 + a] * dx[a];
          sni[j] = so[j] + stepsize * ks[j] + as;
        }
      }
      for (int a = 0; a < 10; a++) {
        double t = 0.0;
        for (int b = 0; b < 9; b++) t += Ku[a * 9 + b] * dx[b];
        uni[a] = uo[a] + stepsize * ku[a] + t;
      }
      if (!r->infeas) {
        computecminvo(r, xni, uni, i, cni);
        for (int j = 0; j < nc; j++)
          if (cni[j] > (1 - tau) * co[j] || sni[j] < (1 - tau) * so[j]) failed = 1;
        if (failed) break;
      }
      computenextx(xni, uni, r->xn + (size_t)(i + 1) * 9);
    }
    if (failed) continue;
    /* DDP:712-734 */
    double qs = 0.0;
    for (int i = 0; i < N; i++) {
      r->qn[i] = computeq(r, r->un + (size_t)i * 10);
      qs += r->qn[i];
    }
    cost = qs + computep(r, r->xn + (size_t)N * 9);
    costq = qs;
    logcost = cost;
    err = 0.0;
    if (r->infeas) {
      for (int i = 0; i < N; i++) {
        double *cni = r->cn + (size_t)i * ncm, *yni = r->yn + (size_t)i * ncm;
        double sl = 0.0, e = 0.0;
        for (int j = 0; j < r->nc[i]; j++) sl += log(yni[j]);
        logcost -= r->mu * sl;
        computecminvo(r, r->xn + (size_t)i * 9, r->un + (size_t)i * 10, i, cni);
        for (int j = 0; j < r->nc[i]; j++) e += fabs(cni[j] + yni[j]);
        err += e;
      }
      err = fmax(r->tol, err);
    } else {
      for (int i = 0; i < N; i++) {
        double* cni = r->cn + (size_t)i * ncm;
        computecminvo(r, r->xn + (size_t)i * 9, r->un + (size_t)i * 10, i, cni);
        double sl = 0.0;
        for (int j = 0; j < r->nc[i]; j++) sl += log(-cni[j]);
        logcost -= r->mu * sl;
      }
      err = 0.0;
    }
    /* DDP:737-757 filter */
    int nkeep = 0;
    for (int i = 0; i < r->nfilter; i++) {
      double f0 = r->filter[2 * i], f1 = r->filter[2 * i + 1];
      if (logcost >= f0 && err >= f1) {
        failed = 1;
        break;
      } else if (logcost > f0 || err > f1) {
        /* keep: compacting in place is safe because nkeep <= i and a failed trial leaves
         * the reference's filter untouched -- so compact into a scratch tail instead */
        r->filter[2 * (r->filter_cap / 2 + nkeep)] = f0;
        r->filter[2 * (r->filter_cap / 2 + nkeep) + 1] = f1;
        nkeep++;
      }
    }
    if (failed) continue;
    for (int i = 0; i < nkeep; i++) {
      r->filter[2 * i] = r->filter[2 * (r->filter_cap / 2 + i)];
      r->filter[2 * i + 1] = r->filter[2 * (r->filter_cap / 2 + i) + 1];
    }
    r->filter[2 * nkeep] = logcost;
    r->filter[2 * nkeep + 1] = err;
    r->nfilter = nkeep + 1;
    break;
  }
  if (failed) { /* DDP:760-762 */
    r->fp_failed = 1;
    r->stepsize = 0.0;
  } else { /* DDP:763-776 */
    r->cost = cost;
    r->costq = costq;
    r->logcost = logcost;
    memcpy(r->x, r->xn, sizeof(double) * nx);
    memcpy(r->u, r->un, sizeof(double) * nu);
    memcpy(r->y, r->yn, sizeof(double) * ncs);
    memcpy(r->s, r->sn, sizeof(double) * ncs);
    memcpy(r->c, r->cn, sizeof(double) * ncs);
    memcpy(r->q, r->qn, sizeof(double) * (size_t)N);
    r->err = err;
    r->stepsize = stepsize;
    r->step = step;
    r->fp_failed = 0;
  }
}

/* ---- conversions ------------------------------------------------------------------------ */
/* beztau2polyt = poly2bez * t2tauMat with poly2bez = TEMPM^T, t2tauMat = diag(T^-i)  (DDP:1018-1059, 788) */
static void beztau2polyt(double T, double M[6][6]) {
  double it = 1.0 / T, pw[6];
  pw[0] = 1.0;
  for (int i = 1; i < 6; i++) pw[i] = pw[i - 1] * it;
  for (int a = 0; a < 6; a++)
    for (int b = 0; b < 6; b++) M[a][b] = TEMPM[b][a] * pw[b];
}
/* DDP:782-796 bez2polyFunc for one segment.  bez_il = interleaved row [b0xyz .. b5xyz]. */
static void bez2poly_row(const double* bez_il, double T, double* poly) {
  double M[6][6];
  beztau2polyt(T, M);
  /* Map<MatrixXd>(tempv, 3, 6) column-major: Bm(d, j) = T * bez_il[j*3+d]; P = Bm * M (3x6) stored col-major */
  for (int j = 0; j < 6; j++)
    for (int d = 0; d < 3; d++) {
      double a = 0.0;
      for (int l = 0; l < 6; l++) a += (T * bez_il[l * 3 + d]) * M[l][j];
      poly[j * 3 + d] = a;
    }
}
static int inv6(double A[6][6], double Ai[6][6]) {
  /* Gauss-Jordan with partial pivoting (Eigen's .inverse() uses PartialPivLU for 6x6) */
  double w[6][12];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      w[i][j] = A[i][j];
      w[i][6 + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int c = 0; c < 6; c++) {
    int p = c;
    for (int i = c + 1; i < 6; i++)
      if (fabs(w[i][c]) > fabs(w[p][c])) p = i;
    if (w[p][c] == 0.0) return 0;
    if (p != c)
      for (int j = 0; j < 12; j++) {
        double t = w[c][j];
        w[c][j] = w[p][j];
        w[p][j] = t;
      }
    double d = 1.0 / w[c][c];
    for (int j = 0; j < 12; j++) w[c][j] *= d;
    for (int i = 0; i < 6; i++)
      if (i != c) {
        double f = w[i][c];
        if (f != 0.0)
          for (int j = 0; j < 12; j++) w[i][j] -= f * w[c][j];
      }
  }
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) Ai[i][j] = w[i][6 + j];
  return 1;
}
/* DDP:799-812 poly2bezFunc for one segment */
static void poly2bez_row(const double* poly, double T, double* bez_il) {
  double M[6][6], Mi[6][6];
  beztau2polyt(T, M);
  inv6(M, Mi);
  for (int j = 0; j < 6; j++)
    for (int d = 0; d < 3; d++) {
      double a = 0.0;
      for (int l = 0; l < 6; l++) a += (1.0 / T * poly[l * 3 + d]) * Mi[l][j];
      bez_il[j * 3 + d] = a;
    }
}
/* DDP:814-823 sysparam2polyFunc for one segment */
static void sysparam2poly_row(const double* x, const double* u, double* poly, double* T) {
  *T = u[9];
  for (int k = 0; k < 3; k++)
    for (int d = 0; d < 3; d++) poly[k * 3 + d] = EK_INV[k] * x[k * 3 + d];
  for (int a = 0; a < 9; a++) poly[9 + a] = u[a];
}

/* ---- life cycle ------------------------------------------------------------------------- */
static double* dalloc(size_t n) {
  double* p = (double*)calloc(n ? n : 1, sizeof(double));
  if (!p) {
    fprintf(stderr, "direct_ref: out of memory\n");
    abort();
  }
  return p;
}

typedef struct {
  int N, pmax;
  const double *x0, *xd, *T0, *planes, *seeds, *init_bez, *init_poly;
  const int* n_planes;
  int infeas;
} ref_problem_t;

static void ref_free(ref_t* r) {
  if (!r) return;
  free(r->nc);
  free(r->x); free(r->u); free(r->c); free(r->s); free(r->y); free(r->q);
  free(r->fx); free(r->fu); free(r->qu); free(r->quu); free(r->cx); free(r->cu);
  free(r->filter);
  free(r->ku); free(r->Ku); free(r->ks); free(r->ky); free(r->Ks); free(r->Ky);
  free(r->xn); free(r->un); free(r->cn); free(r->sn); free(r->yn); free(r->qn); free(r->bw_scratch);
  free(r->trace);
  free(r);
}

/* DDP:28-286: everything of polyCurveGeneration before the outer loop */
static ref_t* ref_begin(const ref_problem_t* pb, const direct_ddp_params_t* pr, int trace_cap) {
  ref_t* r = (ref_t*)calloc(1, sizeof(ref_t));
  int N = pb->N;
  r->N = N;
  r->pmax = pb->pmax;
  r->ncmax = 6 * pb->pmax + 55;
  r->maxiter = pr->iter_max;
  r->tol = 1.0e-7;
  r->infeas = pb->infeas;
  r->infeas_ref = pb->infeas;
  r->line_failed = 1; /* the caller's initial value, TRP:887 */
  r->w_snap = pr->w_snap;
  r->Rtime = pr->w_time;
  r->w_term = pr->w_terminal;
  r->time_power = pr->time_power;
  r->maxVel = pr->max_vel;
  r->maxAcc = pr->max_acc;
  r->minvo = pr->minvo;
  r->zero_init = pr->zero_init;
  r->line_init = pr->line_init;
  r->fixed_iters = pr->fixed_iters;
  r->exact_dt = pr->exact_dt;
  r->reg_exp_base = pr->zero_init ? 1.6 : 4.0; /* DDP:60-61 */
  memcpy(r->M6, pr->minvo ? MINVO6 : BEZ6, sizeof r->M6);
  memcpy(r->Mv, pr->minvo ? MINVO_V6 : BEZ_V6, sizeof r->Mv);
  memcpy(r->Ma, pr->minvo ? MINVO_A6 : BEZ_A6, sizeof r->Ma);
  memcpy(r->xd, pb->xd, sizeof r->xd);
  r->np = pb->n_planes;
  r->planes = pb->planes;
  r->seeds = pb->seeds;
  int ncm = r->ncmax;
  r->nc = (int*)calloc((size_t)N, sizeof(int));
  r->x = dalloc((size_t)(N + 1) * 9);
  r->u = dalloc((size_t)N * 10);
  r->c = dalloc((size_t)N * ncm);
  r->s = dalloc((size_t)N * ncm);
  r->y = dalloc((size_t)N * ncm);
  r->q = dalloc((size_t)N);
  r->fx = dalloc((size_t)N * 81);
  r->fu = dalloc((size_t)N * 90);
  r->qu = dalloc((size_t)N * 10);
  r->quu = dalloc((size_t)N * 100);
  r->cx = dalloc((size_t)N * ncm * 9);
  r->cu = dalloc((size_t)N * ncm * 10);
  r->ku = dalloc((size_t)N * 10);
  r->Ku = dalloc((size_t)N * 90);
  r->ks = dalloc((size_t)N * ncm);
  r->ky = dalloc((size_t)N * ncm);
  r->Ks = dalloc((size_t)N * ncm * 9);
  r->Ky = dalloc((size_t)N * ncm * 9);
  r->xn = dalloc((size_t)(N + 1) * 9);
  r->un = dalloc((size_t)N * 10);
  r->cn = dalloc((size_t)N * ncm);
  r->sn = dalloc((size_t)N * ncm);
  r->yn = dalloc((size_t)N * ncm);
  r->qn = dalloc((size_t)N);
  r->bw_scratch = dalloc((size_t)ncm * 33);
  r->filter_cap = 2 * (pr->iter_max + 8);
  r->filter = dalloc((size_t)2 * r->filter_cap);
  r->trace_cap = trace_cap;
  if (trace_cap > 0) r->trace = dalloc((size_t)trace_cap * TRACE_W);

  memcpy(r->x, pb->x0, sizeof(double) * 9); /* DDP:115-122 */
  for (int i = 0; i < N; i++) {            /* DDP:124-160 */
    r->u[(size_t)i * 10 + 9] = pb->T0[i];
    r->nc[i] = 6 * r->np[i] + 55;
    for (int j = 0; j < r->nc[i]; j++) {
      r->s[(size_t)i * ncm + j] = 1.0e-1;
      r->y[(size_t)i * ncm + j] = 0.01;
    }
  }
  /* DDP:167-193: initial u from the Bezier warm start */
  if (!pr->zero_init) {
    if (!pr->line_init && pb->init_poly) { /* C-ABI extension: monomial warm start (no reference counterpart) */
      for (int i = 0; i < N; i++)
        for (int a = 0; a < 9; a++) r->u[(size_t)i * 10 + a] = pb->init_poly[(size_t)i * 18 + 9 + a];
    } else if (!pr->line_init && !pb->init_bez) {
      /* no warm start handed over: the controls stay zero, as direct_ddp_batch_in_t defines it (the reference's caller
       * always passes initbezCoeff, TRP:918-921) */
    } else if (!pr->line_init) {
      for (int i = 0; i < N; i++) {
        double il[18], poly[18];
        const double* row = pb->init_bez + (size_t)i * 18;
        /* DDP:170-176: Map(6,3) col-major -> transpose -> flatten: il[j*3+d] = row[d*6+j] */
        for (int j = 0; j < 6; j++)
          for (int d = 0; d < 3; d++) il[j * 3 + d] = row[d * 6 + j];
        bez2poly_row(il, pb->T0[i], poly);
        for (int a = 0; a < 9; a++) r->u[(size_t)i * 10 + a] = poly[9 + a];
      }
    } else { /* DDP:194-248 line initialisation */
      for (int l = 0; l < N; l++) {
        double pa[3], pbn[3];
        for (int d = 0; d < 3; d++) {
          pa[d] = (l == 0) ? pb->x0[d] : pb->seeds[(size_t)l * 3 + d];
          pbn[d] = (l == N - 1) ? pb->xd[d] : pb->seeds[(size_t)(l + 1) * 3 + d];
        }
        int vio = 1, cnt = 0;
        double* ul = r->u + (size_t)l * 10;
        double* cons = dalloc((size_t)ncm);
        while (vio && cnt <= 4) {
          double Tk = ul[9], Tk2 = Tk * Tk, Tk3 = Tk2 * Tk, Tk4 = Tk3 * Tk, Tk5 = Tk4 * Tk;
          double Gi[3][3] = {{10.0 / Tk3, -4.0 / Tk2, 0.5 / Tk}, {-15.0 / Tk4, 7.0 / Tk3, -1.0 / Tk2}, {6.0 / Tk5, -3.0 / Tk4, 0.5 / Tk3}};
          double Fk[3][3] = {{1.0, Tk, Tk2 / 2.0}, {0.0, 1.0, Tk}, {0.0, 0.0, 1.0}};
          double xn[9] = {0}, xc[9] = {0}, t[9];
          for (int d = 0; d < 3; d++) {
            xn[d] = pbn[d];
            xc[d] = pa[d];
          }
          for (int a = 0; a < 3; a++)
            for (int d = 0; d < 3; d++) {
              double f = 0.0;
              for (int b = 0; b < 3; b++) f += Fk[a][b] * xc[b * 3 + d];
              t[a * 3 + d] = xn[a * 3 + d] - f;
            }
          for (int a = 0; a < 3; a++)
            for (int d = 0; d < 3; d++) {
              double g = 0.0;
              for (int b = 0; b < 3; b++) g += Gi[a][b] * t[b * 3 + d];
              ul[a * 3 + d] = g;
            }
          int nc = computecminvo(r, xc, ul, l, cons);
          int all_neg = 1;
          for (int j = 0; j < nc; j++)
            if (!(cons[j] < 0)) all_neg = 0;
          if (all_neg)
            vio = 0;
          else {
            ul[9] = 2 * Tk;
            cnt++;
          }
        }
        free(cons);
      }
    }
  }
  initialroll(r); /* DDP:252 */
  if (pr->line_init) { /* DDP:255-269 */
    int count = 0;
    for (int i = 0; i < N; i++)
      for (int j = 0; j < r->nc[i]; j++)
        if (r->c[(size_t)i * ncm + j] > 0) count++;
    if (count == 0) r->infeas = 0;
  }
  r->prev_cost = r->cost; /* costTraj / costqTraj first entries, DDP:277-278 */
  r->prev_costq = r->costq;
  r->mu = r->cost / r->N / r->nc[0]; /* DDP:281, quirk Q6 */
  resetfilter(r);
  r->reg = 0.0; /* resetreg, DDP:1675-1680 */
  r->bp_failed = 0;
  if (pr->line_init) r->reg = 10.0; /* DDP:284-286 */
  r->iter = 0;
  r->rtn = 0;
  return r;
}

/* One trip of the outer loop, DDP:295-412.  Returns 1 when the loop breaks. */
static int ref_iterate_once(ref_t* r) {
  int N = r->N, ncm = r->ncmax;
  int n_sweeps = 0;
  while (1) { /* DDP:297-310 */
    backwardpass(r);
    n_sweeps++;
    if (!r->bp_failed) break;
    if (r->reg == 24 && r->bp_failed)
      r->bp_no_upd_count++;
    else
      r->bp_no_upd_count = 0;
    if (r->bp_no_upd_count > 20) break;
  }
  forwardpass(r);
  r->fwd_passes++;
  if (r->trace && r->iter < r->trace_cap) {
    double* t = r->trace + (size_t)r->iter * TRACE_W;
    t[0] = r->cost; t[1] = r->costq; t[2] = r->logcost; t[3] = r->err; t[4] = r->mu;
    t[5] = r->reg; t[6] = r->step; t[7] = r->opterr; t[8] = r->stepsize; t[9] = r->fp_failed;
    t[10] = n_sweeps; t[11] = r->bp_failed; /* backward sweeps of the retry loop; did it give up (DDP:306-309) */
  }
  /* DDP:314-326 negative time */
  int timePosiInd = 1;
  for (int i = 0; i < N; i++)
    if (r->u[(size_t)i * 10 + 9] < 0) timePosiInd = 0;
  if (!timePosiInd) {
    r->rtn = -3;
    return 1;
  }
  double prev_cost = r->prev_cost, prev_costq = r->prev_costq; /* costTraj.end()[-2] after push */
  r->prev_cost = r->cost;
  r->prev_costq = r->costq;
  /* DDP:335-338 */
  if (!r->fixed_iters && fmax(r->opterr, r->mu) <= r->tol) return 1;
  /* DDP:340-344 */
  if (r->opterr <= 0.2 * r->mu) {
    r->mu = fmax(r->tol / 10.0, fmin(0.2 * r->mu, pow(r->mu, 1.2)));
    resetfilter(r);
    r->reg = 0.0;
    r->bp_failed = 0;
  }
  /* DDP:346-390 */
  int count = 0;
  for (int i = 0; i < N; i++)
    for (int j = 0; j < r->nc[i]; j++)
      if (r->c[(size_t)i * ncm + j] >= 2.0e-4) count++;
  if (count == 0 && !r->fixed_iters) {
    if (r->zero_init) {
      r->infeas_ref = 0;
      r->rtn = 2;
      return 1;
    }
    if (!r->zero_init && !r->line_init) {
      double d = r->cost - prev_cost;
      if (d * d < prev_cost * 1.0e-2 && r->opterr < 5.0e1) {
        r->rtn = 1;
        return 1;
      }
    }
    if (r->line_init) {
      double d = r->cost - prev_cost;
      if (d * d < prev_cost * 0.01) {
        r->line_failed = 0;
        return 1;
      }
    }
  }
  (void)prev_costq; /* opt_no_upd_count is counted but never read in the reference (DDP:366-371) */
  /* DDP:392-396 */
  if (r->bp_no_upd_count > 20) {
    r->rtn = -4;
    return 1;
  }
  /* DDP:398-409 */
  if (r->line_init) {
    if (r->stepsize < 1.0e-6)
      r->no_upd_count++;
    else
      r->no_upd_count = 0;
    if (r->no_upd_count > 100) return 1;
  }
  return 0;
}

/* DDP:295, 412: the for loop */
static void ref_run(ref_t* r) {
  for (r->iter = 0; r->iter < r->maxiter; r->iter++)
    if (ref_iterate_once(r)) break;
}

/* DDP:414-437: finalroll + output conversions.  Any output pointer may be NULL. */
static void ref_finish(ref_t* r, double* bez, double* poly, double* T, double* jerk_cost, double* term_norm2) {
  int N = r->N;
  double jc = 0.0;
  for (int i = 0; i < N; i++) { /* DDP:1624-1634 */
    const double* u = r->u + (size_t)i * 10;
    double R[3][3], Rp[3][3], Rpp[3][3], bR[81];
    time2R(u[9], R, Rp, Rpp);
    kron3(bR, R);
    double a = 0.0;
    for (int p = 0; p < 9; p++) {
      double t = 0.0;
      for (int q = 0; q < 9; q++) t += bR[p * 9 + q] * u[q];
      a += u[p] * t;
    }
    jc += a;
  }
  if (jerk_cost) *jerk_cost = jc;
  if (term_norm2) { /* DDPH:327-330 */
    const double* xN = r->x + (size_t)N * 9;
    double a = 0.0;
    for (int i = 0; i < 9; i++) a += (xN[i] - r->xd[i]) * (xN[i] - r->xd[i]);
    *term_norm2 = a;
  }
  for (int i = 0; i < N; i++) {
    double pr[18], Ti, il[18];
    sysparam2poly_row(r->x + (size_t)i * 9, r->u + (size_t)i * 10, pr, &Ti);
    if (poly) memcpy(poly + (size_t)i * 18, pr, sizeof pr);
    if (T) T[i] = Ti;
    if (bez) {
      poly2bez_row(pr, Ti, il);
      /* DDP:430-436: Map(3,6) col-major -> transpose -> flatten: out[d*6+j] = il[j*3+d] */
      for (int j = 0; j < 6; j++)
        for (int d = 0; d < 3; d++) bez[(size_t)i * 18 + d * 6 + j] = il[j * 3 + d];
    }
  }
}

/* ---- exported API (ctypes) -------------------------------------------------------------- */
typedef struct {
  ref_t* r;
} ref_handle_t;

static void fill_problem(ref_problem_t* pb, const direct_ddp_batch_in_t* in, int b, const direct_ddp_params_t* pr) {
  size_t nm = (size_t)in->n_seg_max;
  pb->N = in->n_seg[b];
  pb->pmax = in->p_max;
  pb->x0 = (const double*)in->x0 + (size_t)b * 9;
  pb->xd = (const double*)in->xd + (size_t)b * 9;
  pb->T0 = (const double*)in->T0 + (size_t)b * nm;
  pb->n_planes = in->n_planes + (size_t)b * nm;
  pb->planes = (const double*)in->planes + (size_t)b * nm * in->p_max * 4;
  pb->seeds = in->seeds ? (const double*)in->seeds + (size_t)b * nm * 3 : NULL;
  pb->init_bez = in->init_bez ? (const double*)in->init_bez + (size_t)b * nm * 18 : NULL;
  pb->init_poly = in->init_poly ? (const double*)in->init_poly + (size_t)b * nm * 18 : NULL;
  pb->infeas = in->infeas_in ? in->infeas_in[b] : pr->infeas;
}

static void write_out(ref_t* r, const direct_ddp_batch_in_t* in, direct_ddp_batch_out_t* out, int b) {
  size_t nm = (size_t)in->n_seg_max;
  double jc, tn;
  double* bez = out->bez ? (double*)out->bez + (size_t)b * nm * 18 : NULL;
  double* poly = out->poly ? (double*)out->poly + (size_t)b * nm * 18 : NULL;
  double* T = out->T ? (double*)out->T + (size_t)b * nm : NULL;
  ref_finish(r, bez, poly, T, &jc, &tn);
  if (out->rtn) out->rtn[b] = r->rtn;
  if (out->iter_used) out->iter_used[b] = r->iter;
  if (out->fwd_passes) out->fwd_passes[b] = r->fwd_passes;
  if (out->infeas_out) out->infeas_out[b] = (uint8_t)r->infeas_ref;
  if (out->line_failed_out) out->line_failed_out[b] = (uint8_t)r->line_failed;
  if (out->cost) ((double*)out->cost)[b] = r->cost;
  if (out->costq) ((double*)out->costq)[b] = r->costq;
  if (out->jerk_cost) ((double*)out->jerk_cost)[b] = jc;
  if (out->terminal_norm2) ((double*)out->terminal_norm2)[b] = tn;
  if (out->opterr) ((double*)out->opterr)[b] = r->opterr;
  if (out->mu) ((double*)out->mu)[b] = r->mu;
}

/* Batched polyCurveGeneration, one problem per OpenMP task.  fp64 host arrays only.
 * trace (may be NULL): [batch][trace_cap][12] per-iteration record. */
int direct_ref_solve_batch(const direct_ddp_params_t* pr, const direct_ddp_batch_in_t* in,
                           direct_ddp_batch_out_t* out, int n_threads, double* trace, int trace_cap) {
  if (!pr || !in || !out) return DIRECT_ERR_INVALID;
  if (pr->time_power != 1 && pr->time_power != 2) return DIRECT_ERR_INVALID;
  int B = in->batch;
  /* keep the per-problem work arrays (a few MB) on the per-thread malloc arenas: with the default
   * mmap threshold every solve would mmap/munmap them and 100+ threads serialise in the kernel */
  mallopt(M_MMAP_THRESHOLD, 1 << 30);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
  for (int b = 0; b < B; b++) {
    ref_problem_t pb;
    fill_problem(&pb, in, b, pr);
    ref_t* r = ref_begin(&pb, pr, trace ? trace_cap : 0);
    ref_run(r);
    write_out(r, in, out, b);
    if (trace) memcpy(trace + (size_t)b * trace_cap * TRACE_W, r->trace, sizeof(double) * (size_t)trace_cap * TRACE_W);
    ref_free(r);
  }
  return DIRECT_OK;
}

/* fastTrajPlanning protocol, TRP:886-921: phase 0, UpdateTime if rtn0 == 2, phase 1. */
int direct_ref_plan_batch(const direct_ddp_params_t* p0, const direct_ddp_params_t* p1,
                          const direct_ddp_batch_in_t* in, direct_ddp_batch_out_t* out0,
                          direct_ddp_batch_out_t* out1, int n_threads) {
  int B = in->batch;
  size_t nm = (size_t)in->n_seg_max;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
  for (int b = 0; b < B; b++) {
    ref_problem_t pb;
    fill_problem(&pb, in, b, p0);
    pb.infeas = 1; /* TRP:886 */
    ref_t* r0 = ref_begin(&pb, p0, 0);
    ref_run(r0);
    double* bez0 = dalloc(nm * 18);
    double* T0n = dalloc(nm);
    ref_finish(r0, bez0, NULL, T0n, NULL, NULL);
    if (out0) write_out(r0, in, out0, b);
    int rtn0 = r0->rtn, infeas = r0->infeas_ref;
    ref_free(r0);
    ref_problem_t pb1 = pb;
    if (rtn0 == 2) pb1.T0 = T0n; /* UpdateTime, TRP:911-915 */
    pb1.init_bez = bez0;         /* TRP:918 */
    pb1.init_poly = NULL;
    pb1.infeas = infeas;
    ref_t* r1 = ref_begin(&pb1, p1, 0);
    ref_run(r1);
    write_out(r1, in, out1, b);
    ref_free(r1);
    free(bez0);
    free(T0n);
  }
  return DIRECT_OK;
}

/* stepwise handle for per-pass parity tests (single problem = batch index b of `in`) */
void* direct_ref_begin(const direct_ddp_params_t* pr, const direct_ddp_batch_in_t* in, int b) {
  ref_problem_t pb;
  fill_problem(&pb, in, b, pr);
  return ref_begin(&pb, pr, 0);
}
void direct_ref_backwardpass(void* h) { backwardpass((ref_t*)h); }
void direct_ref_forwardpass(void* h) { forwardpass((ref_t*)h); }
int direct_ref_iterate(void* h, int n) {
  ref_t* r = (ref_t*)h;
  int done = 0;
  for (int k = 0; k < n && r->iter < r->maxiter; k++) {
    done = ref_iterate_once(r);
    if (done) break;
    r->iter++;
  }
  return done;
}
void direct_ref_computeall(void* h) { computeall((ref_t*)h); }
void direct_ref_end(void* h) { ref_free((ref_t*)h); }
int direct_ref_ncmax(void* h) { return ((ref_t*)h)->ncmax; }
/* field ids follow direct_field_t; extra ids >= 100 are oracle-only */
int direct_ref_get(void* h, int field, double* dst) {
  ref_t* r = (ref_t*)h;
  size_t N = (size_t)r->N, ncm = (size_t)r->ncmax;
  switch (field) {
    case DIRECT_FIELD_X: memcpy(dst, r->x, sizeof(double) * (N + 1) * 9); break;
    case DIRECT_FIELD_U: memcpy(dst, r->u, sizeof(double) * N * 10); break;
    case DIRECT_FIELD_S: memcpy(dst, r->s, sizeof(double) * N * ncm); break;
    case DIRECT_FIELD_Y: memcpy(dst, r->y, sizeof(double) * N * ncm); break;
    case DIRECT_FIELD_C: memcpy(dst, r->c, sizeof(double) * N * ncm); break;
    case DIRECT_FIELD_KU: memcpy(dst, r->ku, sizeof(double) * N * 10); break;
    case DIRECT_FIELD_KUU: memcpy(dst, r->Ku, sizeof(double) * N * 90); break;
    case DIRECT_FIELD_KS: memcpy(dst, r->ks, sizeof(double) * N * ncm); break;
    case DIRECT_FIELD_KY: memcpy(dst, r->ky, sizeof(double) * N * ncm); break;
    case DIRECT_FIELD_SCALARS:
      dst[0] = r->cost; dst[1] = r->costq; dst[2] = r->logcost; dst[3] = r->err; dst[4] = r->mu;
      dst[5] = r->reg; dst[6] = r->opterr; dst[7] = r->stepsize; dst[8] = r->step;
      dst[9] = r->fp_failed; dst[10] = r->bp_failed; dst[11] = r->rtn; dst[12] = r->iter;
      dst[13] = 0; dst[14] = r->nfilter; dst[15] = r->infeas;
      break;
    case 100: memcpy(dst, r->cx, sizeof(double) * N * ncm * 9); break;
    case 101: memcpy(dst, r->cu, sizeof(double) * N * ncm * 10); break;
    case 102: memcpy(dst, r->fx, sizeof(double) * N * 81); break;
    case 103: memcpy(dst, r->fu, sizeof(double) * N * 90); break;
    case 104: memcpy(dst, r->qu, sizeof(double) * N * 10); break;
    case 105: memcpy(dst, r->quu, sizeof(double) * N * 100); break;
    case 106: memcpy(dst, r->Ks, sizeof(double) * N * ncm * 9); break;
    case 107: memcpy(dst, r->Ky, sizeof(double) * N * ncm * 9); break;
    case 108: memcpy(dst, r->filter, sizeof(double) * 2 * (size_t)r->nfilter); break;
    case 109: memcpy(dst, r->q, sizeof(double) * N); break;
    default: return DIRECT_ERR_INVALID;
  }
  return DIRECT_OK;
}
int direct_ref_set(void* h, int field, const double* src) {
  ref_t* r = (ref_t*)h;
  size_t N = (size_t)r->N, ncm = (size_t)r->ncmax;
  switch (field) {
    case DIRECT_FIELD_X: memcpy(r->x, src, sizeof(double) * (N + 1) * 9); break;
    case DIRECT_FIELD_U: memcpy(r->u, src, sizeof(double) * N * 10); break;
    case DIRECT_FIELD_S: memcpy(r->s, src, sizeof(double) * N * ncm); break;
    case DIRECT_FIELD_Y: memcpy(r->y, src, sizeof(double) * N * ncm); break;
    case DIRECT_FIELD_SCALARS:
      r->cost = src[0]; r->costq = src[1]; r->logcost = src[2]; r->err = src[3]; r->mu = src[4];
      r->reg = src[5]; r->opterr = src[6]; r->stepsize = src[7]; r->step = (int)src[8];
      r->fp_failed = (int)src[9]; r->bp_failed = (int)src[10];
      break;
    default: return DIRECT_ERR_INVALID;
  }
  return DIRECT_OK;
}

/* single-segment model evaluations for the analytic tests */
void direct_ref_eval_c(const direct_ddp_params_t* pr, const double* x, const double* u, int P,
                       const double* planes, double* c) {
  ref_t r;
  memset(&r, 0, sizeof r);
  r.N = 1; r.pmax = P; r.ncmax = 6 * P + 55;
  r.maxVel = pr->max_vel; r.maxAcc = pr->max_acc; r.minvo = pr->minvo;
  memcpy(r.M6, pr->minvo ? MINVO6 : BEZ6, sizeof r.M6);
  memcpy(r.Mv, pr->minvo ? MINVO_V6 : BEZ_V6, sizeof r.Mv);
  memcpy(r.Ma, pr->minvo ? MINVO_A6 : BEZ_A6, sizeof r.Ma);
  r.np = &P; r.planes = planes;
  computecminvo(&r, x, u, 0, c);
}
void direct_ref_eval_nextx(const double* x, const double* u, double* xn) { computenextx(x, u, xn); }
double direct_ref_eval_q(const direct_ddp_params_t* pr, const double* u) {
  ref_t r;
  memset(&r, 0, sizeof r);
  r.w_snap = pr->w_snap; r.Rtime = pr->w_time; r.time_power = pr->time_power;
  return computeq(&r, u);
}
void direct_ref_bez2poly(const double* bez_il, double T, double* poly) { bez2poly_row(bez_il, T, poly); }
void direct_ref_poly2bez(const double* poly, double T, double* bez_il) { poly2bez_row(poly, T, bez_il); }

/* initTimeAllocation, TRP:583-639 (v0 = 0, so V0 = aV0 = 0) */
int direct_ref_time_allocation(int batch, int n_seg_max, const int* n_seg, const double* start,
                               const double* goal, const double* seeds, double max_vel,
                               double max_acc, double* T_out) {
  for (int b = 0; b < batch; b++) {
    int N = n_seg[b];
    for (int k = 0; k < N; k++) {
      double p0[3], p1[3];
      for (int d = 0; d < 3; d++) {
        p0[d] = (k == 0) ? start[b * 3 + d] : seeds[((size_t)b * n_seg_max + k) * 3 + d];
        p1[d] = (k == N - 1) ? goal[b * 3 + d] : seeds[((size_t)b * n_seg_max + k + 1) * 3 + d];
      }
      double dd[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
      double D = sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]);
      double V0 = 0.0 * (dd[0] / D) + 0.0 * (dd[1] / D) + 0.0 * (dd[2] / D);
      double aV0 = fabs(V0);
      double _Vel = max_vel, _Acc = max_acc;
      double acct = (_Vel - V0) / _Acc * ((_Vel > V0) ? 1 : -1);
      double accd = V0 * acct + (_Acc * acct * acct / 2) * ((_Vel > V0) ? 1 : -1);
      double dcct = _Vel / _Acc;
      double dccd = _Acc * dcct * dcct / 2;
      double dtxyz;
      if (D < aV0 * aV0 / (2 * _Acc)) {
        double t1 = (V0 < 0) ? 2.0 * aV0 / _Acc : 0.0;
        double t2 = aV0 / _Acc;
        dtxyz = t1 + t2;
      } else if (D < accd + dccd) {
        double t1 = (V0 < 0) ? 2.0 * aV0 / _Acc : 0.0;
        double t2 = (-aV0 + sqrt(aV0 * aV0 + _Acc * D - aV0 * aV0 / 2)) / _Acc;
        double t3 = (aV0 + _Acc * t2) / _Acc;
        dtxyz = t1 + t2 + t3;
      } else {
        double t1 = acct;
        double t2 = (D - accd - dccd) / _Vel;
        double t3 = dcct;
        dtxyz = t1 + t2 + t3;
      }
      T_out[(size_t)b * n_seg_max + k] = dtxyz;
    }
  }
  return DIRECT_OK;
}

int direct_ref_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---- output sampling (test infrastructure, like everything in this file) -----------------------
 * Literal restatement of the caller's sampling loop (teach_repeat_planner.cpp:1551-1566; the same loop
 * with dt = 0.1 at :1380-1394, :1440-1455, :1493-1508) over Bernstein::getPosFromBezier, getVel and
 * getAcc (global_planner/include/global_planner/utils/bezier_base.h:77-127): pow() per term, the
 * reference's order of the products, sequential traj_len.  acc is divided by T_i to give SI units (the
 * reference's getAcc leaves that to its caller).  Returns the number of points the loop produces, -1
 * when a duration is negative (TRP:1552-1555).  Arrays may be NULL except pos. */
static double bern_C(int n, int k) {
  static const double t5[6] = {1, 5, 10, 10, 5, 1}, t4[5] = {1, 4, 6, 4, 1}, t3[4] = {1, 3, 3, 1};
  return n == 5 ? t5[k] : (n == 4 ? t4[k] : t3[k]);
}
int direct_ref_sample(int n_seg, const double* bez, const double* T, double dt, int capacity, int derivs,
                      int* seg_first, double* pos, double* vel, double* acc, double* length, double* vmax,
                      double* amax) {
  const int order = 5, n1 = 6;
  int count = 0;
  double traj_len = 0.0, pre[3] = {0, 0, 0}, vm = 0.0, am = 0.0;
  for (int i = 0; i < n_seg; i++)
    if (T[i] < 0) return -1;
  for (int i = 0; i < n_seg; i++) {
    const double* c = bez + (size_t)i * 18;
    if (seg_first) seg_first[i] = count;
    for (double t = 0.0; t < 1.0; t += dt / T[i], count += 1) {
      double cur[3], v[3] = {0, 0, 0}, a[3] = {0, 0, 0};
      for (int d = 0; d < 3; d++) {
        double ret = 0.0;
        for (int j = 0; j < n1; j++) ret += bern_C(5, j) * c[d * n1 + j] * pow(t, j) * pow(1 - t, order - j);
        cur[d] = T[i] * ret;
        if (derivs >= 1) {
          double r = 0.0;
          for (int j = 0; j < n1 - 1; j++)
            r += bern_C(4, j) * order * (c[d * n1 + j + 1] - c[d * n1 + j]) * pow(t, j) * pow(1 - t, order - j - 1);
          v[d] = r;
          if (fabs(r) > vm) vm = fabs(r);
        }
        if (derivs >= 2) {
          double r = 0.0;
          for (int j = 0; j < n1 - 2; j++)
            r += bern_C(3, j) * order * (order - 1) * (c[d * n1 + j + 2] - 2 * c[d * n1 + j + 1] + c[d * n1 + j]) *
                 pow(t, j) * pow(1 - t, order - j - 2);
          a[d] = r / T[i];
          if (fabs(a[d]) > am) am = fabs(a[d]);
        }
      }
      if (count < capacity) {
        for (int d = 0; d < 3; d++) {
          pos[(size_t)count * 3 + d] = cur[d];
          if (vel && derivs >= 1) vel[(size_t)count * 3 + d] = v[d];
          if (acc && derivs >= 2) acc[(size_t)count * 3 + d] = a[d];
        }
      }
      if (count) {
        double dx = pre[0] - cur[0], dy = pre[1] - cur[1], dz = pre[2] - cur[2];
        traj_len += sqrt(dx * dx + dy * dy + dz * dz);
      }
      for (int d = 0; d < 3; d++) pre[d] = cur[d];
      if (!(dt / T[i] > 0)) { count += 1; break; } /* the reference would loop forever; the product takes one sample */
    }
  }
  if (length) *length = traj_len;
  if (vmax) *vmax = vm;
  if (amax) *amax = am;
  return count;
}
