/*
 * direct_quad.h -- C-ABI of the BASELINE-LABEL model: batched iLQR for a 12-state / 4-control quadrotor.
 *
 * NO REFERENCE COUNTERPART.  BASELINE.json's metric reads "DDP iterations/sec over batch (12-state/4-ctrl quad,
 * N=100)"; ntu-caokun/DIRECT contains no such optimiser (its ddpTrajOptimizer has 9 states / 10 controls, see
 * include/direct_ddp.h).  SURVEY.md 8(d) config 2 asks for this model as a second policy,
 * "reported separately, flagged no reference counterpart".  Nothing in the Teach-Repeat-Replan pipeline calls
 * it; it exists so that the number the metric string literally names can be measured next to the real path.
 *
 * Model (constants default to the reference's simulator plant, simulation/so3_quadrotor_simulator/src/dynamics/
 * Quadrotor.cpp:16-20):
 *   x = [p(3), v(3), euler(phi, theta, psi), omega(3)],  u = [thrust T, torque(3)]
 *   pdot = v;  vdot = (T/m) R(euler) e3 - g e3;  eulerdot = W(euler) omega;  omegadot = J^-1 (tau - omega x J omega)
 *   x+ = x + dt f(x, u)                                       (explicit Euler, dt = 0.05 s, N = 100 knots)
 *   cost = sum_k dt/2 [ (x_k-xg)' Q (x_k-xg) + (u_k-uh)' R (u_k-uh) ] + 1/2 (x_N-xg)' Qf (x_N-xg), uh = (m g,0,0,0)
 * Solver: Gauss-Newton iLQR with the reference's outer-loop conventions where they carry over (regulariser
 * schedule ddp_optimizer.cpp:452-474, retry on LLT failure :297-310, 11 step sizes :666-670); a trial is accepted on
 * a strict cost decrease.  One 64-lane wavefront per trajectory, like the main path.
 * Algorithmic HBM words per knot-iteration: 3 nx + 5 nu + 2 nu nx = 152.
 */
#ifndef DIRECT_QUAD_H_
#define DIRECT_QUAD_H_

#include <stdint.h>

#include "direct_ddp.h" /* direct_status_t, direct_dtype_t, direct_mem_t */

#ifdef __cplusplus
extern "C" {
#endif

#define DIRECT_QUAD_NX 12
#define DIRECT_QUAD_NU 4

typedef struct {
  double mass, gravity, inertia[3], dt;
  double q_pos, q_vel, q_ang, q_rate; /* running state weights (diagonal Q) */
  double r_thrust, r_torque;          /* running control weights (diagonal R) */
  double qf_pos, qf_vel, qf_ang, qf_rate; /* terminal weights */
  double reg_base;                    /* regulariser lam = reg_base^reg - 1 */
  double tol;                         /* exit when J_prev - J <= tol * J_prev (unless fixed_iters) */
  int32_t iter_max;
  int32_t fixed_iters;                /* benchmark: run exactly iter_max iterations */
} direct_quad_params_t;

/* plant constants of the reference's simulator, dt 0.05, Q (1, 0.1, 1, 0.05), R (0.05, 50), Qf (1000, 500, 500, 100),
 * reg_base 4, tol 1e-6, iter_max 50 */
void direct_quad_default_params(direct_quad_params_t* p);

typedef struct direct_quad_handle_s* direct_quad_handle_t;

direct_status_t direct_quad_create(int32_t dtype /* direct_dtype_t: storage type */, int32_t device, int32_t max_batch,
                                   int32_t n_knots, direct_quad_handle_t* out);
direct_status_t direct_quad_destroy(direct_quad_handle_t h);
const char* direct_quad_last_error(void);

/* Whole solves.  x0[batch][12], xg[batch][12] in; cost[batch], iters[batch] (forward passes = DDP iterations),
 * x[batch][n_knots+1][12], u[batch][n_knots][4] out (any may be NULL).  Real arrays in the handle's storage type,
 * all in memory kind `mem`.  Blocks for host memory, enqueues for device memory. */
direct_status_t direct_quad_solve_batch(direct_quad_handle_t h, const direct_quad_params_t* p, int32_t batch, int32_t mem,
                                        const void* x0, const void* xg, void* cost, int32_t* iters, void* x, void* u);

/* stepwise interface (parity tests): begin = initial roll from the hover input; iterate = n trips of the outer loop;
 * get: HOST arrays of Real x[b][N+1][12], u[b][N][4], K[b][N][4][12], kf[b][N][4] and double scalars[b][8] =
 * cost, reg, step, fp_failed, bp_failed, iter, done, fwd_passes (any may be NULL) */
direct_status_t direct_quad_begin(direct_quad_handle_t h, const direct_quad_params_t* p, int32_t batch, int32_t mem,
                                  const void* x0, const void* xg);
direct_status_t direct_quad_iterate(direct_quad_handle_t h, int32_t n_iters);
direct_status_t direct_quad_get(direct_quad_handle_t h, void* x, void* u, void* K, void* kf, double* scalars);
/* The HIP stream (hipStream_t) the handle enqueues its copies, kernels and timing events on; NULL (the default) is
 * the legacy default stream.  Mirrors direct_ddp_set_stream: device-memory arguments are ordered on this stream. */
direct_status_t direct_quad_set_stream(direct_quad_handle_t h, void* hip_stream);
direct_status_t direct_quad_last_kernel_ms(direct_quad_handle_t h, double* ms);

#ifdef __cplusplus
}
#endif
#endif /* DIRECT_QUAD_H_ */
