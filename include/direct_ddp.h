/*
 * direct_ddp.h -- C-ABI of the MI355X-native batched IPDDP trajectory optimiser.
 *
 * This is the drop-in boundary for ONE path of ntu-caokun/DIRECT: the call
 *
 *   int ddpTrajOptimizer::polyCurveGeneration(const FlightCorridor&, ... 21 more args)
 *       global_planner/include/global_planner/ddp_optimizer.h:267-289
 *       global_planner/src/ddp_optimizer.cpp:5-438
 *
 * and the getters that read its result (ddp_optimizer.h:299-340).  The
 * reference solves one corridor per call on the ROS spin thread; this library
 * solves a batch of independent corridors per call, one 64-lane wavefront per
 * trajectory, on a gfx950 device.  Plain pointers and sizes only: no C++,
 * Eigen, ROS or torch types cross this boundary.
 *
 * Data layout ("batch-major": a trajectory's knot records are contiguous):
 *   Real = float (DIRECT_F32) or double (DIRECT_F64), chosen at create time.
 *   x0[b][9], xd[b][9]          start / goal state [pos xyz, vel xyz, acc xyz]
 *                               (ddp_optimizer.cpp:104-121: rows 0 / 1 of pos,vel,acc)
 *   T0[b][k]                    corridor.durations   (data_type.h:192)
 *   n_planes[b][k]              Polytope.planes.size() (data_type.h:130)
 *   planes[b][k][p][4]          (a,b,c,d), outward normal, inside <=> ax+by+cz+d <= 0
 *                               (poly_utils.cpp:42-52); entries p >= n_planes are ignored
 *   seeds[b][k][3]              Polytope.seed_coord (line-init only, may be NULL)
 *   init_bez[b][k][18]          initbezCoeff row [x0..x5,y0..y5,z0..z5], time-scaled
 *                               (ddp_optimizer.cpp:167); ignored when zero_init
 *   init_poly[b][k][18]         optional replacement of init_bez: getPolyCoeff() rows
 *   k runs over 0..n_seg[b]-1; strides use n_seg_max and p_max.
 */
#ifndef DIRECT_DDP_H_
#define DIRECT_DDP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIRECT_DDP_ABI_VERSION 1

#define DIRECT_NX 9   /* dim*sys_order, ddp_optimizer.cpp:37-39 */
#define DIRECT_NU 10  /* dim*sys_order+1 (c3,c4,c5 per axis + T), ddp_optimizer.cpp:126 */
#define DIRECT_P_LIMIT 128 /* largest planes-per-polytope the kernels are built for: what polyhedronGenerator can emit (poly_utils.hpp: 128-plane capacity); 6 P + 55 rows in up to fourteen row slots per lane */

typedef enum {
  DIRECT_OK = 0,
  DIRECT_ERR_INVALID = 1,     /* bad argument (NULL, size, time_power not in {1,2}, ...) */
  DIRECT_ERR_UNSUPPORTED = 2, /* e.g. p_max > DIRECT_P_LIMIT */
  DIRECT_ERR_DEVICE = 3,      /* HIP allocation / launch / copy failure */
  DIRECT_ERR_NO_DEVICE = 4    /* no gfx950 device visible: there is no CPU fallback */
} direct_status_t;

/* Storage type of every array at this boundary and of the solver's arrays in device memory; the arithmetic is double
 * for both.  With DIRECT_F32 the handle keeps the ITERATE itself (states, polynomial coefficients, durations) as
 * hi + lo float pairs internally, so that a float solve follows the double one to its exit;
 * gains, slacks and duals are single floats. */
typedef enum { DIRECT_F32 = 0, DIRECT_F64 = 1 } direct_dtype_t;
typedef enum { DIRECT_MEM_HOST = 0, DIRECT_MEM_DEVICE = 1 } direct_mem_t;

/* Return codes of one solve, exactly the reference's (ddp_optimizer.cpp:33, 322, 360, 375, 393). */
#define DIRECT_RTN_DONE 0        /* ran to iter_max, or optimality, or line-init exit */
#define DIRECT_RTN_FEAS_OPT 1    /* phase 1: feasible + cost stagnation */
#define DIRECT_RTN_FEAS_FOUND 2  /* phase 0: all c < 2e-4 */
#define DIRECT_RTN_NEG_TIME (-3)
#define DIRECT_RTN_BP_STUCK (-4)
/* Extensions (no reference counterpart; both are < 0, so "rtn >= 0" keeps meaning "usable result"): */
#define DIRECT_RTN_INVALID (-100)     /* this row's n_seg / n_planes lie outside [1, n_seg_max] / [1, p_max] */
#define DIRECT_RTN_SCHED_ERROR (-101) /* the launch's ticket scheduler reported an error: results incomplete */

/* By-value scalar arguments of polyCurveGeneration (ddp_optimizer.h:275-289). */
typedef struct {
  double max_vel;     /* max_vel */
  double max_acc;     /* max_acc */
  double w_snap;      /* w_snap (weights the integral of jerk^2, ddp_optimizer.cpp:991-999) */
  double w_terminal;  /* w_terminal */
  double w_time;      /* w_time */
  int32_t iter_max;   /* iter_max */
  int32_t time_power; /* 1 or 2; anything else is UB in the reference -> DIRECT_ERR_INVALID */
  int32_t zero_init;  /* zero_init_flag */
  int32_t line_init;  /* line_init_flag */
  int32_t minvo;      /* minvo_flag */
  int32_t infeas;     /* initial value of `bool& infeas` for every problem (see infeas_in) */
  /* Extensions with no reference counterpart (0 = reference behaviour): */
  int32_t fixed_iters; /* benchmark: disable the optimality / feasibility early exits */
  int32_t exact_dt;    /* use the exact Bezier dc/dT column instead of quirk Q1 */
} direct_ddp_params_t;

typedef struct {
  int32_t batch;
  int32_t n_seg_max;
  int32_t p_max;
  int32_t mem;               /* direct_mem_t: where every pointer below lives */
  const int32_t* n_seg;      /* [batch] */
  const void* x0;            /* [batch][9] Real */
  const void* xd;            /* [batch][9] Real */
  const void* T0;            /* [batch][n_seg_max] Real; NULL: initTimeAllocation (teach_repeat_planner.cpp:583-639) runs on the
                                device, from x0 / xd positions and `seeds` (required then) with params->max_vel / max_acc -
                                direct_time_allocation()'s expressions in double (within 1e-14 of it); for DIRECT_MEM_DEVICE inputs the
                                seeds -> durations -> plan chain then never visits the host */
  const int32_t* n_planes;   /* [batch][n_seg_max] */
  const void* planes;        /* [batch][n_seg_max][p_max][4] Real */
  const void* seeds;         /* [batch][n_seg_max][3] Real or NULL */
  const void* init_bez;      /* [batch][n_seg_max][18] Real or NULL */
  const uint8_t* infeas_in;  /* [batch] or NULL (then params.infeas) */
  /* Extension (no reference counterpart): warm start as monomial coefficients in the getPolyCoeff()
   * layout [c0xyz..c5xyz] per segment; when non-NULL it replaces init_bez.  World-frame Bezier
   * control points are an ill-conditioned hand-off in fp32 (c3..c5 are 5th-order differences of
   * numbers of magnitude |position|/T), so phase chaining in float should use this. */
  const void* init_poly;     /* [batch][n_seg_max][18] Real or NULL */
} direct_ddp_batch_in_t;

/* What the getters of ddp_optimizer.h:299-340 return, per problem.  Any pointer may be NULL. */
typedef struct {
  int32_t mem;               /* direct_mem_t */
  int32_t* rtn;              /* [batch] return value of polyCurveGeneration */
  int32_t* iter_used;        /* [batch] getIterUsed(): loop index at exit (quirk Q11) */
  int32_t* fwd_passes;       /* [batch] forward passes executed = DDP iterations (the metric) */
  uint8_t* infeas_out;       /* [batch] `bool& infeas` after the call */
  uint8_t* line_failed_out;  /* [batch] `bool& line_failed` after the call */
  void* cost;                /* [batch] getDDPObjective() */
  void* costq;               /* [batch] running cost without the terminal term */
  void* jerk_cost;           /* [batch] getJerkCost() */
  void* terminal_norm2;      /* [batch] getTerminalNorm() */
  void* opterr;              /* [batch] bp.opterr at exit */
  void* mu;                  /* [batch] alg.mu at exit */
  void* bez;                 /* [batch][n_seg_max][18] getBezCoeff() layout */
  void* poly;                /* [batch][n_seg_max][18] getPolyCoeff(): [c0xyz..c5xyz] */
  void* T;                   /* [batch][n_seg_max] getPolyTime() */
} direct_ddp_batch_out_t;

typedef struct {
  int32_t dtype;         /* direct_dtype_t */
  int32_t device;        /* HIP device ordinal */
  int32_t max_batch;
  int32_t n_seg_max;
  int32_t p_max;
  int32_t reserved;      /* flags: DIRECT_FLAG_* (0 = defaults) */
} direct_ddp_config_t;

/* Launch k_iterate as one workgroup per trajectory instead of the default ticket scheduler (persistent
 * waves drawing (trajectory, iteration) tickets: no tail when batch > resident waves).  Results are
 * identical; the environment variable DIRECT_DDP_SCHED=static|dynamic overrides the flag.
 * Two more scheduling choices are made per launch from the batch size and never change a result bit (both have a
 * bitwise-equality test): two line-search steps per forward sweep (batches up to 2 x the resident waves;
 * DIRECT_DDP_PAIR=0|1 forces) and the shared line search, in which waves waiting for a trajectory evaluate later
 * steps of its line search (batches up to 1.5 x the resident waves and n_seg_max >= 80 - shorter trajectories do not
 * pay for the hand-over unless the batch is at most an eighth of the resident waves -, handles of at most 2 x;
 * DIRECT_DDP_HELP=0|1 forces, read at create time: such a handle keeps 12 iterate buffers instead of 3).  With the shared
 * line search the ticket scheduler also serves batches below the resident waves, where the waves left over become helpers
 * (single-trajectory latency: -20 %).  A third choice of the same kind: for batches up to an eighth of the resident waves
 * (row-slot classes of up to 33 planes) the waiting waves also compute the value-independent half of the knots of the
 * trajectory's BACKWARD sweep and hand it over through records in HBM (DIRECT_DDP_BSHARE=0 off, 1 for every batch with a
 * shared line search, 2 the same path forced without helpers - the tests' bitwise check; read at create time).  A launch of a
 * fixed number of iterations with a shared line search also hands out DIRECT_DDP_TAIL (default 8) rounds of help-only
 * tickets behind its last epoch: waves that would otherwise leave the kernel wait for a trajectory's last chunk and join
 * its open line searches - the slowest chains end the launch (0: none; bitwise-equality test as for the others). */
#define DIRECT_FLAG_STATIC_SCHEDULE 1
/* Pipelined batches (the reference's contract: an independent optimiser object per call, teach_repeat_planner.cpp:853-854):
 * create TWO handles with this flag, give each a stream of its own (direct_ddp_set_stream) and alternate device-memory
 * direct_ddp_solve_batch / direct_ddp_plan_batch calls between them - each call returns once its kernels are enqueued.
 * The hot kernel is a grid of persistent waves that normally stay until the launch's last chunk (parked on tickets of epochs
 * to come, or as helpers); with the flag the waves a launch no longer needs (more than one per unfinished trajectory, 64 at
 * least) leave as soon as they would draw their next ticket and no help-only tickets are handed out, so that the OTHER handle's next
 * launch fills the CUs this launch's slowest chains leave idle.  Results are bit-identical to serial launches
 * (tests/test_gpu_fullsize.py); DIRECT_DDP_YIELD=k sets the waves kept per unfinished trajectory (0: off). */
#define DIRECT_FLAG_YIELD 2

typedef struct direct_ddp_handle_s* direct_ddp_handle_t;

int32_t direct_ddp_abi_version(void);
const char* direct_ddp_last_error(void);

direct_status_t direct_ddp_create(const direct_ddp_config_t* cfg, direct_ddp_handle_t* out);
direct_status_t direct_ddp_destroy(direct_ddp_handle_t h);

/* HIP stream (hipStream_t cast to void*) the kernels are launched on; NULL = default stream. */
direct_status_t direct_ddp_set_stream(direct_ddp_handle_t h, void* hip_stream);

/* One polyCurveGeneration per problem (replaces ddp_optimizer.cpp:5-438).  Blocks until
 * the results are in `out` when out->mem is host; with device memory the call returns
 * after enqueueing on the handle's stream.
 *
 * Validation.  Scalars, pointers-not-NULL and the batch / n_seg_max / p_max fields are checked on the host
 * for both memory kinds.  The CONTENTS of n_seg and n_planes are checked on the host for DIRECT_MEM_HOST
 * (-> DIRECT_ERR_INVALID, nothing is launched) and ON THE DEVICE for DIRECT_MEM_DEVICE: a row whose sizes
 * are out of range is not solved and reports rtn = DIRECT_RTN_INVALID (its other outputs are zero).
 * Nothing else about device arrays (extents, finiteness of the reals) can be or is validated.
 *
 * Scheduler errors.  The ticket scheduler's spin limit (a scheduling bug, never expected) sets a sticky
 * per-handle flag: it is cleared when a solve / plan / begin call starts, makes every host-memory finish
 * return DIRECT_ERR_DEVICE, makes the finish kernel write rtn = DIRECT_RTN_SCHED_ERROR for every row of a
 * device-memory result, and can be polled with direct_ddp_sched_error(). */
direct_status_t direct_ddp_solve_batch(direct_ddp_handle_t h, const direct_ddp_params_t* params,
                                       const direct_ddp_batch_in_t* in, direct_ddp_batch_out_t* out);

/* Synchronises the handle's stream and reads the sticky scheduler-error flag (see above). */
direct_status_t direct_ddp_sched_error(direct_ddp_handle_t h, int32_t* flag);
/* What the FIRST wait that ran into the spin limit saw, since the handle was created (diagnostics; all zero when none did):
 * out64[0..11] = { ticket, epoch it waited for, trajectory, done_epoch[trajectory], ticket counter, waves inside, unfinished
 * trajectories, 1, milliseconds waited, batch, tickets of the launch, spins }; [12..63]: development builds
 * (-DDDP_SCHED_DEBUG) only: waves waiting / inside a chunk, a histogram of where the launch's waves stand. */
direct_status_t direct_ddp_sched_debug(direct_ddp_handle_t h, int32_t* out64);

/* fastTrajPlanning's protocol (teach_repeat_planner.cpp:886-921) for a batch: phase 0
 * (params0: zero init, infeasible start), UpdateTime where rtn0 == 2, phase 1 (params1) from
 * the phase-0 Bezier coefficients: where phase 0 did not return 2 the reference converts them back
 * with the CALLER's durations (ddp_optimizer.cpp:167-193 after 799-812), i.e. the warm start is the
 * phase-0 curve in normalised time - reproduced on the device (coefficient c_i scaled by
 * (T_0 / T_1)^(i-1), handed over as monomial coefficients).  out0 may be NULL; when the call fails
 * in phase 1 the contents of out0 are undefined. */
direct_status_t direct_ddp_plan_batch(direct_ddp_handle_t h, const direct_ddp_params_t* params0,
                                      const direct_ddp_params_t* params1,
                                      const direct_ddp_batch_in_t* in, direct_ddp_batch_out_t* out0,
                                      direct_ddp_batch_out_t* out1);

/* initTimeAllocation (teach_repeat_planner.cpp:583-639): trapezoid-profile duration per
 * segment from start, seeds[1..n-1] and goal.  Host pointers, double precision.  (The device-side twin runs inside
 * solve / plan / begin when direct_ddp_batch_in_t.T0 is NULL.) */
direct_status_t direct_time_allocation(int32_t batch, int32_t n_seg_max, const int32_t* n_seg,
                                       const double* start, const double* goal,
                                       const double* seeds, double max_vel, double max_acc,
                                       double* T_out);

/* ---- corridor wire format and replay (the step right before the path) ------------------
 * msgs/corridor (msgs/msg/corridor.msg, polyhedron.msg, facet3.msg) in ROS 1 serialisation, i.e. what the
 * reference's recorder publishes and its replay reads (writeCorridorMsg / readCorridorMsg,
 * teach_repeat_planner.cpp:354-410): little endian,
 *   int32 path_id; uint32 n; n x { float64 center[3]; float64 seed_coord[3]; uint32 m; m x float64 (a,b,c,d) }.
 * Host arrays, double precision; planes[k][p_max][4], seeds[k][3], centers[k][3]. */
size_t direct_corridor_wire_size(int32_t n_seg, const int32_t* n_planes);
/* writes direct_corridor_wire_size() bytes to buf */
direct_status_t direct_corridor_pack(int32_t path_id, int32_t n_seg, const int32_t* n_planes, const double* planes,
                                     int32_t p_max, const double* seeds, const double* centers, uint8_t* buf,
                                     size_t capacity, size_t* written);
/* DIRECT_ERR_INVALID: truncated / malformed buffer; DIRECT_ERR_UNSUPPORTED: more polytopes than n_seg_max
 * or more facets than p_max.  *used = bytes consumed (messages may be concatenated in one buffer). */
direct_status_t direct_corridor_unpack(const uint8_t* buf, size_t len, int32_t n_seg_max, int32_t p_max,
                                       int32_t* path_id, int32_t* n_seg, int32_t* n_planes, double* planes,
                                       double* seeds, double* centers, size_t* used);
/* The replay protocol of corridorRecCallBack + fastTrajPlanning (teach_repeat_planner.cpp:308-352, 796-823):
 * problem b = the first n_first + b polytopes of the recorded corridor, start = center of polytope 0, goal =
 * center of the last one, at rest, durations from initTimeAllocation.  Fills the arrays of a
 * direct_ddp_batch_in_t with n_seg_max = n_first + batch - 1 rows per problem (double precision, host):
 * n_seg[batch], x0/xd[batch][9], T0[batch][n_seg_max], n_planes[batch][n_seg_max],
 * planes[batch][n_seg_max][p_max][4], seeds_out[batch][n_seg_max][3]. */
direct_status_t direct_corridor_replay_batch(int32_t n_rec, const int32_t* n_planes_rec, const double* planes_rec,
                                             int32_t p_max, const double* seeds_rec, const double* centers_rec,
                                             int32_t n_first, int32_t batch, double max_vel, double max_acc,
                                             int32_t* n_seg, double* x0, double* xd, double* T0, int32_t* n_planes,
                                             double* planes, double* seeds_out);

/* ---- output sampling (the step right after the path) ----------------------------------
 * Batched form of the sampling loops of the caller's visualisation / audit helpers
 * (teach_repeat_planner.cpp:1380-1394, 1440-1455, 1493-1508, 1551-1566) over
 * Bernstein::getPosFromBezier / getVel / getAcc (utils/bezier_base.h:77-127):
 *   for every segment i:  for (double t = 0.0; t < 1.0; t += dt / T_i)
 *     pos = T_i * getPosFromBezier(bez, t, i);  vel = getVel(bez, i, t);  acc = getAcc(bez, i, t) / T_i
 *     traj_len += |pos - previous pos|
 * The sample times reproduce the reference's accumulation of t exactly, so count is the reference's.
 * Real = the handle's dtype; arithmetic is double.  All arrays of `in` and `out` live where `mem` says. */
typedef struct {
  int32_t batch, n_seg_max;
  int32_t capacity;          /* points per trajectory the output arrays hold */
  int32_t derivs;            /* 0: positions; 1: + velocities; 2: + accelerations */
  int32_t mem;               /* direct_mem_t */
  const int32_t* n_seg;      /* [batch] */
  const void* bez;           /* [batch][n_seg_max][18] getBezCoeff() layout (time-scaled control points) */
  const void* T;             /* [batch][n_seg_max] getPolyTime() */
  double dt;                 /* sample period [s] (> 0): 0.1 and 0.2 in the reference's helpers */
  /* optional corridor for the containment audit (out->cmax); all three NULL / 0 when not wanted */
  int32_t p_max;
  const int32_t* n_planes;   /* [batch][n_seg_max] */
  const void* planes;        /* [batch][n_seg_max][p_max][4] as in direct_ddp_batch_in_t */
} direct_sample_in_t;

typedef struct {
  int32_t* count;            /* [batch] points the loop produces (only the first `capacity` are stored);
                                -1 when a duration is negative (the reference returns, TRP:1552-1555) */
  int32_t* seg_first;        /* [batch][n_seg_max] index of each segment's first point, or NULL */
  void* pos;                 /* [batch][capacity][3] */
  void* vel;                 /* [batch][capacity][3] or NULL */
  void* acc;                 /* [batch][capacity][3] or NULL */
  void* length;              /* [batch] traj_len, or NULL */
  void* vmax;                /* [batch] max over samples and axes of |vel| (derivs >= 1), or NULL */
  void* amax;                /* [batch] max over samples and axes of |acc| (derivs >= 2), or NULL */
  void* cmax;                /* [batch] max over samples of max_p (a x + b y + c z + d) against the planes of the
                                sample's own segment: <= 0 iff every sample lies in its polytope; needs in->planes */
} direct_sample_out_t;

direct_status_t direct_traj_sample_batch(direct_ddp_handle_t h, const direct_sample_in_t* in,
                                         direct_sample_out_t* out);
/* HIP-event time of the last direct_traj_sample_batch kernel on the handle's stream [ms] */
direct_status_t direct_traj_sample_last_ms(direct_ddp_handle_t h, float* ms);

/* ---- stepwise interface (per-pass parity tests and profiling) ------------------------ */
/* begin: setup + initialroll + mu/filter/reg reset (ddp_optimizer.cpp:42-286). */
direct_status_t direct_ddp_begin(direct_ddp_handle_t h, const direct_ddp_params_t* params,
                                 const direct_ddp_batch_in_t* in);
/* one backwardpass() (ddp_optimizer.cpp:440-644), no retry loop */
direct_status_t direct_ddp_backward_pass(direct_ddp_handle_t h);
/* one forwardpass() (ddp_optimizer.cpp:647-778).  After a direct_ddp_backward_pass that failed (LLT, ddp_optimizer.cpp:546-551,
 * 595-600) it runs as the reference's does: with the gains the knots below the failure still hold from the last
 * completed backward pass (ddp_optimizer.cpp:568-572, 611-614, 630-631 were not reached for them). */
direct_status_t direct_ddp_forward_pass(direct_ddp_handle_t h);
/* The same forwardpass() evaluated in the stored-gain form whatever the state of the backward pass (the form the solver
 * itself only uses after ddp_optimizer.cpp:297-310 gave up, rtn = -4): with every gain current it must agree with
 * direct_ddp_forward_pass to rounding - a test hook, no reference counterpart of its own. */
direct_status_t direct_ddp_forward_pass_stored(direct_ddp_handle_t h);
/* n trips of the outer loop (ddp_optimizer.cpp:295-412) for every unfinished problem */
direct_status_t direct_ddp_iterate(direct_ddp_handle_t h, int32_t n_iters);
/* finalroll + conversions (ddp_optimizer.cpp:414-437) */
direct_status_t direct_ddp_finish(direct_ddp_handle_t h, direct_ddp_batch_out_t* out);

typedef enum {
  DIRECT_FIELD_X = 0,      /* [b][n_seg_max+1][9] */
  DIRECT_FIELD_U = 1,      /* [b][n_seg_max][10] */
  DIRECT_FIELD_S = 2,      /* [b][n_seg_max][nc_max] */
  DIRECT_FIELD_Y = 3,      /* [b][n_seg_max][nc_max]; the dual iterate exists in infeasible mode only (in feasible
                              mode the array is scratch: s/c of the nominal iterate for the line search) */
  DIRECT_FIELD_C = 4,      /* [b][n_seg_max][nc_max] (recomputed from x,u on read) */
  DIRECT_FIELD_KU = 5,     /* [b][n_seg_max][10] */
  DIRECT_FIELD_KUU = 6,    /* [b][n_seg_max][10][9] */
  DIRECT_FIELD_KS = 7,     /* [b][n_seg_max][nc_max]; the slack / dual gains ks, ky (ddp_optimizer.cpp:568-571, 611) are formed ONLY by
                              direct_ddp_backward_pass: valid after that call, not after direct_ddp_iterate / solve_batch / plan_batch,
                              whose forward passes step s and y without them */
  DIRECT_FIELD_KY = 8,     /* [b][n_seg_max][nc_max]; infeasible mode only; see DIRECT_FIELD_KS */
  DIRECT_FIELD_SCALARS = 9 /* [b][16]: cost,costq,logcost,err,mu,reg,opterr,stepsize,
                              step,fp_failed,bp_failed,rtn,iter,done,filter_n,infeas */
} direct_field_t;
/* nc_max = 6*p_max + 55.  dst/src are HOST buffers of Real. */
direct_status_t direct_ddp_get_field(direct_ddp_handle_t h, int32_t field, void* dst);
direct_status_t direct_ddp_set_field(direct_ddp_handle_t h, int32_t field, const void* src);

/* Device time of the last solve/iterate kernel(s), measured with HIP events on the handle's
 * stream (milliseconds); *n_launches receives how many kernel launches that covered. */
direct_status_t direct_ddp_last_kernel_ms(direct_ddp_handle_t h, double* ms, int32_t* n_launches);

/* How the last direct_ddp_iterate / solve / plan launch of the hot kernel was scheduled, and how much sweep work it
 * executed (observability only - no reference counterpart; results never depend on any of it).  Blocks until the
 * launch has finished. */
typedef struct {
  int32_t dynamic;        /* 1: ticket-scheduled persistent waves (k_iterate_dyn); 0: one workgroup per trajectory */
  int32_t shared_search;  /* 1: waiting waves evaluate later step sizes of the trajectory they wait for */
  int32_t pair_trials;    /* 1: two line-search steps per forward sweep from the second attempt on */
  int32_t single_steps;   /* 1: shared searches hand out single steps (few trajectories on many waves) */
  int32_t n_buffers;      /* iterate buffers allocated per array (3, or 12 with the shared line search) */
  int32_t resident_waves; /* persistent waves launched / one-wave workgroups resident at once on this device */
  int32_t batch;
  int32_t shared_sweep;   /* 1: waiting waves compute the value-independent half of backward knots of the trajectory they wait
                             for (hand-over records in HBM; results bitwise those of the owner-only sweep); 2: the same path
                             forced without helpers (DIRECT_DDP_BSHARE=2, tests); 0: every sweep stays with its owner */
  uint64_t bwd_knot_visits; /* backward-sweep knots executed by the launch (all trajectories, retries included) */
  uint64_t fwd_knot_visits; /* forward trial-knots executed (every trial of every line search, helpers' included;
                               a trial cut short by the fraction-to-boundary rule counts the knots it reached) */
} direct_ddp_launch_info_t;
direct_status_t direct_ddp_last_launch_info(direct_ddp_handle_t h, direct_ddp_launch_info_t* info);
/* Work counters of the last hot-kernel launch (observability only): out4[0] backward-sweep knots, out4[1] forward
 * trial-knots (as in direct_ddp_launch_info_t), out4[2] backward knots whose value-independent half a helper wave (or the
 * forced split) computed and handed over through HBM, out4[3] line searches that accepted a step (out of the forward
 * passes the launch ran: the rest ended with fp_failed, ddp_optimizer.cpp:760-762). */
direct_status_t direct_ddp_last_counters(direct_ddp_handle_t h, uint64_t* out4);

/* Config-5 reduction: index and value of the smallest cost among problems with rtn >= 0.
 * cost/rtn are device or host arrays per `mem`; the result is written to host. */
direct_status_t direct_ddp_best_cost(direct_ddp_handle_t h, int32_t mem, const void* cost,
                                     const int32_t* rtn, int32_t batch, int32_t* best_index,
                                     double* best_cost);

/* ---- config-5 reduction across the GPUs of one node, for a C / C++ host ---------------------------
 * No reference counterpart (the reference is one process, no collectives); specified by BASELINE.json
 * configs[4] and SURVEY.md 8(e): one process (or thread) per GPU, each with its own handle, solves its shard
 * [first_index, first_index + batch) with no exchange; then ONE ncclAllGather of (cost, global index) and ONE
 * of every rank's local-best block put the cheapest trajectory with rtn >= 0 on every rank.  RCCL is resolved
 * at run time (librccl.so.1); DIRECT_ERR_UNSUPPORTED when it cannot be found.
 *
 *   direct_rccl_id_t id;  if (rank == 0) direct_rccl_unique_id(&id);   // ship the 128 bytes to the other ranks
 *   void* comm;  direct_rccl_comm_create(h, &id, n_ranks, rank, &comm); // ncclCommInitRank on h's device
 *   direct_ddp_solve_batch(h, &params, &in, &out);                      // out->mem may be device
 *   direct_ddp_gather_best(h, comm, n_ranks, rank, out.mem, out.cost, out.rtn, out.bez, out.T, batch, first, ...);
 * `comm` is an ncclComm_t; a communicator the host created itself with ncclCommInitRank is equally valid. */
typedef struct { char internal[128]; } direct_rccl_id_t; /* = ncclUniqueId */
direct_status_t direct_rccl_unique_id(direct_rccl_id_t* id);
direct_status_t direct_rccl_comm_create(direct_ddp_handle_t h, const direct_rccl_id_t* id, int32_t n_ranks,
                                        int32_t rank, void** comm);
direct_status_t direct_rccl_comm_destroy(void* comm);
/* cost[batch], rtn[batch], bez[batch][n_seg_max][18], T[batch][n_seg_max] as written by solve/plan (memory kind
 * `mem`).  Results: best_index (global, -1 if no rank has a feasible trajectory), best_cost, owner_rank on the
 * host; best_bez[n_seg_max][18], best_T[n_seg_max] (may be NULL) in memory kind `mem`.  Blocks until done. */
direct_status_t direct_ddp_gather_best(direct_ddp_handle_t h, void* nccl_comm, int32_t n_ranks, int32_t rank,
                                       int32_t mem, const void* cost, const int32_t* rtn, const void* bez,
                                       const void* T, int32_t batch, int64_t first_index, int64_t* best_index,
                                       double* best_cost, int32_t* owner_rank, void* best_bez, void* best_T);

#ifdef __cplusplus
}
#endif
#endif /* DIRECT_DDP_H_ */
