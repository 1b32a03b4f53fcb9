// libdirect_ddp.so: gfx950 kernels + the C-ABI of include/direct_ddp.h.
//
// Replaces ddpTrajOptimizer::polyCurveGeneration (global_planner/src/ddp_optimizer.cpp:5-438) for a
// batch of independent corridors.  One workgroup = one 64-lane wavefront = one trajectory
// (ddp_wave.h); the host side below only moves data, launches and times.  There is no CPU path:
// without a gfx950 device direct_ddp_create() fails with DIRECT_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/direct_ddp.h"
#include "ddp_wave.h"
#include "traj_sample.h"
#include "corridor_io.h"
#include "rccl_gather.h"

using namespace direct;

// ---- kernels ---------------------------------------------------------------------------------------
#if defined(DDP_TIMING)
#undef DDP_TICK_OBJ
#define DDP_TICK_OBJ W  // the marks of the kernels' own code (ddp_wave.h: those inside Wave use *this)
#endif
#ifndef DDP_WAVES_F32
#define DDP_WAVES_F32 1
#endif
#ifndef DDP_WAVES_F64
#define DDP_WAVES_F64 1
#endif
// Arithmetic is double for both storage types (DESIGN.md "Precision"): St = float halves the HBM
// traffic, it does not change the arithmetic.
typedef double Cmp;
// Waves per SIMD of the wide row-slot classes.  Classes 5..7 (polytopes of 34..65 planes, where the real pipeline's
// corridors sit: widest polytope per corridor median 32, p90 42) take 258..290 registers when left alone; capped at 256
// they spill 32..240 bytes and run TWO waves per SIMD: 5680 real-corridor plans 39.1 -> 30.8 ms (f32), 38.6 -> 31.7 ms
// (f64), bit-identical (profiles/r05_real_corridors.json).  Classes 8..14: DDP_WAVES_WIDER.
#ifndef DDP_WAVES_WIDE
#define DDP_WAVES_WIDE 2
#endif
#ifndef DDP_WAVES_WIDER
#define DDP_WAVES_WIDER 1
#endif
template <typename St>
struct MinWaves { static constexpr int v = DDP_WAVES_F32; };
template <>
struct MinWaves<double> { static constexpr int v = DDP_WAVES_F64; };

// trajectory of the i-th workgroup / ticket of a launch (Batch::idx: the launch's row-slot class)
template <typename St>
__device__ __forceinline__ int traj_of(const Batch<St>& B, int i) {
  return B.idx ? __builtin_amdgcn_readfirstlane(B.idx[i]) : i;
}

template <typename St, int RPL>
__global__ __launch_bounds__(64) void k_begin(Batch<St> B) {
  __shared__ WaveLds<Cmp, St, RPL> lds;
  Wave<Cmp, St, RPL> W(B, lds, traj_of(B, blockIdx.x));
  W.init_tables();
  W.begin();
  W.store_state();
}

// the hot kernel: n trips of the outer loop (ddp_optimizer.cpp:295-412) per trajectory
template <typename St, int RPL>
__global__ __launch_bounds__(64, (RPL > 7 ? DDP_WAVES_WIDER : (RPL > 4 ? DDP_WAVES_WIDE : MinWaves<St>::v))) void k_iterate(Batch<St> B, int n_iters) {
  __shared__ WaveLds<Cmp, St, RPL> lds;
  Wave<Cmp, St, RPL> W(B, lds, traj_of(B, blockIdx.x));
  W.load_state();
  if (__builtin_amdgcn_readfirstlane(lds.st.done)) return;
  W.init_tables();
  W.iterate(n_iters);
  W.store_state();
}

// The same loop, dynamically scheduled.  A launch of `batch` one-wave workgroups over the 256 x 12
// resident slots leaves a tail: the last batch - 3072 trajectories run alone on an almost empty chip.
// Here a resident set of persistent waves draws TICKETS instead: ticket t = (epoch e, trajectory b) =
// (t / batch, t % batch) is `chunk` trips of the outer loop of trajectory b.  Between iterations a
// trajectory lives entirely in HBM (iterate buffers, TrajState, filter), so consecutive chunks of one
// trajectory may run on different CUs / XCDs; chunk e waits for chunk e-1 through done_epoch[b]
// (agent-scope release / acquire).  The holder of an earlier ticket is always resident and running, so
// the wait cannot deadlock; a spin limit turns a scheduling bug into an error code instead of a hang.
constexpr int kDoneBit = 1 << 30;  // in done_epoch[b]: the trajectory has left the outer loop
#ifndef DDP_POLL_SLEEP
#define DDP_POLL_SLEEP 32  // s_sleep units (64 clocks) between two polls of a waiting wave (next_work)
#endif
struct Sched {
  unsigned* ticket;   // [1] next ticket
  int* done_epoch;    // [batch] chunks completed per trajectory
  int* err;           // [1] set to 1 when the spin limit is hit
  int chunk;
  int prio;           // raise the wave priority of chunks that had to wait for their predecessor
  int tail;           // rounds of help-only tickets behind the last epoch (next_work)
  // Yielding (DIRECT_FLAG_YIELD: another handle's launch is queued behind this one): a wave that is about to draw a ticket
  // leaves the kernel instead when more than max(yield_min, yield_k x unfinished trajectories) waves are still inside - a
  // trajectory needs ONE wave to advance and at most eleven more to help with its line search, the rest only park in
  // next_work() on tickets of epochs to come and keep the other launch's workgroups off the CUs.  Scheduling only: which
  // wave runs a ticket never shows in the results (the holder of the earliest unfinished ticket is always among those
  // that stay).  0: off.
  int* dbg;           // [64] what a wait that ran into the spin limit saw (direct_ddp_sched_debug)
#if defined(DDP_SCHED_DEBUG)
  int* mark;          // development builds: [2 x grid] Batch::mark
  int grid;
#endif
  int* waves;         // [1] waves of this launch still inside the kernel
  int* alive;         // [1] trajectories still in their outer loop (Batch::live when the line search is shared)
  int yield_k, yield_min;
};
// The next piece of work for a persistent wave.  `held` < 0: draws the next ticket (epoch, trajectory) = (t / batch,
// t % batch); else continues to wait with ticket `held` in hand.  Waits for the previous chunk of the ticket's trajectory.
// Returns the ticket, with *help = 0 when its chunk can run now, or *help = 1 when the predecessor is still running and
// has opened its line search to helpers (ddp_wave.h, fwd_pass): the caller runs a round of it and comes back with the
// ticket; *help = 2 likewise when the predecessor's backward sweep is open and has unclaimed knots (Wave::iterate_once, bwd_front_run).  -1: no tickets left; -2: the ticket's trajectory has already finished (kDoneBit in done_epoch); -3 - b: the
// wait for trajectory b timed out (the chunk is then skipped and b marked finished).  Out of line and free of early
// exits on purpose: inlined into the (huge) iterate loop the structuriser turned the nested uniform loops into
// exec-masked ones.
// Tickets total .. total_help - 1 are HELP-ONLY: `tail` more rounds over the trajectories behind the last epoch.  A wave that
// draws one waits for the trajectory's last chunk like any waiter - and so joins its open line searches (and shared
// sweeps) - but never runs a chunk: -2 once that chunk is done.  Without them the waves leave a fixed-length launch as soon
// as the tickets run out, and the last iterations of the slowest chains - the launch's tail - search alone on an emptying chip.
__device__ __attribute__((noinline)) int next_work(Sched S, HelpSlot* slots, BwdShare* sweeps, const int32_t* idx, unsigned nb, unsigned total,
                                                   unsigned total_help, int held, int* waited, int* help) {
  unsigned t = (unsigned)held;
  if (held < 0) {
    unsigned tv = 0;
    if (threadIdx.x == 0) tv = __hip_atomic_fetch_add(S.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = (unsigned)__builtin_amdgcn_readfirstlane((int)tv);
    *waited = 0;
  }
  *help = 0;
  if (t >= total_help) return -1;
  const int et = (int)(t / nb), bi = (int)(t - (unsigned)et * nb);
  const int n_ep = (int)(total / nb);
  const int e = et < n_ep ? et : n_ep;  // what the ticket waits for
  const int b = idx ? __builtin_amdgcn_readfirstlane(idx[bi]) : bi;  // done_epoch / slots are indexed by the trajectory's own number
  int ready = (e == 0) ? 1 : 0, have = 0, wanted = 0;
  int spins = 0;
  const unsigned long long wait_t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz (diagnostics of a timeout only)
#if defined(DDP_SCHED_DEBUG)  // development builds: how many waves wait / run, and a snapshot a quarter of a second into a long wait
  if (S.mark != nullptr && threadIdx.x == 0) S.mark[blockIdx.x] = 60;
  if (!ready && threadIdx.x == 0) __hip_atomic_fetch_add(&S.dbg[12], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int was_waiting = !ready;
#endif
  for (; !ready && !wanted && spins < kSpinLimit; spins++) {
#if defined(DDP_SCHED_DEBUG)
    if (spins == (1 << 18) && threadIdx.x == 0 && __hip_atomic_exchange(&S.dbg[23], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      S.dbg[16] = (int)t; S.dbg[17] = e; S.dbg[18] = S.done_epoch[b];
      S.dbg[19] = (int)__hip_atomic_load(S.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      S.dbg[20] = __hip_atomic_load(&S.dbg[12], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      S.dbg[21] = __hip_atomic_load(&S.dbg[13], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      S.dbg[22] = (int)((__builtin_amdgcn_s_memrealtime() - wait_t0) / 100000ull);
    }
#endif
    int hv = 0, g = 0, r = 0, lr = 0, sw = 0;
    if (threadIdx.x == 0) {
      hv = __hip_atomic_load(&S.done_epoch[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (slots) {
        g = __hip_atomic_load(&slots[b].gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r = __hip_atomic_load(&slots[b].next_round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lr = __hip_atomic_load(&slots[b].last_round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (sweeps) {  // the predecessor's backward sweep is open and has knots nobody has claimed yet
        const unsigned w = __hip_atomic_load(&sweeps[b].word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long c = __hip_atomic_load(&sweeps[b].claim, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sw = ((w >> kBsCountBits) != 0 && (int)(w & kBsCountMask) < kBsMaxHelpers &&
              (long long)(c & 0xffffffffull) < (long long)(c >> 32) - (long long)kClaimBias) ? 1 : 0;
      }
    }
    have = __builtin_amdgcn_readfirstlane(hv);
    ready = (have >= e) ? 1 : 0;
    wanted = (!ready && __builtin_amdgcn_readfirstlane((g != 0 && r <= lr) ? 1 : 0)) ? 1 : 0;
    if (!ready && !wanted && __builtin_amdgcn_readfirstlane(sw)) wanted = 2;
    if (!ready && !wanted) __builtin_amdgcn_s_sleep(DDP_POLL_SLEEP);
  }
#if defined(DDP_SCHED_DEBUG)
  if (was_waiting && threadIdx.x == 0) __hip_atomic_fetch_add(&S.dbg[12], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  if (spins > 1 || wanted) *waited = 1;
  if (wanted) {
    *help = wanted;
    return (int)t;
  }
  if (!ready) {  // a scheduling bug: never a hang, and never a chunk run on a trajectory whose previous chunk
                 // may still be in flight elsewhere.  The flag is sticky until the next solve (direct_ddp.h).
    if (threadIdx.x == 0) {
      __hip_atomic_store(S.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (S.dbg != nullptr && __hip_atomic_exchange(&S.dbg[7], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {  // the first one only
        S.dbg[0] = (int)t; S.dbg[1] = e; S.dbg[2] = b; S.dbg[3] = have;
        S.dbg[4] = (int)__hip_atomic_load(S.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        S.dbg[5] = S.waves ? __hip_atomic_load(S.waves, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
        S.dbg[6] = S.alive ? __hip_atomic_load(S.alive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
        S.dbg[8] = (int)((__builtin_amdgcn_s_memrealtime() - wait_t0) / 100000ull);  // the wait in milliseconds
        S.dbg[9] = (int)nb; S.dbg[10] = (int)total; S.dbg[11] = spins;
#if defined(DDP_SCHED_DEBUG)
        if (S.mark != nullptr) {  // where the launch's waves are: a histogram of their marks, and three of the waves inside a chunk
          int n_in = 0;
          for (int q = 32; q < 64; q++) S.dbg[q] = 0;
          for (int w = 0; w < S.grid; w++) {
            const int m = __hip_atomic_load(&S.mark[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int bucket = m == 0 ? 32 : m == 5 ? 33 : m == 6 ? 34 : m == 11 ? 35 : m == 12 ? 36 : m == 20 ? 37 : m == 30 ? 38 : m == 31 ? 39 : m == 40 ? 40
                               : m == 60 ? 41 : (m >= 100 && m < 200) ? 42 + (m - 100 < 11 ? m - 100 : 11) : (m >= 200 ? 54 : 55);
            S.dbg[bucket]++;
            if (m != 0 && m != 40 && m != 60 && n_in < 3) { S.dbg[56 + 2 * n_in] = w; S.dbg[57 + 2 * n_in] = S.mark[S.grid + w]; n_in++; }
          }
        }
        S.dbg[28] = __hip_atomic_load(&S.dbg[26], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // publish sections entered / left at the timeout
        S.dbg[29] = __hip_atomic_load(&S.dbg[27], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        S.dbg[14] = __hip_atomic_load(&S.dbg[12], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // waves waiting / running chunks at the timeout
        S.dbg[15] = __hip_atomic_load(&S.dbg[13], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
      }
    }
    return -3 - b;
  }
  if (have >= kDoneBit || t >= total) return -2;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return (int)t;
}
// SHARE: the instantiation whose backward sweeps can be shared with waiting waves (Wave<.., SHARE>; row-slot classes 2 .. 4)
template <typename St, int RPL, bool SHARE = false>
__global__ __launch_bounds__(64, (RPL > 7 ? DDP_WAVES_WIDER : (RPL > 4 ? DDP_WAVES_WIDE : MinWaves<St>::v))) void k_iterate_dyn(Batch<St> B, int n_iters, Sched S) {
  __shared__ WaveLds<Cmp, St, RPL> lds;
  Wave<Cmp, St, RPL, SHARE> W(B, lds, 0);
  W.init_tables();
  const unsigned nb = (unsigned)B.B;
  const unsigned n_epochs = (unsigned)((n_iters + S.chunk - 1) / S.chunk);
  const unsigned total = nb * n_epochs;
  const unsigned total_help = B.help ? total + nb * (unsigned)S.tail : total;
  DDP_MARK("Z_0");
  int held = -1, waited = 0;
#pragma unroll 1
  for (;;) {
    int help_v = 0;
    if (S.yield_k > 0 && held < 0) {
      int go = 0;
      if (threadIdx.x == 0) {
        const int lv = __hip_atomic_load(S.alive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int wv = __hip_atomic_load(S.waves, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int keep = S.yield_k * lv;
        if (keep < S.yield_min) keep = S.yield_min;
        if (wv > keep) {
          if (__hip_atomic_fetch_add(S.waves, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > keep) go = 1;
          else __hip_atomic_fetch_add(S.waves, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (__builtin_amdgcn_readfirstlane(go)) break;
    }
    DDP_MARK("X_T");
#if !defined(DDP_KEEP_PRIO_WHILE_WAITING)
    // a wave that looks for work (and may have to wait for it) does so at the base priority: the raised one belongs to the
    // chunk it has just finished
    if (S.prio) __builtin_amdgcn_s_setprio(0);
#endif
    const int t = __builtin_amdgcn_readfirstlane(next_work(S, B.help, SHARE ? B.bshare : nullptr, B.idx, nb, total, total_help, held, &waited, &help_v));
    const int help = __builtin_amdgcn_readfirstlane(help_v);
    DDP_MARK("X_G");
    held = -1;
    if (t == -1) break;
    if (t == -2) continue;
    if (t <= -3) {  // timed out: retire the trajectory so that its later tickets are skipped at once
      if (threadIdx.x == 0) __hip_atomic_fetch_max(&S.done_epoch[-3 - t], kDoneBit | (int)n_epochs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      continue;
    }
    const int e = __builtin_amdgcn_readfirstlane((int)((unsigned)t / nb));
    const int b = traj_of(B, __builtin_amdgcn_readfirstlane(t - e * (int)nb));
    W.b = b;
    // A chunk whose predecessor was still running when its ticket was drawn belongs to a trajectory that lags the
    // batch, i.e. to the critical chain of the launch: it gets the SIMD's issue priority over the co-resident waves
    // (raising its helpers too made no difference).
    if (S.prio) {
      if (!help && __builtin_amdgcn_readfirstlane(waited)) __builtin_amdgcn_s_setprio(3);
      else __builtin_amdgcn_s_setprio(0);
    }
    // One call site of Wave::iterate for both kinds of work (the sweeps are inlined into it).
    int run = 1, n = 1;
    if (help) {
      held = t;  // the ticket stays in hand
      W.N = __builtin_amdgcn_readfirstlane(B.n_seg[b]);
    } else {
      W.load_state();
      run = __builtin_amdgcn_readfirstlane(lds.st.done) ? 0 : 1;
      const int left = n_iters - e * S.chunk;
      n = left < S.chunk ? left : S.chunk;
    }
#if defined(DDP_SCHED_DEBUG)
    if (!help && threadIdx.x == 0) __hip_atomic_fetch_add(&S.dbg[13], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    DDP_DBG_MARK(B, help ? 6 : 5);
    if (B.mark != nullptr && threadIdx.x == 0) B.mark[gridDim.x + blockIdx.x] = b;  // the trajectory
#endif
    if (run) W.iterate(n, help);
    DDP_DBG_MARK(B, 40);
#if defined(DDP_SCHED_DEBUG)
    if (!help && threadIdx.x == 0) __hip_atomic_fetch_add(&S.dbg[13], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    if (!help) {
#if defined(DDP_SCHED_DEBUG)
      if (threadIdx.x == 0) __hip_atomic_fetch_add(&S.dbg[26], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // entered the publish section
#endif
      if (run) W.store_state();
      const int fin = __builtin_amdgcn_readfirstlane(lds.st.done) ? kDoneBit : 0;
      int* const alive = B.live != nullptr ? B.live : S.alive;
      if (run && fin && alive != nullptr && threadIdx.x == 0)  // this chunk took the trajectory out of its outer loop
        __hip_atomic_fetch_add(alive, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      // max, not store: done_epoch only ever grows, and a trajectory that a timed-out waiter has retired
      // (kDoneBit | n_epochs, below) stays retired when its straggling chunk completes afterwards
      if (threadIdx.x == 0) {
        __hip_atomic_fetch_max(&S.done_epoch[b], (e + 1) | fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if defined(DDP_SCHED_DEBUG)
        __hip_atomic_fetch_add(&S.dbg[27], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // left it
#endif
      }
    }
  }
  DDP_MARK("Z_E");
}

// stepwise interface: mode 1 = one backwardpass(), 2 = one forwardpass(), 3 = one forwardpass() through the stored-gain form
template <typename St, int RPL>
__global__ __launch_bounds__(64) void k_pass(Batch<St> B, int mode) {
  __shared__ WaveLds<Cmp, St, RPL> lds;
  Wave<Cmp, St, RPL> W(B, lds, traj_of(B, blockIdx.x));
  W.load_state();
  if (__builtin_amdgcn_readfirstlane(lds.st.done)) return;
  W.init_tables();
  if (mode == 1) {
    W.backward_pass_stepwise();
  } else if (mode == 3 || __builtin_amdgcn_readfirstlane(lds.st.bp_failed)) {
    // after a backwardpass() that gave up the reference's forwardpass() runs with the gains its members still hold
    W.stale_fwd_pass();
  } else {
    W.fwd_pass();
  }
  W.store_state();
}

// The last trip of the outer loop of the trajectories whose backward pass got stuck in the hot kernel (ddp_optimizer.cpp:297-311,
// 392-396; Wave::iterate_once leaves them with rtn = kRtnStuckPending): launched behind every hot-kernel launch, returns at
// once for every other trajectory.
template <typename St, int RPL>
__global__ __launch_bounds__(64) void k_stuck(Batch<St> B) {
  __shared__ WaveLds<Cmp, St, RPL> lds;
  Wave<Cmp, St, RPL> W(B, lds, traj_of(B, blockIdx.x));
  W.load_state();
  if (__builtin_amdgcn_readfirstlane(lds.st.rtn) != kRtnStuckPending) return;
  W.init_tables();
  W.stuck_tail();
  W.store_state();
}

template <typename St, int RPL>
__global__ __launch_bounds__(64) void k_finish(Batch<St> B, OutPtrs<St> O, const int* sched_err) {
  __shared__ WaveLds<Cmp, St, RPL> lds;
  Wave<Cmp, St, RPL> W(B, lds, traj_of(B, blockIdx.x));
  W.load_state();
  W.init_tables();
  // a scheduler error makes the whole launch's results suspect: every row says so (direct_ddp.h)
  if (__builtin_amdgcn_readfirstlane(*sched_err)) lds.st.rtn = -101;
  finish_wave(W, O);
}

template <typename St, int RPL>
__global__ __launch_bounds__(64) void k_field(Batch<St> B, int field, St* buf, int set) {
  __shared__ WaveLds<Cmp, St, RPL> lds;
  Wave<Cmp, St, RPL> W(B, lds, traj_of(B, blockIdx.x));
  W.load_state();
  W.init_tables();
  if (set) {
    set_field_wave(W, field, (const St*)buf);
  } else {
    get_field_wave(W, field, buf);
  }
}

// UpdateTime + warm start between the two phases (teach_repeat_planner.cpp:911-921), on device.
// The reference hands phase 0's getBezCoeff() to phase 1 as initbezCoeff: control points scaled by 1 / T_0 (ddp_optimizer.cpp:799-812),
// which phase 1 converts back with ITS durations T_1 (ddp_optimizer.cpp:167-193, 782-796):  c'_i = (T_1 / T_0) c_i T_0^i / T_1^i.
// Where phase 0 found a feasible trajectory (rtn 2) UpdateTime makes T_1 = T_0 and the round trip is the identity; everywhere
// else T_1 is the caller's duration again and the warm start is the phase-0 curve in NORMALISED time, its coefficients
// scaled by (T_0 / T_1)^(i-1).  The same thing is handed over here as monomial coefficients (getPolyCoeff rows) with exactly
// that factor applied in double: float storage cannot afford the Bezier detour, and phase 1 only reads
// the u part, [c3; c4; c5].
template <typename Real>
__global__ void k_chain(int B, int nmax, const int32_t* rtn0, const Real* T_phase0, const Real* T_in,
                        const uint8_t* infeas0, const Real* poly0, Real* T_next, uint8_t* infeas_next, Real* init_poly) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * nmax) return;
  int b = i / nmax;
  const bool found = rtn0[b] == 2;
  const Real T1 = found ? T_phase0[i] : T_in[i];
  T_next[i] = T1;
  if (i % nmax == 0) infeas_next[b] = infeas0[b];
  const double q = (found || (double)T1 == 0.0) ? 1.0 : (double)T_phase0[i] / (double)T1;
  const Real* src = poly0 + (size_t)i * 18;
  Real* dst = init_poly + (size_t)i * 18;
  double f = 1.0 / q;  // q^(c - 1) for coefficient c = 0 .. 5
  for (int c = 0; c < 6; c++) {
    for (int d = 0; d < 3; d++) dst[3 * c + d] = found ? src[3 * c + d] : (Real)((double)src[3 * c + d] * f);
    f *= q;
  }
}

// initTimeAllocation (teach_repeat_planner.cpp:583-639, v0 = 0) on the device: one thread per (corridor, segment), double
// arithmetic whatever the storage type - the same expressions as direct_time_allocation() below.  Launched in front of
// k_begin when the caller passes T0 == NULL: the generation -> plan chain then never leaves the device.
template <typename Real>
__global__ void k_time_alloc(int B, int nmax, const int32_t* n_seg, const Real* x0, const Real* xd, const Real* seeds,
                             double max_vel, double max_acc, Real* T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * nmax) return;
  const int b = i / nmax, k = i - b * nmax, N = n_seg[b];
  const double acct = max_vel / max_acc, accd = max_acc * acct * acct / 2.0;
  const double dcct = max_vel / max_acc, dccd = max_acc * dcct * dcct / 2.0;
  double t = 0.0;
  if (k < N && N <= nmax) {
    double dd[3];
    for (int d = 0; d < 3; d++) {
      const double p0 = (k == 0) ? (double)x0[(size_t)b * 9 + d] : (double)seeds[((size_t)b * nmax + k) * 3 + d];
      const double p1 = (k == N - 1) ? (double)xd[(size_t)b * 9 + d] : (double)seeds[((size_t)b * nmax + k + 1) * 3 + d];
      dd[d] = p1 - p0;
    }
    // products and sums rounded one by one, as the host twin's are (no contraction into fused multiply-adds)
    const double D = sqrt(__dadd_rn(__dadd_rn(__dmul_rn(dd[0], dd[0]), __dmul_rn(dd[1], dd[1])), __dmul_rn(dd[2], dd[2])));
    if (D < accd + dccd) t = 2.0 * sqrt(max_acc * D) / max_acc;  // triangle profile
    else t = acct + (D - accd - dccd) / max_vel + dcct;          // trapezoid profile
  }
  T[i] = (Real)t;
}

// argmin of cost over problems with rtn >= 0 (config 5); one block
template <typename Real>
__global__ void k_best(const Real* cost, const int32_t* rtn, int n, int* best_idx, double* best_cost) {
  __shared__ double sv[256];
  __shared__ int si[256];
  double v = INFINITY;
  int idx = -1;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    double c = (double)cost[i];
    if (rtn[i] >= 0 && (c < v || (c == v && i < idx))) {
      v = c;
      idx = i;
    }
  }
  sv[threadIdx.x] = v;
  si[threadIdx.x] = idx;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      double v2 = sv[threadIdx.x + o];
      int i2 = si[threadIdx.x + o];
      bool take = i2 >= 0 && (si[threadIdx.x] < 0 || v2 < sv[threadIdx.x] || (v2 == sv[threadIdx.x] && i2 < si[threadIdx.x]));
      if (take) {
        sv[threadIdx.x] = v2;
        si[threadIdx.x] = i2;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *best_idx = si[0];
    *best_cost = sv[0];
  }
}

// ---- host side ---------------------------------------------------------------------------------------
static thread_local std::string g_err;
#if defined(DDP_TIMELINE)
static unsigned long long* g_tl;
static size_t g_tl_n;
#endif
static direct_status_t fail(direct_status_t st, const std::string& msg) {
  g_err = msg;
  return st;
}
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(DIRECT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));           \
  } while (0)

struct direct_ddp_handle_s {
  int dtype = 0, device = 0, max_batch = 0, nmax = 0, pmax = 0, ncs = 0, rpl = 2, fcap = 0;
  // Row-slot classes of the current batch (stage_inputs -> classify_batch): a trajectory runs on the kernels instantiated
  // for ITS widest polytope, not for the handle's p_max - one 59-plane polytope in one corridor used to put the whole
  // batch on the one-wave-per-SIMD kernels.  `order` lists the trajectories class by class; `classes` the non-empty
  // classes (empty: every trajectory runs on the handle's own class `rpl`, identity order - always so for p_max <= 12).
  struct Cls { int rpl, off, cnt; };
  std::vector<Cls> classes;
  int32_t* order = nullptr;     // device [max_batch]
  int32_t* cls_dev = nullptr;   // device [max_batch]: row slots per trajectory (scratch of the classification)
  int* tickets = nullptr;       // device [16]: one ticket counter per class launch
  int slots_of[16] = {};        // resident one-wave workgroups of k_iterate_dyn per row-slot class (0: not asked yet)
  int n_cu = 0;
  hipStream_t cstream[12] = {};  // the class launches of one iterate call run side by side (created on first use)
  hipEvent_t cev[12] = {}, fork_ev = nullptr;
  size_t rsz = 4;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
  bool sample_timed = false;
  int n_launches = 0;
  bool timed = false;
  // device buffers
  void *x0 = nullptr, *xd = nullptr, *T0 = nullptr, *planes = nullptr, *init_bez = nullptr, *T_next = nullptr;
  void* init_poly = nullptr;
  void* seeds = nullptr;
  int32_t *n_seg = nullptr, *n_planes = nullptr;
  uint8_t *infeas_in = nullptr, *infeas_next = nullptr;
  void *X[direct::kMaxBuf] = {}, *S[direct::kMaxBuf] = {}, *Y[direct::kMaxBuf] = {};
  int nbuf = 3;              // iterate buffers allocated: 3, or kMaxBuf when the line search can be shared
  HelpSlot* help = nullptr;  // [max_batch], with nbuf == kMaxBuf
  bool small_dyn = true;     // with helpers the ticket scheduler also serves batches below the resident waves (DIRECT_DDP_SMALL=0: static launch there)
  int help_early = 1;        // single-step shared searches are open from step 0 on (DIRECT_DDP_EARLY=0: after step 0 failed)
  int single_ratio = 8;      // shared line search in single steps when batch * ratio <= resident waves (DIRECT_DDP_SINGLE)
  int help_mode = -1;        // shared line search: -1 auto (batches up to 1.5 x the resident waves), DIRECT_DDP_HELP=0|1 forces
  // helper-assisted backward sweep (ddp_wave.h, bwd_sweep_t): waiting waves compute the value-independent half of knots
  BwdShare* bshare = nullptr;  // [max_batch], allocated with `help` (narrow row-slot classes only)
  int* bflag = nullptr;        // [max_batch][nmax]
  double* brec = nullptr;      // [max_batch][nmax][kRecDoubles]
  // Small handles (a single plan, a few corridors): every input array is a slice of ONE device allocation with a pinned
  // host mirror, and so is every output array - host-memory calls then cost one H2D and one D2H copy instead of one per
  // array (a two-phase plan of one 12-segment corridor: 37 small pageable copies, ~0.5 ms of its 3.2 ms).
  char *in_blob = nullptr, *out_blob = nullptr;          // device
  char *in_host = nullptr, *out_host = nullptr, *out0_host = nullptr;  // pinned mirrors (out0: phase-0 results of a plan)
  size_t in_blob_bytes = 0, out_blob_bytes = 0;
  hipEvent_t in_ev = nullptr;   // the last upload from in_host (the mirror is not rewritten before it has completed)
  bool in_ev_pending = false;
  char* batch_dev = nullptr;   // [16][1024]: device copies of the class launches' Batch structs (Batch::self)
  // their staging copies: PINNED, a ring of kBatchRing slots per class with an event each - a slot is not rewritten before the
  // upload that read it has completed (phase 1 of a plan follows phase 0 at once with other SolveConst values)
  static constexpr int kBatchRing = 4;
  char* batch_pin = nullptr;   // [16][kBatchRing][1024]
  hipEvent_t batch_ev[16][kBatchRing] = {};
  bool batch_ev_used[16][kBatchRing] = {};
  unsigned batch_seq[16] = {};
  int bshare_cap = 0;          // trajectories the shared-sweep arrays (bshare, bflag, brec) are allocated for
  int bshare_mode = -1;        // -1 auto (wherever the line search is shared), DIRECT_DDP_BSHARE=0 off, 1 on, 2 forced split: every
                               // owner runs the helpers' half itself first (tests: the hand-over path without any helper)
  void *KU = nullptr, *KS = nullptr, *KY = nullptr;
  double* filt = nullptr;
  TrajState* st = nullptr;
  GainBase* gbase = nullptr;   // [max_batch] what each trajectory's gains were formed from (Wave::stale_fwd_pass)
  void* fieldbuf = nullptr;
  size_t fieldbuf_bytes = 0;
  // staged outputs (when the caller's buffers are host memory)
  struct {
    int32_t *rtn, *iter_used, *fwd_passes;
    uint8_t *infeas_out, *line_failed_out;
    void *cost, *costq, *jerk_cost, *terminal_norm2, *opterr, *mu, *bez, *poly, *T;
  } o = {};
  int *best_idx = nullptr;
  double* best_cost = nullptr;
  // config-5 gather (allocated on first use): [n_ranks] records, [n_ranks][nmax*19] blocks, winner, owner
  void *g_recs = nullptr, *g_blocks = nullptr, *g_win = nullptr, *g_in = nullptr;
  int g_ranks = 0;
  // dynamic scheduling of k_iterate_dyn: [0] ticket, [1] error flag, [2..] done_epoch[max_batch]
  int* sched = nullptr;
  int* live = nullptr;                   // [1] unfinished trajectories of the running launch (Batch::live)
  unsigned long long* visits = nullptr;  // [2] sweep-work counters of the last hot-kernel launch (direct_ddp_last_launch_info)
  direct_ddp_launch_info_t last_info = {};
  int sched_slots = 0;   // resident one-wave workgroups of k_iterate_dyn on this device (the handle's own class)
  int slots_cap = 0;     // DIRECT_DDP_SLOTS: fewer persistent waves than fit (experiments)
  int sched_chunk = 1;   // outer-loop trips per ticket (DIRECT_DDP_CHUNK at create time; experiments)
  int pair_trials = -1;  // two line-search steps per forward sweep from the second attempt on: -1 auto, DIRECT_DDP_PAIR=0|1 forces
  int sched_prio = 1;    // chunks that had to wait for their predecessor run at raised wave priority (DIRECT_DDP_PRIO=0: off)
  int sched_tail = 8;    // rounds of help-only tickets behind the last epoch (DIRECT_DDP_TAIL=0: none; next_work)
  int yield_k = 0;       // DIRECT_FLAG_YIELD / DIRECT_DDP_YIELD=k: surplus waves leave the hot kernel (Sched::yield_k); 0 = off
  int* nwaves = nullptr; // device [16]: waves inside the kernel, per class launch
  int* sched_dbg = nullptr;  // device [64]: Sched::dbg
  int* sched_mark = nullptr; // development builds: device [2 x 8192]
  bool dynamic = true;
  // current batch
  int B = 0;
  bool begun = false;
  direct_ddp_params_t params = {};
  direct_ddp_batch_in_t cur_in = {};
  std::vector<void*> allocs;
};

template <typename T>
static direct_status_t dalloc(direct_ddp_handle_t h, T** p, size_t bytes) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, bytes ? bytes : 16);
  if (e != hipSuccess) return fail(DIRECT_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
  h->allocs.push_back(q);
  *p = (T*)q;
  return DIRECT_OK;
}
#define TRY(expr)                            \
  do {                                       \
    direct_status_t s_ = (expr);             \
    if (s_ != DIRECT_OK) return s_;          \
  } while (0)

// the classes of the current batch: the recorded ones, or the whole batch on the handle's own class
static std::vector<direct_ddp_handle_s::Cls> batch_classes(direct_ddp_handle_t h) {
  if (!h->classes.empty()) return h->classes;
  return {direct_ddp_handle_s::Cls{h->rpl, 0, h->B}};
}
template <typename Real>
static int resident_slots(direct_ddp_handle_t h, int rpl);

template <typename Real>
static Batch<Real> make_batch(direct_ddp_handle_t h, const direct_ddp_batch_in_t& in, const direct_ddp_params_t& p,
                              const direct_ddp_handle_s::Cls& c) {
  Batch<Real> B;
  memset(&B, 0, sizeof(B));
  B.B = c.cnt; B.idx = h->classes.empty() ? nullptr : h->order + c.off;
  B.nmax = h->nmax; B.pmax = h->pmax; B.ncs = h->ncs; B.fcap = h->fcap;
  B.n_seg = in.n_seg; B.x0 = (const Real*)in.x0; B.xd = (const Real*)in.xd; B.T0 = (const Real*)in.T0;
  B.n_planes = in.n_planes; B.planes = (const Real*)in.planes; B.init_bez = (const Real*)in.init_bez;
  B.init_poly = (const Real*)in.init_poly;
  B.seeds = (const Real*)in.seeds;
  B.infeas_in = in.infeas_in;
  for (int i = 0; i < h->nbuf; i++) {
    B.X[i] = (Real*)h->X[i]; B.S[i] = (Real*)h->S[i]; B.Y[i] = (Real*)h->Y[i];
  }
  B.nbuf = h->nbuf;
  B.help = nullptr;  // set by the dynamic launch only
  B.sched_err = h->sched + 1;
  B.visits = nullptr;  // set by launch_iterate_t (the hot-kernel launch only)
  B.live = nullptr;    // set by the ticket-scheduled launch with a shared line search
  B.KU = (Real*)h->KU; B.KS = (Real*)h->KS; B.KY = (Real*)h->KY; B.filt = h->filt; B.st = h->st;
  B.gbase = h->gbase;
  SolveConst& k = B.k;
  k.max_vel = p.max_vel; k.max_acc = p.max_acc; k.w_snap = p.w_snap; k.w_term = p.w_terminal;
  k.w_time = p.w_time;
  k.reg_base = p.zero_init ? 1.6 : 4.0;  // ddp_optimizer.cpp:60-61
  k.shift = p.minvo ? 0.0 : 2.0e-4;      // ddp_optimizer.cpp:1281-1283
  k.tol = 1.0e-7;                        // ddp_optimizer.cpp:43
  k.iter_max = p.iter_max; k.time_power = p.time_power; k.zero_init = p.zero_init;
  k.line_init = p.line_init; k.minvo = p.minvo; k.fixed_iters = p.fixed_iters; k.exact_dt = p.exact_dt;
  // two trials per sweep: +8 % where the launch is bound by its slowest chain (B = 4096) and, since the trials share
  // the old iterate's row values (c, 1 / c: run_round, phase R), +0.7 % where it is throughput-bound (B = 16384; r03: -1 %);
  // results are identical either way
  const int slots = resident_slots<Real>(h, c.rpl);
  k.pair_trials = h->pair_trials >= 0 ? h->pair_trials : 1;
  // few trajectories on many waves and a shared line search: single steps, one wave each, beat pairs
  if (h->pair_trials < 0 && h->help && h->help_mode != 0 && h->dynamic && h->nmax >= 80 && (long long)c.cnt * h->single_ratio <= (long long)slots)
    k.pair_trials = 0;
  return B;
}

// Row-slot classes the kernels are instantiated for: 2 .. 8 slots per lane (polytopes of up to 12, 22, 33, 44, 54, 65, 76
// planes) and 10, 12, 14 (97, 118, 128 = DIRECT_P_LIMIT).  Class numbers passed to RPL_LAUNCH are always one of them.
#if defined(DDP_DEV_RPL2)  // development builds (tools/fastbuild.sh): only the two-slot kernels are instantiated, 7 x faster to compile
#define RPL_LAUNCH(rpl, stream, KERNEL, Real, grid, ...) \
  hipLaunchKernelGGL((KERNEL<Real, 2>), dim3(grid), dim3(64), 0, stream, __VA_ARGS__)
#else
#define RPL_CASE(stream, KERNEL, Real, grid, R, ...) \
  case R: hipLaunchKernelGGL((KERNEL<Real, R>), dim3(grid), dim3(64), 0, stream, __VA_ARGS__); break;
#define RPL_LAUNCH(rpl, stream, KERNEL, Real, grid, ...)            \
  do {                                                              \
    switch (rpl) {                                                  \
      RPL_CASE(stream, KERNEL, Real, grid, 2, __VA_ARGS__)          \
      RPL_CASE(stream, KERNEL, Real, grid, 3, __VA_ARGS__)          \
      RPL_CASE(stream, KERNEL, Real, grid, 4, __VA_ARGS__)          \
      RPL_CASE(stream, KERNEL, Real, grid, 5, __VA_ARGS__)          \
      RPL_CASE(stream, KERNEL, Real, grid, 6, __VA_ARGS__)          \
      RPL_CASE(stream, KERNEL, Real, grid, 7, __VA_ARGS__)          \
      RPL_CASE(stream, KERNEL, Real, grid, 8, __VA_ARGS__)          \
      RPL_CASE(stream, KERNEL, Real, grid, 10, __VA_ARGS__)         \
      RPL_CASE(stream, KERNEL, Real, grid, 12, __VA_ARGS__)         \
      default: hipLaunchKernelGGL((KERNEL<Real, 14>), dim3(grid), dim3(64), 0, stream, __VA_ARGS__); \
    }                                                               \
  } while (0)
#endif

// smallest instantiated row-slot class that holds `slots` row slots per lane
static int rpl_class(int slots) {
#if defined(DDP_DEV_RPL2)
  (void)slots;
  return 2;
#else
  return slots <= 2 ? 2 : (slots <= 8 ? slots : (slots <= 10 ? 10 : (slots <= 12 ? 12 : 14)));
#endif
}
// resident one-wave workgroups of k_iterate_dyn of one class on this device (asked once per class)
template <typename Real>
static int resident_slots(direct_ddp_handle_t h, int rpl) {
  if (rpl < 0 || rpl > 15) return 0;
  if (h->slots_of[rpl] != 0) return h->slots_of[rpl] < 0 ? 0 : h->slots_of[rpl];
  int per_cu = 0;
  hipError_t e;
#if defined(DDP_DEV_RPL2)
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_iterate_dyn<Real, 2>, 64, 0);
#else
  switch (rpl) {
#define OCC_CASE(R) case R: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_iterate_dyn<Real, R>, 64, 0); break;
    OCC_CASE(2) OCC_CASE(3) OCC_CASE(4) OCC_CASE(5) OCC_CASE(6) OCC_CASE(7) OCC_CASE(8) OCC_CASE(10) OCC_CASE(12)
#undef OCC_CASE
    default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_iterate_dyn<Real, 14>, 64, 0);
  }
#endif
  int v = (e == hipSuccess && per_cu > 0) ? per_cu * h->n_cu : 0;
  if (h->slots_cap > 0 && v > h->slots_cap) v = h->slots_cap;  // DIRECT_DDP_SLOTS (experiments)
  h->slots_of[rpl] = v > 0 ? v : -1;
  return v;
}

template <typename Real>
static void launch_begin_t(direct_ddp_handle_t h) {
  for (const auto& c : batch_classes(h)) {
    auto Bt = make_batch<Real>(h, h->cur_in, h->params, c);
    RPL_LAUNCH(c.rpl, h->stream, k_begin, Real, c.cnt, Bt);
  }
}
// The classes of one iterate call run SIDE BY SIDE, each on a stream of its own forked from the handle's: a small
// batch is bound by its longest chain of iterations, and one class after the other would add those chains up.
static hipStream_t class_stream(direct_ddp_handle_t h, size_t ci, size_t n_classes) {
  if (n_classes <= 1 || ci >= 12) return h->stream;
  if (!h->fork_ev && hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming) != hipSuccess) return h->stream;
  if (!h->cstream[ci]) {
    if (hipStreamCreateWithFlags(&h->cstream[ci], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->cev[ci], hipEventDisableTiming) != hipSuccess) {
      h->cstream[ci] = nullptr;
      return h->stream;
    }
  }
  return h->cstream[ci];
}
template <typename Real>
static void launch_iterate_t(direct_ddp_handle_t h, int n, int mode) {
  const auto classes = batch_classes(h);
  (void)hipMemsetAsync(h->visits, 0, 4 * sizeof(unsigned long long), h->stream);
  direct_ddp_launch_info_t& li = h->last_info;
  li = direct_ddp_launch_info_t{};
  li.n_buffers = h->nbuf; li.batch = h->B;
  if (mode == 0) {
    (void)hipMemsetAsync(h->tickets, 0, 16 * sizeof(int), h->stream);
    (void)hipMemsetAsync(h->sched + 2, 0, (size_t)h->B * sizeof(int), h->stream);  // done_epoch; the error flag [1] is sticky: cleared in stage_inputs
    if (h->help) (void)hipMemsetAsync(h->help, 0, (size_t)h->B * sizeof(HelpSlot), h->stream);
    if (h->bshare && h->bshare_mode != 0) {  // sweep tags restart with every launch: no flag of an earlier one may survive
      const size_t nb = (size_t)(h->B < h->bshare_cap ? h->B : h->bshare_cap);
      (void)hipMemsetAsync(h->bshare, 0, nb * sizeof(BwdShare), h->stream);
      (void)hipMemsetAsync(h->bflag, 0, nb * h->nmax * sizeof(int), h->stream);
    }
  }
  bool forked = false;
  if (mode == 0 && classes.size() > 1 && class_stream(h, 0, classes.size()) != h->stream) {
    (void)hipEventRecord(h->fork_ev, h->stream);
    forked = true;
  }
  int biggest = -1;
  for (size_t ci = 0; ci < classes.size(); ci++) {
    const auto& c = classes[ci];
    auto Bt = make_batch<Real>(h, h->cur_in, h->params, c);
    Bt.visits = h->visits;
#if defined(DDP_TIMELINE)
    if (g_tl_n < (size_t)h->B * kTimelineDepth * 4) {
      if (g_tl) (void)hipFree(g_tl);
      g_tl_n = (size_t)h->B * kTimelineDepth * 4;
      (void)hipMalloc((void**)&g_tl, g_tl_n * 8);
    }
    (void)hipMemsetAsync(g_tl, 0, (size_t)h->B * kTimelineDepth * 4 * 8, h->stream);
    Bt.tl = g_tl;
#endif
    hipStream_t st = h->stream;
    if (forked) {
      st = class_stream(h, ci, classes.size());
      if (st != h->stream) (void)hipStreamWaitEvent(st, h->fork_ev, 0);
    }
    if (mode != 0) {
      RPL_LAUNCH(c.rpl, st, k_pass, Real, c.cnt, Bt, mode);
      continue;
    }
    const int slots = resident_slots<Real>(h, c.rpl);
    // shared line search: where the launch is bound by its slowest chain and a round is long enough to pay for the
    // protocol's fences (agent-scope release = L2 write-back).  Measured at N = 100: +5 .. +13 % for batches up to 4/3 of
    // the resident waves, -1.4 % at twice the resident waves, +15 .. +20 % below them; at N = 60: -1 .. -5 %; at
    // N = 30: up to -35 %.  Unless forced either way.
    // (short trajectories: only where most of the device is idle anyway - a lone 12-segment plan 3.6 -> 3.2 ms with helpers)
    const bool help = h->help && (h->help_mode >= 0 ? h->help_mode != 0 : (2 * c.cnt <= 3 * slots && (h->nmax >= 80 || 8 * c.cnt <= slots)));
    const bool is_big = biggest < 0 || c.cnt > classes[biggest].cnt;
    if (is_big) {
      biggest = (int)ci;
      li.pair_trials = Bt.k.pair_trials; li.resident_waves = slots; li.dynamic = 0; li.shared_search = 0; li.single_steps = 0;
      li.shared_sweep = 0;
    }
    // backward sweeps shared with helpers wherever the line search is; the forced split needs no helper (any launch form)
    // (the arrays are indexed by the trajectory's own number: the whole current batch must fit what was allocated)
    const bool sweep_ok = h->bshare != nullptr && c.rpl <= 4 && h->B <= h->bshare_cap;
    static_assert(sizeof(Batch<Real>) <= 1024, "Batch grew past its device-copy slot");
    auto publish = [&](Batch<Real>& Bb) {  // the struct itself in device memory, for the out-of-line halves of a shared sweep
      if (ci >= 16 || h->batch_pin == nullptr) { Bb.bshare = nullptr; return; }
      const int slot = (int)(h->batch_seq[ci]++ % direct_ddp_handle_s::kBatchRing);
      if (h->batch_ev_used[ci][slot]) (void)hipEventSynchronize(h->batch_ev[ci][slot]);  // the upload that last read this slot
      char* stage = h->batch_pin + ((size_t)ci * direct_ddp_handle_s::kBatchRing + slot) * 1024;
      Bb.self = h->batch_dev + ci * 1024;
      memcpy(stage, &Bb, sizeof(Bb));
      (void)hipMemcpyAsync(h->batch_dev + ci * 1024, stage, sizeof(Bb), hipMemcpyHostToDevice, st);
      if (h->batch_ev[ci][slot] == nullptr && hipEventCreateWithFlags(&h->batch_ev[ci][slot], hipEventDisableTiming) != hipSuccess) {
        h->batch_ev[ci][slot] = nullptr;
        (void)hipStreamSynchronize(st);  // no event: the slot is free again once the stream has drained
        h->batch_ev_used[ci][slot] = false;
      } else {
        (void)hipEventRecord(h->batch_ev[ci][slot], st);
        h->batch_ev_used[ci][slot] = true;
      }
    };
    if (sweep_ok && h->bshare_mode == 2) {
      Bt.bshare = h->bshare; Bt.bflag = h->bflag; Bt.brec = h->brec; Bt.bforce = 1; Bt.bvisits = h->visits + 2;
      if (is_big) li.shared_sweep = 2;
    }
    // The static launch is tail-free when every trajectory is resident at once - but then the waves that are left
    // over have nothing to do, while the ticket scheduler turns them into helpers: with help it is used for small
    // batches too (DIRECT_DDP_SMALL=0: not below the resident waves).
    if (h->dynamic && slots > 0 && (c.cnt > slots || (help && h->small_dyn))) {
      Sched S;
      S.ticket = (unsigned*)(h->tickets + (ci < 16 ? ci : 15));
      S.err = h->sched + 1;
      S.done_epoch = h->sched + 2;
      S.chunk = h->sched_chunk;
      S.prio = h->sched_prio;
      S.tail = help ? h->sched_tail : 0;
      S.dbg = h->sched_dbg;
#if defined(DDP_SCHED_DEBUG)
      S.mark = nullptr; S.grid = slots;
      if (slots <= 8192) { S.mark = h->sched_mark; Bt.mark = h->sched_mark; (void)hipMemsetAsync(h->sched_mark, 0, 2 * 8192 * sizeof(int), st); }
#endif
      S.waves = h->nwaves + (ci < 16 ? ci : 15);
      S.alive = h->live + (ci < 16 ? ci : 15);
      S.yield_k = h->yield_k;
      S.yield_min = 64;
      if (h->yield_k > 0) {
        S.tail = 0;  // help-only tickets would keep every wave inside until the last chunk
        (void)hipMemsetD32Async((hipDeviceptr_t)S.waves, slots, 1, st);
        (void)hipMemsetD32Async((hipDeviceptr_t)S.alive, c.cnt, 1, st);
      }
      if (help) {
        Bt.help = h->help;
        Bt.help_early = h->help_early;
        // the launch's tail: when at most slots / single_ratio trajectories are left (natural exits), their line
        // searches go to single steps open from step 0, as a batch that small would from the start
        int* live = h->live + (ci < 16 ? ci : 15);
        (void)hipMemsetD32Async((hipDeviceptr_t)live, c.cnt, 1, st);
        Bt.live = live;
        Bt.tail_thresh = h->single_ratio > 0 ? slots / h->single_ratio : 0;
        if (is_big) { li.shared_search = 1; li.single_steps = Bt.k.pair_trials ? 0 : 1; }
        // Shared backward sweeps pay where most waves are idle (a lone N = 100 trajectory 12.2 -> 10.3 ms per 20 iterations,
        // B = 256 19.1 -> 18.8 ms); from a third of the resident waves on, the waiters' help no longer covers the sharing
        // instantiation's slower fused path (B = 1024 22.3 -> 23.6 ms, 3072 29.5 -> 32.0, 4096 35.1 -> 37.8; same-box A/B):
        // auto = batches up to an eighth of the resident waves, like the single-step line search
        if (sweep_ok && h->bshare_mode != 0 && h->bshare_mode != 2 && h->sched_chunk == 1 && (h->bshare_mode > 0 || 8 * c.cnt <= slots)) {
          Bt.bshare = h->bshare; Bt.bflag = h->bflag; Bt.brec = h->brec; Bt.bforce = 0; Bt.bvisits = h->visits + 2;
          if (is_big) li.shared_sweep = 1;
        }
      }
      if (is_big) li.dynamic = 1;
      if (Bt.bshare) {
        publish(Bt);
        switch (c.rpl) {  // sweep_ok: row-slot classes 2 .. 4
          case 2: hipLaunchKernelGGL((k_iterate_dyn<Real, 2, true>), dim3(slots), dim3(64), 0, st, Bt, n, S); break;
#if !defined(DDP_DEV_RPL2)
          case 3: hipLaunchKernelGGL((k_iterate_dyn<Real, 3, true>), dim3(slots), dim3(64), 0, st, Bt, n, S); break;
          default: hipLaunchKernelGGL((k_iterate_dyn<Real, 4, true>), dim3(slots), dim3(64), 0, st, Bt, n, S);
#endif
        }
      } else {
        RPL_LAUNCH(c.rpl, st, k_iterate_dyn, Real, slots, Bt, n, S);
      }
    } else {
      Bt.bshare = nullptr;  // the one-workgroup-per-trajectory kernel has no waiting waves and no sharing instantiation
      if (is_big && li.shared_sweep) li.shared_sweep = 0;
      RPL_LAUNCH(c.rpl, st, k_iterate, Real, c.cnt, Bt, n);
    }
    {  // trajectories whose backward pass got stuck finish their last trip here (rare; every other workgroup returns at once)
      auto Bs = make_batch<Real>(h, h->cur_in, h->params, c);
      Bs.visits = h->visits;
      RPL_LAUNCH(c.rpl, st, k_stuck, Real, c.cnt, Bs);
    }
    if (forked && st != h->stream) {
      (void)hipEventRecord(h->cev[ci], st);
      (void)hipStreamWaitEvent(h->stream, h->cev[ci], 0);
    }
  }
}
template <typename Real>
static void launch_field_t(direct_ddp_handle_t h, int field, int set) {
  for (const auto& c : batch_classes(h)) {
    auto Bt = make_batch<Real>(h, h->cur_in, h->params, c);
    RPL_LAUNCH(c.rpl, h->stream, k_field, Real, c.cnt, Bt, field, (Real*)h->fieldbuf, set);
  }
}
template <typename Real>
static OutPtrs<Real> out_ptrs(direct_ddp_handle_t h, const direct_ddp_batch_out_t* out) {
  OutPtrs<Real> O;
  const bool dev = out->mem == DIRECT_MEM_DEVICE;
  // device memory: write straight into the caller's arrays; host memory: stage in the handle
#define PICK(f) (dev ? out->f : (out->f ? h->o.f : nullptr))
  O.rtn = PICK(rtn); O.iter_used = PICK(iter_used); O.fwd_passes = PICK(fwd_passes);
  O.infeas_out = PICK(infeas_out); O.line_failed_out = PICK(line_failed_out);
  O.cost = (Real*)PICK(cost); O.costq = (Real*)PICK(costq); O.jerk_cost = (Real*)PICK(jerk_cost);
  O.terminal_norm2 = (Real*)PICK(terminal_norm2); O.opterr = (Real*)PICK(opterr); O.mu = (Real*)PICK(mu);
  O.bez = (Real*)PICK(bez); O.poly = (Real*)PICK(poly); O.T = (Real*)PICK(T);
#undef PICK
  return O;
}
template <typename Real>
static void launch_finish_t(direct_ddp_handle_t h, const direct_ddp_batch_out_t* out) {
  auto O = out_ptrs<Real>(h, out);
  for (const auto& c : batch_classes(h)) {
    auto Bt = make_batch<Real>(h, h->cur_in, h->params, c);
    RPL_LAUNCH(c.rpl, h->stream, k_finish, Real, c.cnt, Bt, O, (const int*)(h->sched + 1));
  }
}

// direct_traj_sample_batch for one storage type: host arrays are staged through temporary device buffers
template <typename Real>
static direct_status_t sample_t(direct_ddp_handle_t h, const direct_sample_in_t* in, direct_sample_out_t* out) {
  const size_t B = in->batch, nm = in->n_seg_max, cap = in->capacity, r = sizeof(Real);
  const bool host = in->mem == DIRECT_MEM_HOST;
  SampleArgs<Real> A;
  A.batch = in->batch; A.nmax = in->n_seg_max; A.capacity = in->capacity; A.derivs = in->derivs; A.dt = in->dt; A.inv_dt = 1.0 / in->dt;
  std::vector<void*> tmp;
  auto dev = [&](size_t bytes) -> void* {
    void* q = nullptr;
    if (hipMalloc(&q, bytes ? bytes : 16) != hipSuccess) return nullptr;
    tmp.push_back(q);
    return q;
  };
  auto cleanup = [&]() { for (void* q : tmp) (void)hipFree(q); };
  auto in_arr = [&](const void* src, size_t bytes) -> const void* {
    if (!host) return src;
    void* q = dev(bytes);
    if (q && hipMemcpyAsync(q, src, bytes, hipMemcpyHostToDevice, h->stream) != hipSuccess) return nullptr;
    return q;
  };
  // staged outputs are zero-filled: entries past `count` then read 0 on the host (device-resident
  // output arrays are left untouched past `count`)
  auto out_arr = [&](void* dst, size_t bytes) -> void* {
    if (!host || !dst) return dst;
    void* q = dev(bytes);
    if (q && hipMemsetAsync(q, 0, bytes, h->stream) != hipSuccess) return nullptr;
    return q;
  };
  A.n_seg = (const int32_t*)in_arr(in->n_seg, B * 4);
  A.bez = (const Real*)in_arr(in->bez, B * nm * 18 * r);
  A.T = (const Real*)in_arr(in->T, B * nm * r);
  A.count = (int32_t*)out_arr(out->count, B * 4);
  A.seg_first = (int32_t*)out_arr(out->seg_first, B * nm * 4);
  A.pos = (Real*)out_arr(out->pos, B * cap * 3 * r);
  A.vel = (Real*)out_arr(out->vel, B * cap * 3 * r);
  A.acc = (Real*)out_arr(out->acc, B * cap * 3 * r);
  A.length = (Real*)out_arr(out->length, B * r);
  A.vmax = (Real*)out_arr(out->vmax, B * r);
  A.amax = (Real*)out_arr(out->amax, B * r);
  A.cmax = (Real*)out_arr(out->cmax, B * r);
  A.pmax = in->p_max;
  A.n_planes = out->cmax ? (const int32_t*)in_arr(in->n_planes, B * nm * 4) : nullptr;
  A.planes = out->cmax ? (const Real*)in_arr(in->planes, B * nm * (size_t)in->p_max * 4 * r) : nullptr;
  if (out->cmax && (!A.n_planes || !A.planes || !A.cmax)) {
    cleanup();
    return fail(DIRECT_ERR_DEVICE, "staging buffers for direct_traj_sample_batch");
  }
  if (!A.n_seg || !A.bez || !A.T || !A.count || !A.pos || (out->vel && !A.vel) || (out->acc && !A.acc)) {
    cleanup();
    return fail(DIRECT_ERR_DEVICE, "staging buffers for direct_traj_sample_batch");
  }
  (void)hipEventRecord(h->ev2, h->stream);
  hipLaunchKernelGGL(k_sample<Real>, dim3(in->batch), dim3(64), 0, h->stream, A);
  (void)hipEventRecord(h->ev3, h->stream);
  h->sample_timed = true;
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && host) {
    auto dn = [&](void* dst, const void* src, size_t bytes) {
      return dst ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream) : hipSuccess;
    };
    if (e == hipSuccess) e = dn(out->count, A.count, B * 4);
    if (e == hipSuccess) e = dn(out->seg_first, A.seg_first, B * nm * 4);
    if (e == hipSuccess) e = dn(out->pos, A.pos, B * cap * 3 * r);
    if (e == hipSuccess) e = dn(out->vel, A.vel, B * cap * 3 * r);
    if (e == hipSuccess) e = dn(out->acc, A.acc, B * cap * 3 * r);
    if (e == hipSuccess) e = dn(out->length, A.length, B * r);
    if (e == hipSuccess) e = dn(out->vmax, A.vmax, B * r);
    if (e == hipSuccess) e = dn(out->amax, A.amax, B * r);
    if (e == hipSuccess) e = dn(out->cmax, A.cmax, B * r);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  }
  if (host) { (void)hipStreamSynchronize(h->stream); cleanup(); }
  if (e != hipSuccess) return fail(DIRECT_ERR_DEVICE, std::string("direct_traj_sample_batch: ") + hipGetErrorString(e));
  return DIRECT_OK;
}

extern "C" {

int32_t direct_ddp_abi_version(void) { return DIRECT_DDP_ABI_VERSION; }
const char* direct_ddp_last_error(void) { return g_err.c_str(); }

direct_status_t direct_ddp_create(const direct_ddp_config_t* cfg, direct_ddp_handle_t* out) {
  if (!cfg || !out) return fail(DIRECT_ERR_INVALID, "null argument");
  if (cfg->dtype != DIRECT_F32 && cfg->dtype != DIRECT_F64) return fail(DIRECT_ERR_INVALID, "bad dtype");
  if (cfg->max_batch <= 0 || cfg->n_seg_max <= 0 || cfg->p_max <= 0) return fail(DIRECT_ERR_INVALID, "bad sizes");
  if (cfg->p_max > DIRECT_P_LIMIT) return fail(DIRECT_ERR_UNSUPPORTED, "p_max > DIRECT_P_LIMIT");
#if defined(DDP_DEV_RPL2)  // development builds carry the two-slot kernels only: (2 * 64 - 55) / 6 planes
  if (cfg->p_max > 12) return fail(DIRECT_ERR_UNSUPPORTED, "development build (DDP_DEV_RPL2): p_max > 12");
#endif
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(DIRECT_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(DIRECT_ERR_INVALID, "bad device ordinal");
  HIP_TRY(hipSetDevice(cfg->device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return fail(DIRECT_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
  direct_ddp_handle_t h = new direct_ddp_handle_s();
  h->dtype = cfg->dtype; h->device = cfg->device; h->max_batch = cfg->max_batch;
  h->nmax = cfg->n_seg_max; h->pmax = cfg->p_max;
  h->rsz = cfg->dtype == DIRECT_F64 ? 8 : 4;
  const int ncm = 6 * cfg->p_max + 55;
  h->ncs = (ncm + 3) / 4 * 4;
  h->rpl = rpl_class((ncm + 63) / 64);
  h->fcap = 0;
  const size_t B = cfg->max_batch, nm = cfg->n_seg_max, r = h->rsz;
  const size_t xs = cfg->dtype == DIRECT_F64 ? x_stride<double>() : x_stride<float>();  // knot record of the iterate
  direct_status_t st = DIRECT_OK;
  auto A = [&](auto pp, size_t bytes) { if (st == DIRECT_OK) st = dalloc(h, pp, bytes); };
  // slices of one allocation (small handles, see in_blob) or allocations of their own
  struct Slice { void** pp; size_t off; };
  auto carve = [&](std::vector<Slice>& v, size_t& total, auto pp, size_t bytes) {
    v.push_back(Slice{(void**)pp, total});
    total += (bytes + 255) & ~(size_t)255;
  };
  std::vector<Slice> in_sl, out_sl;
  size_t in_total = 0, out_total = 0;
  carve(in_sl, in_total, &h->x0, B * 9 * r); carve(in_sl, in_total, &h->xd, B * 9 * r); carve(in_sl, in_total, &h->T0, B * nm * r);
  carve(in_sl, in_total, &h->planes, B * nm * cfg->p_max * 4 * r); carve(in_sl, in_total, &h->init_bez, B * nm * 18 * r);
  carve(in_sl, in_total, &h->init_poly, B * nm * 18 * r); carve(in_sl, in_total, &h->seeds, B * nm * 3 * r);
  carve(in_sl, in_total, &h->n_seg, B * 4); carve(in_sl, in_total, &h->n_planes, B * nm * 4); carve(in_sl, in_total, &h->infeas_in, B);
  carve(out_sl, out_total, &h->o.rtn, B * 4); carve(out_sl, out_total, &h->o.iter_used, B * 4); carve(out_sl, out_total, &h->o.fwd_passes, B * 4);
  carve(out_sl, out_total, &h->o.infeas_out, B); carve(out_sl, out_total, &h->o.line_failed_out, B);
  carve(out_sl, out_total, &h->o.cost, B * r); carve(out_sl, out_total, &h->o.costq, B * r); carve(out_sl, out_total, &h->o.jerk_cost, B * r);
  carve(out_sl, out_total, &h->o.terminal_norm2, B * r); carve(out_sl, out_total, &h->o.opterr, B * r); carve(out_sl, out_total, &h->o.mu, B * r);
  carve(out_sl, out_total, &h->o.bez, B * nm * 18 * r); carve(out_sl, out_total, &h->o.poly, B * nm * 18 * r); carve(out_sl, out_total, &h->o.T, B * nm * r);
  const bool packed = in_total + out_total <= (512u << 10) && !getenv("DIRECT_DDP_NO_PACK");
  if (packed) {
    A(&h->in_blob, in_total); A(&h->out_blob, out_total);
    if (st == DIRECT_OK && (hipHostMalloc((void**)&h->in_host, in_total) != hipSuccess || hipHostMalloc((void**)&h->out_host, out_total) != hipSuccess ||
                            hipHostMalloc((void**)&h->out0_host, out_total) != hipSuccess || hipEventCreateWithFlags(&h->in_ev, hipEventDisableTiming) != hipSuccess))
      st = fail(DIRECT_ERR_DEVICE, "pinned staging buffers");
    if (st == DIRECT_OK) {
      for (const Slice& q : in_sl) *q.pp = h->in_blob + q.off;
      for (const Slice& q : out_sl) *q.pp = h->out_blob + q.off;
      h->in_blob_bytes = in_total; h->out_blob_bytes = out_total;
    }
  } else {
    A(&h->x0, B * 9 * r); A(&h->xd, B * 9 * r); A(&h->T0, B * nm * r);
    A(&h->planes, B * nm * cfg->p_max * 4 * r); A(&h->init_bez, B * nm * 18 * r); A(&h->init_poly, B * nm * 18 * r);
    A(&h->seeds, B * nm * 3 * r);
    A(&h->n_seg, B * 4); A(&h->n_planes, B * nm * 4); A(&h->infeas_in, B);
  }
  A(&h->T_next, B * nm * r); A(&h->infeas_next, B);
  h->dynamic = !(cfg->reserved & DIRECT_FLAG_STATIC_SCHEDULE);
  // measured (two handles, B = 4096, N = 100, natural exits; M iter/s): k = 12: 1.96, 6: 2.05, 3: 2.15, 2: 2.24, 1: 2.37 (serial 1.76);
  // three or four handles in rotation are no better (2.26 / 2.08): tools/pipeline_bench.py, profiles/r06_pipeline.json
  if (cfg->reserved & DIRECT_FLAG_YIELD) h->yield_k = 1;
  if (const char* ev = getenv("DIRECT_DDP_YIELD")) h->yield_k = std::max(0, std::min(atoi(ev), 64));
  if (const char* ev = getenv("DIRECT_DDP_SCHED")) h->dynamic = std::string(ev) != "static";
  h->n_cu = prop.multiProcessorCount;
  if (const char* ev = getenv("DIRECT_DDP_SLOTS")) h->slots_cap = atoi(ev) > 0 ? atoi(ev) : 0;  // experiments: fewer persistent waves than fit
  h->sched_slots = cfg->dtype == DIRECT_F64 ? resident_slots<double>(h, h->rpl) : resident_slots<float>(h, h->rpl);
  if (const char* ev = getenv("DIRECT_DDP_HELP")) h->help_mode = atoi(ev);
  if (const char* ev = getenv("DIRECT_DDP_SINGLE")) h->single_ratio = atoi(ev) > 0 ? atoi(ev) : (1 << 30);
  if (const char* ev = getenv("DIRECT_DDP_SMALL")) h->small_dyn = atoi(ev) != 0;
  if (const char* ev = getenv("DIRECT_DDP_EARLY")) h->help_early = atoi(ev);
  if (const char* ev = getenv("DIRECT_DDP_BSHARE")) h->bshare_mode = atoi(ev);
  // The shared line search needs a trial buffer per step.  It only ever runs where trials are paired, i.e. (unless
  // forced) for batches up to twice the resident waves: larger handles keep the three-buffer layout.
  const bool can_help = h->dynamic && h->sched_slots > 0 && h->help_mode != 0 &&
                        (h->help_mode > 0 || (cfg->max_batch <= 2 * h->sched_slots && cfg->n_seg_max >= 80) || 8 * cfg->max_batch <= h->sched_slots);
  bool fits = true;
  if (can_help && h->help_mode < 0) {  // the extra trial buffers must stay a small part of the device's memory
    size_t free_b = 0, total_b = 0;
    const size_t extra = (size_t)(kMaxBuf - 3) * (B * (nm + 1) * xs * r + 2 * B * nm * h->ncs * r) +
                         (h->rpl <= 4 ? B * nm * (size_t)kRecDoubles * 8 : 0);
    fits = hipMemGetInfo(&free_b, &total_b) == hipSuccess && extra <= free_b / 8;
  }
  h->nbuf = (can_help && fits) ? kMaxBuf : 3;
  for (int i = 0; i < h->nbuf; i++) {
    A(&h->X[i], B * (nm + 1) * xs * r); A(&h->S[i], B * nm * h->ncs * r); A(&h->Y[i], B * nm * h->ncs * r);
  }
  if (h->nbuf == kMaxBuf) A(&h->help, B * sizeof(HelpSlot));
  if (h->rpl <= 4 && h->bshare_mode != 0 && (h->nbuf == kMaxBuf || h->bshare_mode == 2)) {
    // 3 KB per knot (brec): in auto mode sweeps are only shared by batches of up to an eighth of the resident waves
    // (launch_iterate_t), so that is all the arrays are sized for; forced on (1 / 2) they cover the handle's whole batch
    const size_t auto_cap = h->sched_slots / 8 > 0 ? (size_t)h->sched_slots / 8 : 1;
    const size_t nb = (h->bshare_mode > 0 || B < auto_cap) ? B : auto_cap;
    h->bshare_cap = (int)nb;
    A(&h->bshare, nb * sizeof(BwdShare)); A(&h->bflag, nb * nm * sizeof(int)); A(&h->brec, nb * nm * (size_t)kRecDoubles * 8);
    A(&h->batch_dev, 16 * 1024);
    if (st == DIRECT_OK && hipHostMalloc((void**)&h->batch_pin, (size_t)16 * direct_ddp_handle_s::kBatchRing * 1024) != hipSuccess) {
      h->batch_pin = nullptr;
      st = fail(DIRECT_ERR_DEVICE, "pinned staging of the launch descriptors");
    }
  }
  A(&h->KU, B * nm * 100 * r); A(&h->KS, B * nm * h->ncs * r); A(&h->KY, B * nm * h->ncs * r);
  A(&h->st, B * sizeof(TrajState));
  A(&h->gbase, B * sizeof(GainBase));
  if (!packed) {
    A(&h->o.rtn, B * 4); A(&h->o.iter_used, B * 4); A(&h->o.fwd_passes, B * 4);
    A(&h->o.infeas_out, B); A(&h->o.line_failed_out, B);
    A(&h->o.cost, B * r); A(&h->o.costq, B * r); A(&h->o.jerk_cost, B * r); A(&h->o.terminal_norm2, B * r);
    A(&h->o.opterr, B * r); A(&h->o.mu, B * r);
    A(&h->o.bez, B * nm * 18 * r); A(&h->o.poly, B * nm * 18 * r); A(&h->o.T, B * nm * r);
  }
  A(&h->best_idx, 16); A(&h->best_cost, 16);
  A(&h->sched, (B + 2) * sizeof(int));
  A(&h->visits, 4 * sizeof(unsigned long long));  // [0] backward knots, [1] forward trial-knots, [2] knots whose front half a helper computed, [3] accepted line searches
  A(&h->live, 16 * sizeof(int));
  A(&h->nwaves, 16 * sizeof(int));
  A(&h->sched_mark, 2 * 8192 * sizeof(int));
  A(&h->sched_dbg, 64 * sizeof(int));
  if (st == DIRECT_OK && hipMemset(h->sched_dbg, 0, 64 * sizeof(int)) != hipSuccess) st = fail(DIRECT_ERR_DEVICE, "hipMemset");
  A(&h->tickets, 16 * sizeof(int));
  A(&h->order, B * sizeof(int32_t)); A(&h->cls_dev, B * sizeof(int32_t));
  if (const char* ev = getenv("DIRECT_DDP_CHUNK")) h->sched_chunk = atoi(ev) > 0 ? atoi(ev) : 1;
  if (const char* ev = getenv("DIRECT_DDP_PRIO")) h->sched_prio = atoi(ev);
  if (const char* ev = getenv("DIRECT_DDP_TAIL")) h->sched_tail = std::max(0, std::min(atoi(ev), 64));
  if (const char* ev = getenv("DIRECT_DDP_PAIR")) h->pair_trials = atoi(ev);
  h->fieldbuf_bytes = B * nm * (size_t)std::max(ncm, 100) * r + B * 16 * r + B * 9 * r;
  A(&h->fieldbuf, h->fieldbuf_bytes);
  if (st == DIRECT_OK && (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess ||
                          hipEventCreate(&h->ev2) != hipSuccess || hipEventCreate(&h->ev3) != hipSuccess))
    st = fail(DIRECT_ERR_DEVICE, "hipEventCreate failed");
  if (st == DIRECT_OK && hipMemset(h->sched, 0, (B + 2) * sizeof(int)) != hipSuccess)
    st = fail(DIRECT_ERR_DEVICE, "hipMemset failed");
  if (st != DIRECT_OK) {
    direct_ddp_destroy(h);
    return st;
  }
  *out = h;
  return DIRECT_OK;
}

direct_status_t direct_ddp_destroy(direct_ddp_handle_t h) {
  if (!h) return DIRECT_OK;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  for (void* p : h->allocs) (void)hipFree(p);
  for (char* q : {h->in_host, h->out_host, h->out0_host, h->batch_pin})
    if (q) (void)hipHostFree(q);
  for (auto& row : h->batch_ev)
    for (hipEvent_t e : row)
      if (e) (void)hipEventDestroy(e);
  if (h->in_ev) (void)hipEventDestroy(h->in_ev);
  if (h->filt) (void)hipFree(h->filt);
  for (void* q : {h->g_recs, h->g_blocks, h->g_win, h->g_in})
    if (q) (void)hipFree(q);
  for (int i = 0; i < 12; i++) {
    if (h->cstream[i]) { (void)hipStreamSynchronize(h->cstream[i]); (void)hipStreamDestroy(h->cstream[i]); }
    if (h->cev[i]) (void)hipEventDestroy(h->cev[i]);
  }
  if (h->fork_ev) (void)hipEventDestroy(h->fork_ev);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev2) (void)hipEventDestroy(h->ev2);
  if (h->ev3) (void)hipEventDestroy(h->ev3);
  delete h;
  return DIRECT_OK;
}

direct_status_t direct_ddp_set_stream(direct_ddp_handle_t h, void* hip_stream) {
  if (!h) return fail(DIRECT_ERR_INVALID, "null handle");
  h->stream = (hipStream_t)hip_stream;
  return DIRECT_OK;
}

static direct_status_t check_params(const direct_ddp_params_t* p) {
  if (!p) return fail(DIRECT_ERR_INVALID, "null params");
  if (p->time_power != 1 && p->time_power != 2)
    return fail(DIRECT_ERR_INVALID, "time_power must be 1 or 2 (computeq has no other branch, ddp_optimizer.cpp:1294-1305)");
  if (p->iter_max < 0) return fail(DIRECT_ERR_INVALID, "iter_max < 0");
  return DIRECT_OK;
}

// Row slots per lane the widest polytope of every trajectory needs (device-resident inputs: the host never sees them).
// Sizes outside the handle's configuration are clamped: begin() turns such rows into DIRECT_RTN_INVALID whatever
// kernels they are sent to.
__global__ void k_classify(int B, int nmax, int pmax, const int32_t* n_seg, const int32_t* n_planes, int32_t* slots) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int N = n_seg[b];
  N = N < 0 ? 0 : (N > nmax ? nmax : N);
  int pm = 1;
  for (int k = 0; k < N; k++) {
    const int p = n_planes[(size_t)b * nmax + k];
    pm = p > pm ? p : pm;
  }
  pm = pm > pmax ? pmax : pm;
  slots[b] = (6 * pm + 55 + 63) / 64;
}
// Sorts the trajectories of the current batch into row-slot classes (direct_ddp_handle_s::classes).  Handles whose
// p_max fits the two-slot kernels have one class by construction and skip all of this.  n_seg / n_planes: host arrays
// (host-memory inputs) or device arrays (one small read-back).
static direct_status_t classify_batch(direct_ddp_handle_t h, const direct_ddp_batch_in_t* in, bool on_host) {
  h->classes.clear();
  // Classes would buy OCCUPANCY (three waves per SIMD instead of one for the trajectories whose polytopes allow it) - but
  // every class launch is bound by the chain of ITS longest solve, and the class launches, each a grid of persistent
  // waves sized for the whole device, end up one after the other.  Measured on real-corridor replay plans (N <= 21, widest
  // polytope per plan 12 .. 59 planes, six classes; profiles/r04_real_corridors.json): 710 plans 36 ms on the
  // one class of the widest polytope against 65 ms in classes; 5680 plans (5.5 x the resident waves of that class)
  // 38.5 against 68.7 ms.  What pays is skipping, per knot, the row slots its polytope does not fill (ddp_wave.h,
  // slot_on) on the wide kernels.  So ONE class is the default; DIRECT_DDP_CLASSES=1 sorts the batch into classes
  // (bit-identical results: tests/test_gpu_real_corridors.py), for workloads of long trajectories far beyond the
  // resident waves, which none of the measured ones is.
  const char* force = getenv("DIRECT_DDP_CLASSES");
  if (h->rpl <= 2 || !(force && atoi(force) != 0)) return DIRECT_OK;
  const int B = in->batch, nm = h->nmax;
  std::vector<int32_t> sl(B);
  if (on_host) {
    for (int b = 0; b < B; b++) {
      int pm = 1;
      const int N = std::min(std::max((int)in->n_seg[b], 0), nm);  // as k_classify: a bad n_seg is begin()'s to report, not ours to follow
      for (int k = 0; k < N; k++) pm = std::max(pm, (int)in->n_planes[(size_t)b * nm + k]);
      sl[b] = (6 * std::min(pm, h->pmax) + 55 + 63) / 64;
    }
  } else {
    hipLaunchKernelGGL(k_classify, dim3((B + 255) / 256), dim3(256), 0, h->stream, B, nm, h->pmax, in->n_seg, in->n_planes, h->cls_dev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(sl.data(), h->cls_dev, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  int cnt[16] = {0};
  for (int b = 0; b < B; b++) {
    sl[b] = rpl_class(sl[b]);
    cnt[sl[b]]++;
  }
  int n_cls = 0;
  for (int r = 0; r < 16; r++) n_cls += cnt[r] > 0;
  if (n_cls == 1 && cnt[h->rpl] == B) return DIRECT_OK;  // everything on the handle's own class: identity order
  std::vector<int32_t> order(B);
  int off[16];
  for (int r = 0, o = 0; r < 16; r++) {
    off[r] = o;
    if (cnt[r] > 0) h->classes.push_back(direct_ddp_handle_s::Cls{r, o, cnt[r]});
    o += cnt[r];
  }
  for (int b = 0; b < B; b++) order[off[sl[b]]++] = b;  // stable: ascending trajectory numbers inside a class
  HIP_TRY(hipMemcpyAsync(h->order, order.data(), (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));  // `order` is a local
  return DIRECT_OK;
}

// copy (or alias) the inputs into device memory and fill h->cur_in with device pointers
static direct_status_t stage_inputs(direct_ddp_handle_t h, const direct_ddp_params_t* p, const direct_ddp_batch_in_t* in,
                                    bool fresh = true) {
  if (!h || !in) return fail(DIRECT_ERR_INVALID, "null argument");
  TRY(check_params(p));
  if (in->batch <= 0 || in->batch > h->max_batch) return fail(DIRECT_ERR_INVALID, "batch exceeds the handle's max_batch");
  if (in->n_seg_max != h->nmax || in->p_max != h->pmax)
    return fail(DIRECT_ERR_INVALID, "n_seg_max / p_max differ from the handle's configuration");
  if (!in->n_seg || !in->x0 || !in->xd || !in->n_planes || !in->planes)
    return fail(DIRECT_ERR_INVALID, "null input array");
  if (!in->T0 && !in->seeds)
    return fail(DIRECT_ERR_INVALID, "T0 == NULL asks for initTimeAllocation on the device: that needs the polytope seeds");
  if (p->line_init && !p->zero_init && !in->seeds)
    return fail(DIRECT_ERR_INVALID, "line_init needs the polytope seeds (direct_ddp_batch_in_t.seeds)");
  if (!p->zero_init && !p->line_init && !in->init_bez && !in->init_poly)
    return fail(DIRECT_ERR_INVALID, "init_bez (or init_poly) required unless zero_init");
  HIP_TRY(hipSetDevice(h->device));
  const size_t B = in->batch, nm = h->nmax, r = h->rsz;
  direct_ddp_batch_in_t d = *in;
  if (in->mem == DIRECT_MEM_HOST) {
    // validate sizes on the host before anything reaches the device
    for (size_t b = 0; b < B; b++) {
      if (in->n_seg[b] < 1 || in->n_seg[b] > (int)nm) return fail(DIRECT_ERR_INVALID, "n_seg out of range");
      for (int k = 0; k < in->n_seg[b]; k++) {
        int np = in->n_planes[b * nm + k];
        if (np < 1 || np > h->pmax) return fail(DIRECT_ERR_INVALID, "n_planes out of range");
      }
    }
    const bool packed = h->in_blob != nullptr;
    if (packed && h->in_ev_pending) {  // the mirror's previous upload (a call whose outputs stayed on the device never synchronised)
      HIP_TRY(hipEventSynchronize(h->in_ev));
      h->in_ev_pending = false;
    }
    // packed: the array goes into the pinned mirror at its slice's offset; ONE copy of the whole mirror follows
    auto up = [&](void* dst, const void* src, size_t bytes) {
      if (!packed) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream);
      memcpy(h->in_host + ((char*)dst - h->in_blob), src, bytes);
      return hipSuccess;
    };
    HIP_TRY(up(h->n_seg, in->n_seg, B * 4)); d.n_seg = h->n_seg;
    HIP_TRY(up(h->x0, in->x0, B * 9 * r)); d.x0 = h->x0;
    HIP_TRY(up(h->xd, in->xd, B * 9 * r)); d.xd = h->xd;
    if (in->T0) { HIP_TRY(up(h->T0, in->T0, B * nm * r)); d.T0 = h->T0; }
    HIP_TRY(up(h->n_planes, in->n_planes, B * nm * 4)); d.n_planes = h->n_planes;
    HIP_TRY(up(h->planes, in->planes, B * nm * h->pmax * 4 * r)); d.planes = h->planes;
    if (in->init_bez) { HIP_TRY(up(h->init_bez, in->init_bez, B * nm * 18 * r)); d.init_bez = h->init_bez; }
    if (in->init_poly) { HIP_TRY(up(h->init_poly, in->init_poly, B * nm * 18 * r)); d.init_poly = h->init_poly; }
    if (in->seeds) { HIP_TRY(up(h->seeds, in->seeds, B * nm * 3 * r)); d.seeds = h->seeds; }
    if (in->infeas_in) { HIP_TRY(up(h->infeas_in, in->infeas_in, B)); d.infeas_in = h->infeas_in; }
    if (packed) {
      HIP_TRY(hipMemcpyAsync(h->in_blob, h->in_host, h->in_blob_bytes, hipMemcpyHostToDevice, h->stream));
      HIP_TRY(hipEventRecord(h->in_ev, h->stream));
      h->in_ev_pending = true;
    }
  }
  if (!in->infeas_in) {
    HIP_TRY(hipMemsetAsync(h->infeas_in, p->infeas ? 1 : 0, B, h->stream));
    d.infeas_in = h->infeas_in;
  }
  if (!in->T0) {  // initTimeAllocation on the device, from the (device-resident) start / goal positions and seeds
    const int n = (int)(B * nm);
    if (h->dtype == DIRECT_F64)
      hipLaunchKernelGGL(k_time_alloc<double>, dim3((n + 255) / 256), dim3(256), 0, h->stream, (int)B, (int)nm, d.n_seg, (const double*)d.x0,
                         (const double*)d.xd, (const double*)d.seeds, p->max_vel, p->max_acc, (double*)h->T0);
    else
      hipLaunchKernelGGL(k_time_alloc<float>, dim3((n + 255) / 256), dim3(256), 0, h->stream, (int)B, (int)nm, d.n_seg, (const float*)d.x0,
                         (const float*)d.xd, (const float*)d.seeds, p->max_vel, p->max_acc, (float*)h->T0);
    HIP_TRY(hipGetLastError());
    d.T0 = h->T0;
  }
  if (fresh) {  // (phase 1 of a plan keeps phase 0's classes: the same polytopes)
    h->B = in->batch;
    TRY(classify_batch(h, in->mem == DIRECT_MEM_HOST ? in : &d, in->mem == DIRECT_MEM_HOST));
  }
  // the scheduler-error flag is sticky within one API call (both phases of a plan) and cleared between calls
  if (fresh) HIP_TRY(hipMemsetAsync(h->sched + 1, 0, sizeof(int), h->stream));
  d.mem = DIRECT_MEM_DEVICE;
  h->cur_in = d;
  h->params = *p;
  h->B = in->batch;
  // filter capacity follows iter_max
  if (p->iter_max + 4 > h->fcap) {
    if (h->filt) { HIP_TRY(hipStreamSynchronize(h->stream)); HIP_TRY(hipFree(h->filt)); h->filt = nullptr; }
    h->fcap = p->iter_max + 4;
    HIP_TRY(hipMalloc((void**)&h->filt, (size_t)h->max_batch * h->fcap * 2 * sizeof(double)));
  }
  return DIRECT_OK;
}

static direct_status_t launch_begin(direct_ddp_handle_t h) {
  if (h->dtype == DIRECT_F64) launch_begin_t<double>(h);
  else launch_begin_t<float>(h);
  HIP_TRY(hipGetLastError());
  h->begun = true;
  return DIRECT_OK;
}

static direct_status_t launch_iterate(direct_ddp_handle_t h, int n, int mode) {
  if (!h->begun) return fail(DIRECT_ERR_INVALID, "direct_ddp_begin has not been called");
  HIP_TRY(hipEventRecord(h->ev0, h->stream));
  if (h->dtype == DIRECT_F64) launch_iterate_t<double>(h, n, mode);
  else launch_iterate_t<float>(h, n, mode);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(h->ev1, h->stream));
  h->n_launches = 1;
  h->timed = true;
  return DIRECT_OK;
}

static direct_status_t launch_finish(direct_ddp_handle_t h, direct_ddp_batch_out_t* out) {
  if (!h->begun) return fail(DIRECT_ERR_INVALID, "direct_ddp_begin has not been called");
  if (!out) return fail(DIRECT_ERR_INVALID, "null out");
  if (h->dtype == DIRECT_F64) launch_finish_t<double>(h, out);
  else launch_finish_t<float>(h, out);
  HIP_TRY(hipGetLastError());
  if (out->mem == DIRECT_MEM_HOST) {
    const size_t B = h->B, nm = h->nmax, r = h->rsz;
    const bool packed = h->out_blob != nullptr;
    int sched_err_p = 0;
    if (packed) {  // one copy of every result array into the pinned mirror; the caller's arrays are filled from it
      HIP_TRY(hipMemcpyAsync(h->out_host, h->out_blob, h->out_blob_bytes, hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(hipMemcpyAsync(&sched_err_p, h->sched + 1, sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(hipStreamSynchronize(h->stream));
      h->in_ev_pending = false;
    }
    auto dn = [&](void* dst, const void* src, size_t bytes) {
      if (!dst) return hipSuccess;
      if (packed) {
        memcpy(dst, h->out_host + ((const char*)src - h->out_blob), bytes);
        return hipSuccess;
      }
      return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream);
    };
    HIP_TRY(dn(out->rtn, h->o.rtn, B * 4)); HIP_TRY(dn(out->iter_used, h->o.iter_used, B * 4));
    HIP_TRY(dn(out->fwd_passes, h->o.fwd_passes, B * 4));
    HIP_TRY(dn(out->infeas_out, h->o.infeas_out, B)); HIP_TRY(dn(out->line_failed_out, h->o.line_failed_out, B));
    HIP_TRY(dn(out->cost, h->o.cost, B * r)); HIP_TRY(dn(out->costq, h->o.costq, B * r));
    HIP_TRY(dn(out->jerk_cost, h->o.jerk_cost, B * r)); HIP_TRY(dn(out->terminal_norm2, h->o.terminal_norm2, B * r));
    HIP_TRY(dn(out->opterr, h->o.opterr, B * r)); HIP_TRY(dn(out->mu, h->o.mu, B * r));
    HIP_TRY(dn(out->bez, h->o.bez, B * nm * 18 * r)); HIP_TRY(dn(out->poly, h->o.poly, B * nm * 18 * r));
    HIP_TRY(dn(out->T, h->o.T, B * nm * r));
    int sched_err = sched_err_p;
    if (!packed) {
      HIP_TRY(hipMemcpyAsync(&sched_err, h->sched + 1, sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(hipStreamSynchronize(h->stream));
      h->in_ev_pending = false;
    }
    if (sched_err) return fail(DIRECT_ERR_DEVICE, "ticket scheduler of k_iterate hit its spin limit; results are incomplete");
  }
  return DIRECT_OK;
}

direct_status_t direct_ddp_begin(direct_ddp_handle_t h, const direct_ddp_params_t* params, const direct_ddp_batch_in_t* in) {
  TRY(stage_inputs(h, params, in));
  return launch_begin(h);
}
direct_status_t direct_ddp_backward_pass(direct_ddp_handle_t h) {
  if (!h) return fail(DIRECT_ERR_INVALID, "null handle");
  return launch_iterate(h, 1, 1);
}
direct_status_t direct_ddp_forward_pass(direct_ddp_handle_t h) {
  if (!h) return fail(DIRECT_ERR_INVALID, "null handle");
  return launch_iterate(h, 1, 2);
}
direct_status_t direct_ddp_forward_pass_stored(direct_ddp_handle_t h) {
  if (!h) return fail(DIRECT_ERR_INVALID, "null handle");
  return launch_iterate(h, 1, 3);
}
direct_status_t direct_ddp_iterate(direct_ddp_handle_t h, int32_t n_iters) {
  if (!h) return fail(DIRECT_ERR_INVALID, "null handle");
  return launch_iterate(h, n_iters, 0);
}
direct_status_t direct_ddp_finish(direct_ddp_handle_t h, direct_ddp_batch_out_t* out) {
  if (!h) return fail(DIRECT_ERR_INVALID, "null handle");
  return launch_finish(h, out);
}

direct_status_t direct_ddp_solve_batch(direct_ddp_handle_t h, const direct_ddp_params_t* params,
                                       const direct_ddp_batch_in_t* in, direct_ddp_batch_out_t* out) {
  if (!out) return fail(DIRECT_ERR_INVALID, "null out");
  TRY(stage_inputs(h, params, in));
  TRY(launch_begin(h));
  TRY(launch_iterate(h, params->iter_max, 0));
  return launch_finish(h, out);
}

direct_status_t direct_ddp_plan_batch(direct_ddp_handle_t h, const direct_ddp_params_t* p0, const direct_ddp_params_t* p1,
                                      const direct_ddp_batch_in_t* in, direct_ddp_batch_out_t* out0,
                                      direct_ddp_batch_out_t* out1) {
  if (!h || !out1 || !in || !p0 || !p1) return fail(DIRECT_ERR_INVALID, "null argument");
  TRY(check_params(p0));
  TRY(check_params(p1));
  // phase 0 (teach_repeat_planner.cpp:886-897): infeas = true for every problem
  direct_ddp_batch_in_t in0 = *in;
  in0.infeas_in = nullptr;
  direct_ddp_params_t q0 = *p0;
  q0.infeas = 1;
  TRY(stage_inputs(h, &q0, &in0));
  TRY(launch_begin(h));
  TRY(launch_iterate(h, q0.iter_max, 0));
  // results of phase 0 stay on the device (staging buffers of the handle)
  direct_ddp_batch_out_t stage = {};
  stage.mem = DIRECT_MEM_DEVICE;
  stage.rtn = h->o.rtn; stage.iter_used = h->o.iter_used; stage.fwd_passes = h->o.fwd_passes;
  stage.infeas_out = h->o.infeas_out; stage.line_failed_out = h->o.line_failed_out;
  stage.cost = h->o.cost; stage.costq = h->o.costq; stage.jerk_cost = h->o.jerk_cost;
  stage.terminal_norm2 = h->o.terminal_norm2; stage.opterr = h->o.opterr; stage.mu = h->o.mu;
  stage.bez = h->o.bez; stage.poly = h->o.poly; stage.T = h->o.T;
  TRY(launch_finish(h, &stage));
  const bool out0_packed = out0 && out0->mem == DIRECT_MEM_HOST && h->out_blob != nullptr;
  if (out0_packed) {  // one copy into the second pinned mirror now; the caller's arrays are filled after phase 1 has synchronised
    HIP_TRY(hipMemcpyAsync(h->out0_host, h->out_blob, h->out_blob_bytes, hipMemcpyDeviceToHost, h->stream));
  } else if (out0) {  // hand phase-0 results to the caller as well
    const size_t B = h->B, nm = h->nmax, r = h->rsz;
    hipMemcpyKind kind = out0->mem == DIRECT_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    auto cp = [&](void* dst, const void* src, size_t bytes) { return dst ? hipMemcpyAsync(dst, src, bytes, kind, h->stream) : hipSuccess; };
    HIP_TRY(cp(out0->rtn, h->o.rtn, B * 4)); HIP_TRY(cp(out0->iter_used, h->o.iter_used, B * 4));
    HIP_TRY(cp(out0->fwd_passes, h->o.fwd_passes, B * 4)); HIP_TRY(cp(out0->infeas_out, h->o.infeas_out, B));
    HIP_TRY(cp(out0->line_failed_out, h->o.line_failed_out, B));
    HIP_TRY(cp(out0->cost, h->o.cost, B * r)); HIP_TRY(cp(out0->costq, h->o.costq, B * r));
    HIP_TRY(cp(out0->jerk_cost, h->o.jerk_cost, B * r)); HIP_TRY(cp(out0->terminal_norm2, h->o.terminal_norm2, B * r));
    HIP_TRY(cp(out0->opterr, h->o.opterr, B * r)); HIP_TRY(cp(out0->mu, h->o.mu, B * r));
    HIP_TRY(cp(out0->bez, h->o.bez, B * nm * 18 * r)); HIP_TRY(cp(out0->poly, h->o.poly, B * nm * 18 * r));
    HIP_TRY(cp(out0->T, h->o.T, B * nm * r));
  }
  // UpdateTime where rtn0 == 2, warm start from the phase-0 Bezier coefficients (TRP:911-918)
  const int n = h->B * h->nmax;
  // (TRP:918 hands over Bezier control points; the same warm start goes over as monomial coefficients, see k_chain)
  if (h->dtype == DIRECT_F64)
    hipLaunchKernelGGL(k_chain<double>, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->B, h->nmax, h->o.rtn,
                       (const double*)h->o.T, (const double*)h->cur_in.T0, h->o.infeas_out, (const double*)h->o.poly,
                       (double*)h->T_next, h->infeas_next, (double*)h->init_poly);
  else
    hipLaunchKernelGGL(k_chain<float>, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->B, h->nmax, h->o.rtn,
                       (const float*)h->o.T, (const float*)h->cur_in.T0, h->o.infeas_out, (const float*)h->o.poly,
                       (float*)h->T_next, h->infeas_next, (float*)h->init_poly);
  HIP_TRY(hipGetLastError());
  direct_ddp_batch_in_t in1 = h->cur_in;  // device pointers
  in1.T0 = h->T_next;
  in1.init_bez = nullptr;
  in1.init_poly = h->init_poly;
  in1.infeas_in = h->infeas_next;
  TRY(stage_inputs(h, p1, &in1, false));
  TRY(launch_begin(h));
  TRY(launch_iterate(h, p1->iter_max, 0));
  const direct_status_t fin = launch_finish(h, out1);
  if (out0_packed) {
    // (host results of phase 1 have synchronised already - unless that call failed before its synchronisation: the copy
    // into out0_host may then still be in flight)
    if (out1->mem != DIRECT_MEM_HOST || fin != DIRECT_OK) HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t B = h->B, nm = h->nmax, r = h->rsz;
    auto sc = [&](void* dst, const void* src, size_t bytes) {
      if (dst) memcpy(dst, h->out0_host + ((const char*)src - h->out_blob), bytes);
    };
    sc(out0->rtn, h->o.rtn, B * 4); sc(out0->iter_used, h->o.iter_used, B * 4); sc(out0->fwd_passes, h->o.fwd_passes, B * 4);
    sc(out0->infeas_out, h->o.infeas_out, B); sc(out0->line_failed_out, h->o.line_failed_out, B);
    sc(out0->cost, h->o.cost, B * r); sc(out0->costq, h->o.costq, B * r); sc(out0->jerk_cost, h->o.jerk_cost, B * r);
    sc(out0->terminal_norm2, h->o.terminal_norm2, B * r); sc(out0->opterr, h->o.opterr, B * r); sc(out0->mu, h->o.mu, B * r);
    sc(out0->bez, h->o.bez, B * nm * 18 * r); sc(out0->poly, h->o.poly, B * nm * 18 * r); sc(out0->T, h->o.T, B * nm * r);
  }
  return fin;
}

static size_t field_elems(direct_ddp_handle_t h, int field) {
  const size_t B = h->B, nm = h->nmax, ncm = 6 * h->pmax + 55;
  switch (field) {
    case DIRECT_FIELD_X: return B * (nm + 1) * 9;
    case DIRECT_FIELD_U: case DIRECT_FIELD_KU: return B * nm * 10;
    case DIRECT_FIELD_KUU: return B * nm * 90;
    case DIRECT_FIELD_S: case DIRECT_FIELD_Y: case DIRECT_FIELD_C: case DIRECT_FIELD_KS: case DIRECT_FIELD_KY: return B * nm * ncm;
    case DIRECT_FIELD_SCALARS: return B * 16;
    default: return 0;
  }
}

direct_status_t direct_ddp_get_field(direct_ddp_handle_t h, int32_t field, void* dst) {
  if (!h || !dst) return fail(DIRECT_ERR_INVALID, "null argument");
  if (!h->begun) return fail(DIRECT_ERR_INVALID, "direct_ddp_begin has not been called");
  size_t n = field_elems(h, field);
  if (!n) return fail(DIRECT_ERR_INVALID, "unknown field");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemsetAsync(h->fieldbuf, 0, n * h->rsz, h->stream));
  if (h->dtype == DIRECT_F64) launch_field_t<double>(h, field, 0);
  else launch_field_t<float>(h, field, 0);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(dst, h->fieldbuf, n * h->rsz, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return DIRECT_OK;
}

direct_status_t direct_ddp_set_field(direct_ddp_handle_t h, int32_t field, const void* src) {
  if (!h || !src) return fail(DIRECT_ERR_INVALID, "null argument");
  if (!h->begun) return fail(DIRECT_ERR_INVALID, "direct_ddp_begin has not been called");
  if (field != DIRECT_FIELD_X && field != DIRECT_FIELD_U && field != DIRECT_FIELD_S && field != DIRECT_FIELD_Y)
    return fail(DIRECT_ERR_INVALID, "only X, U, S, Y can be set");
  size_t n = field_elems(h, field);
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemcpyAsync(h->fieldbuf, src, n * h->rsz, hipMemcpyHostToDevice, h->stream));
  if (h->dtype == DIRECT_F64) launch_field_t<double>(h, field, 1);
  else launch_field_t<float>(h, field, 1);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return DIRECT_OK;
}

direct_status_t direct_ddp_last_kernel_ms(direct_ddp_handle_t h, double* ms, int32_t* n_launches) {
  if (!h || !ms) return fail(DIRECT_ERR_INVALID, "null argument");
  if (!h->timed) return fail(DIRECT_ERR_INVALID, "nothing has been timed yet");
  HIP_TRY(hipEventSynchronize(h->ev1));
  float t = 0.f;
  HIP_TRY(hipEventElapsedTime(&t, h->ev0, h->ev1));
  *ms = (double)t;
  if (n_launches) *n_launches = h->n_launches;
  return DIRECT_OK;
}

direct_status_t direct_ddp_last_launch_info(direct_ddp_handle_t h, direct_ddp_launch_info_t* info) {
  if (!h || !info) return fail(DIRECT_ERR_INVALID, "null argument");
  if (!h->timed) return fail(DIRECT_ERR_INVALID, "no hot-kernel launch yet");
  HIP_TRY(hipSetDevice(h->device));
  unsigned long long v[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(v, h->visits, sizeof v, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  *info = h->last_info;
  info->bwd_knot_visits = v[0];
  info->fwd_knot_visits = v[1];
  return DIRECT_OK;
}

#if defined(DDP_TIMELINE)  // debug builds only (tools/timeline.py)
direct_status_t direct_ddp_debug_timeline(direct_ddp_handle_t h, void* dst) {
  if (!g_tl) return fail(DIRECT_ERR_INVALID, "no timeline");
  HIP_TRY(hipMemcpy(dst, g_tl, (size_t)h->B * kTimelineDepth * 4 * 8, hipMemcpyDeviceToHost));
  return DIRECT_OK;
}
#endif

direct_status_t direct_ddp_last_counters(direct_ddp_handle_t h, uint64_t* out4) {
  if (!h || !out4) return fail(DIRECT_ERR_INVALID, "null argument");
  if (!h->timed) return fail(DIRECT_ERR_INVALID, "no hot-kernel launch yet");
  HIP_TRY(hipSetDevice(h->device));
  unsigned long long v[4] = {0, 0, 0, 0};
  HIP_TRY(hipMemcpyAsync(v, h->visits, sizeof v, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int i = 0; i < 4; i++) out4[i] = v[i];
  return DIRECT_OK;
}

direct_status_t direct_ddp_sched_debug(direct_ddp_handle_t h, int32_t* out64) {
  if (!h || !out64) return fail(DIRECT_ERR_INVALID, "null argument");
  HIP_TRY(hipMemcpyAsync(out64, h->sched_dbg, 64 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return DIRECT_OK;
}
direct_status_t direct_ddp_sched_error(direct_ddp_handle_t h, int32_t* flag) {
  if (!h || !flag) return fail(DIRECT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  int v = 0;
  HIP_TRY(hipMemcpyAsync(&v, h->sched + 1, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  *flag = v;
  return DIRECT_OK;
}

direct_status_t direct_ddp_best_cost(direct_ddp_handle_t h, int32_t mem, const void* cost, const int32_t* rtn,
                                     int32_t batch, int32_t* best_index, double* best_cost) {
  if (!h || !cost || !rtn || !best_index || !best_cost || batch <= 0 || batch > h->max_batch)
    return fail(DIRECT_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(h->device));
  const void* dc = cost;
  const int32_t* dr = rtn;
  if (mem == DIRECT_MEM_HOST) {
    HIP_TRY(hipMemcpyAsync(h->o.cost, cost, (size_t)batch * h->rsz, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->o.rtn, rtn, (size_t)batch * 4, hipMemcpyHostToDevice, h->stream));
    dc = h->o.cost;
    dr = h->o.rtn;
  }
  if (h->dtype == DIRECT_F64)
    hipLaunchKernelGGL(k_best<double>, dim3(1), dim3(256), 0, h->stream, (const double*)dc, dr, batch, h->best_idx, h->best_cost);
  else
    hipLaunchKernelGGL(k_best<float>, dim3(1), dim3(256), 0, h->stream, (const float*)dc, dr, batch, h->best_idx, h->best_cost);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(best_index, h->best_idx, 4, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(best_cost, h->best_cost, 8, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return DIRECT_OK;
}

#define NCCL_TRY(expr)                                                                             \
  do {                                                                                             \
    ncclResult_t r_ = (expr);                                                                      \
    if (r_ != ncclSuccess)                                                                         \
      return fail(DIRECT_ERR_DEVICE, std::string(#expr) + ": " + rccl_api().GetErrorString(r_));   \
  } while (0)

direct_status_t direct_rccl_unique_id(direct_rccl_id_t* id) {
  static_assert(sizeof(direct_rccl_id_t) == sizeof(ncclUniqueId), "direct_rccl_id_t must mirror ncclUniqueId");
  if (!id) return fail(DIRECT_ERR_INVALID, "null argument");
  if (!rccl_api().ok) return fail(DIRECT_ERR_UNSUPPORTED, "librccl.so.1 not found");
  NCCL_TRY(rccl_api().GetUniqueId((ncclUniqueId*)id));
  return DIRECT_OK;
}

direct_status_t direct_rccl_comm_create(direct_ddp_handle_t h, const direct_rccl_id_t* id, int32_t n_ranks, int32_t rank,
                                        void** comm) {
  if (!h || !id || !comm || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(DIRECT_ERR_INVALID, "bad argument");
  if (!rccl_api().ok) return fail(DIRECT_ERR_UNSUPPORTED, "librccl.so.1 not found");
  HIP_TRY(hipSetDevice(h->device));
  ncclComm_t c = nullptr;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  NCCL_TRY(rccl_api().CommInitRank(&c, n_ranks, uid, rank));
  *comm = (void*)c;
  return DIRECT_OK;
}

direct_status_t direct_rccl_comm_destroy(void* comm) {
  if (!comm) return DIRECT_OK;
  if (!rccl_api().ok) return fail(DIRECT_ERR_UNSUPPORTED, "librccl.so.1 not found");
  NCCL_TRY(rccl_api().CommDestroy((ncclComm_t)comm));
  return DIRECT_OK;
}

extern "C++" {
template <typename Real>
static direct_status_t gather_best_t(direct_ddp_handle_t h, ncclComm_t comm, int n_ranks, int rank, const Real* cost,
                                     const int32_t* rtn, const Real* bez, const Real* T, int batch, long long first,
                                     Real* out_bez, Real* out_T) {
  const int nm = h->nmax;
  const size_t blk = (size_t)nm * 19;
  hipLaunchKernelGGL(k_best<Real>, dim3(1), dim3(256), 0, h->stream, cost, rtn, batch, h->best_idx, h->best_cost);
  hipLaunchKernelGGL(k_pack_best<Real>, dim3(8), dim3(256), 0, h->stream, h->best_idx, h->best_cost, first, rank, nm, bez, T,
                     (BestRec*)h->g_recs, (Real*)h->g_blocks);
  HIP_TRY(hipGetLastError());
  // in-place all-gathers: every rank's send buffer is its own slot of the receive buffer
  NCCL_TRY(rccl_api().AllGather((char*)h->g_recs + (size_t)rank * sizeof(BestRec), h->g_recs, sizeof(BestRec), ncclChar, comm,
                                h->stream));
  NCCL_TRY(rccl_api().AllGather((Real*)h->g_blocks + (size_t)rank * blk, h->g_blocks, blk * sizeof(Real), ncclChar, comm,
                                h->stream));
  hipLaunchKernelGGL(k_pick_best<Real>, dim3(8), dim3(256), 0, h->stream, (const BestRec*)h->g_recs, (const Real*)h->g_blocks,
                     n_ranks, nm, (BestRec*)h->g_win, (int*)((char*)h->g_win + sizeof(BestRec)), out_bez, out_T);
  HIP_TRY(hipGetLastError());
  return DIRECT_OK;
}
}  // extern "C++"

direct_status_t direct_ddp_gather_best(direct_ddp_handle_t h, void* nccl_comm, int32_t n_ranks, int32_t rank, int32_t mem,
                                       const void* cost, const int32_t* rtn, const void* bez, const void* T, int32_t batch,
                                       int64_t first_index, int64_t* best_index, double* best_cost, int32_t* owner_rank,
                                       void* best_bez, void* best_T) {
  if (!h || !nccl_comm || !cost || !rtn || !bez || !T || !best_index || !best_cost || batch <= 0 || batch > h->max_batch ||
      n_ranks < 1 || rank < 0 || rank >= n_ranks)
    return fail(DIRECT_ERR_INVALID, "bad argument");
  if (!rccl_api().ok) return fail(DIRECT_ERR_UNSUPPORTED, "librccl.so.1 not found");
  HIP_TRY(hipSetDevice(h->device));
  const size_t nm = h->nmax, r = h->rsz, blk = nm * 19 * r;
  if (h->g_ranks < n_ranks) {  // (re)allocate the gather buffers for this communicator size
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (void** q : {&h->g_recs, &h->g_blocks, &h->g_win, &h->g_in})
      if (*q) { (void)hipFree(*q); *q = nullptr; }
    HIP_TRY(hipMalloc(&h->g_recs, (size_t)n_ranks * sizeof(BestRec)));
    HIP_TRY(hipMalloc(&h->g_blocks, (size_t)n_ranks * blk));
    HIP_TRY(hipMalloc(&h->g_win, sizeof(BestRec) + 16));
    HIP_TRY(hipMalloc(&h->g_in, (size_t)h->max_batch * (nm * 19 * r + r + 4) + blk + 256));  // + the 16-byte alignment of each staged array
    h->g_ranks = n_ranks;
  }
  const void *dc = cost, *db = bez, *dT = T;
  const int32_t* dr = rtn;
  void *ob = best_bez, *oT = best_T;
  if (mem == DIRECT_MEM_HOST) {  // stage the host arrays
    char* p = (char*)h->g_in;
    auto up = [&](const void* src, size_t bytes) -> const void* {
      void* d = p;
      p += (bytes + 15) / 16 * 16;
      return hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, h->stream) == hipSuccess ? d : nullptr;
    };
    dc = up(cost, (size_t)batch * r);
    dr = (const int32_t*)up(rtn, (size_t)batch * 4);
    db = up(bez, (size_t)batch * nm * 18 * r);
    dT = up(T, (size_t)batch * nm * r);
    if (!dc || !dr || !db || !dT) return fail(DIRECT_ERR_DEVICE, "staging copy failed");
    ob = best_bez ? (void*)p : nullptr;
    oT = best_T ? (void*)(p + nm * 18 * r) : nullptr;
  }
  if (h->dtype == DIRECT_F64)
    TRY(gather_best_t<double>(h, (ncclComm_t)nccl_comm, n_ranks, rank, (const double*)dc, dr, (const double*)db,
                              (const double*)dT, batch, (long long)first_index, (double*)ob, (double*)oT));
  else
    TRY(gather_best_t<float>(h, (ncclComm_t)nccl_comm, n_ranks, rank, (const float*)dc, dr, (const float*)db,
                             (const float*)dT, batch, (long long)first_index, (float*)ob, (float*)oT));
  struct { BestRec w; int owner; int pad; } res;
  HIP_TRY(hipMemcpyAsync(&res, h->g_win, sizeof(BestRec) + 8, hipMemcpyDeviceToHost, h->stream));
  if (mem == DIRECT_MEM_HOST) {
    if (best_bez) HIP_TRY(hipMemcpyAsync(best_bez, ob, nm * 18 * r, hipMemcpyDeviceToHost, h->stream));
    if (best_T) HIP_TRY(hipMemcpyAsync(best_T, oT, nm * r, hipMemcpyDeviceToHost, h->stream));
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  *best_index = res.w.index;
  *best_cost = res.w.cost;
  if (owner_rank) *owner_rank = res.owner;
  return DIRECT_OK;
}

#if defined(DDP_TIMING)
// debug builds only (not declared in the header): per-phase cycle totals of workgroup 0
direct_status_t direct_ddp_debug_phase_cycles(unsigned long long* out64, int reset) {
  if (reset) {
    unsigned long long z[64] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(direct::g_phase_cycles), z, sizeof z));
    return DIRECT_OK;
  }
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out64, HIP_SYMBOL(direct::g_phase_cycles), 64 * sizeof(unsigned long long)));
  return DIRECT_OK;
}
#endif

// initTimeAllocation (teach_repeat_planner.cpp:583-639) with v0 = 0: host-side, double precision.
direct_status_t direct_time_allocation(int32_t batch, int32_t n_seg_max, const int32_t* n_seg, const double* start,
                                       const double* goal, const double* seeds, double max_vel, double max_acc,
                                       double* T_out) {
  if (batch <= 0 || n_seg_max <= 0 || !n_seg || !start || !goal || !seeds || !T_out || max_vel <= 0 || max_acc <= 0)
    return fail(DIRECT_ERR_INVALID, "bad argument");
  const double acct = max_vel / max_acc, accd = max_acc * acct * acct / 2.0;
  const double dcct = max_vel / max_acc, dccd = max_acc * dcct * dcct / 2.0;
  for (int b = 0; b < batch; b++) {
    const int N = n_seg[b];
    if (N < 1 || N > n_seg_max) return fail(DIRECT_ERR_INVALID, "n_seg out of range");
    for (int k = 0; k < n_seg_max; k++) {
      double t = 0.0;
      if (k < N) {
        const double* p0 = (k == 0) ? start + (size_t)b * 3 : seeds + ((size_t)b * n_seg_max + k) * 3;
        const double* p1 = (k == N - 1) ? goal + (size_t)b * 3 : seeds + ((size_t)b * n_seg_max + k + 1) * 3;
        const double dx = p1[0] - p0[0], dy = p1[1] - p0[1], dz = p1[2] - p0[2];
        const double D = sqrt(dx * dx + dy * dy + dz * dz);
        if (D < accd + dccd) t = 2.0 * sqrt(max_acc * D) / max_acc;   // triangle profile: t2 + t3
        else t = acct + (D - accd - dccd) / max_vel + dcct;          // trapezoid profile
      }
      T_out[(size_t)b * n_seg_max + k] = t;
    }
  }
  return DIRECT_OK;
}

size_t direct_corridor_wire_size(int32_t n_seg, const int32_t* n_planes) {
  return (n_seg < 0 || !n_planes) ? 0 : corridor_wire_size(n_seg, n_planes);
}

direct_status_t direct_corridor_pack(int32_t path_id, int32_t n_seg, const int32_t* n_planes, const double* planes,
                                     int32_t p_max, const double* seeds, const double* centers, uint8_t* buf,
                                     size_t capacity, size_t* written) {
  if (n_seg < 0 || p_max <= 0 || !n_planes || !planes || !seeds || !centers || !buf)
    return fail(DIRECT_ERR_INVALID, "bad argument");
  for (int k = 0; k < n_seg; k++)
    if (n_planes[k] < 0 || n_planes[k] > p_max) return fail(DIRECT_ERR_INVALID, "n_planes out of range");
  if (corridor_wire_size(n_seg, n_planes) > capacity) return fail(DIRECT_ERR_INVALID, "buffer too small");
  const size_t n = corridor_pack(path_id, n_seg, n_planes, planes, p_max, seeds, centers, buf);
  if (written) *written = n;
  return DIRECT_OK;
}

direct_status_t direct_corridor_unpack(const uint8_t* buf, size_t len, int32_t n_seg_max, int32_t p_max,
                                       int32_t* path_id, int32_t* n_seg, int32_t* n_planes, double* planes,
                                       double* seeds, double* centers, size_t* used) {
  if (!buf || n_seg_max <= 0 || p_max <= 0 || !path_id || !n_seg || !n_planes || !planes || !seeds || !centers)
    return fail(DIRECT_ERR_INVALID, "bad argument");
  const int rc = corridor_unpack(buf, len, n_seg_max, p_max, path_id, n_seg, n_planes, planes, seeds, centers, used);
  if (rc == 1) return fail(DIRECT_ERR_INVALID, "truncated or malformed msgs/corridor buffer");
  if (rc == 2) return fail(DIRECT_ERR_UNSUPPORTED, "corridor exceeds n_seg_max polytopes or p_max facets");
  return DIRECT_OK;
}

direct_status_t direct_corridor_replay_batch(int32_t n_rec, const int32_t* n_planes_rec, const double* planes_rec,
                                             int32_t p_max, const double* seeds_rec, const double* centers_rec,
                                             int32_t n_first, int32_t batch, double max_vel, double max_acc,
                                             int32_t* n_seg, double* x0, double* xd, double* T0, int32_t* n_planes,
                                             double* planes, double* seeds_out) {
  if (n_rec <= 0 || p_max <= 0 || n_first < 1 || batch < 1 || !n_planes_rec || !planes_rec || !seeds_rec ||
      !centers_rec || !n_seg || !x0 || !xd || !T0 || !n_planes || !planes || !seeds_out)
    return fail(DIRECT_ERR_INVALID, "bad argument");
  const int nm = n_first + batch - 1;
  if (nm > n_rec) return fail(DIRECT_ERR_INVALID, "no enough recorded polyhedrons");  // TRP:802-804
  std::vector<double> start((size_t)batch * 3), goal((size_t)batch * 3);
  for (int b = 0; b < batch; b++) {
    const int n = n_first + b;
    n_seg[b] = n;
    for (int q = 0; q < 9; q++) x0[(size_t)b * 9 + q] = xd[(size_t)b * 9 + q] = 0.0;
    for (int d = 0; d < 3; d++) {  // TRP:810-811: _start_pt / _end_pt are polytope centers
      start[(size_t)b * 3 + d] = x0[(size_t)b * 9 + d] = centers_rec[d];
      goal[(size_t)b * 3 + d] = xd[(size_t)b * 9 + d] = centers_rec[(size_t)(n - 1) * 3 + d];
    }
    for (int k = 0; k < nm; k++) {
      const bool in = k < n;
      n_planes[(size_t)b * nm + k] = in ? n_planes_rec[k] : 1;
      for (int d = 0; d < 3; d++) seeds_out[((size_t)b * nm + k) * 3 + d] = in ? seeds_rec[(size_t)k * 3 + d] : 0.0;
      for (int j = 0; j < p_max * 4; j++)
        planes[((size_t)b * nm + k) * p_max * 4 + j] = in ? planes_rec[(size_t)k * p_max * 4 + j] : 0.0;
    }
  }
  return direct_time_allocation(batch, nm, n_seg, start.data(), goal.data(), seeds_out, max_vel, max_acc, T0);
}

direct_status_t direct_traj_sample_batch(direct_ddp_handle_t h, const direct_sample_in_t* in, direct_sample_out_t* out) {
  if (!h || !in || !out) return fail(DIRECT_ERR_INVALID, "null argument");
  if (in->batch <= 0 || in->n_seg_max <= 0 || in->capacity <= 0 || in->derivs < 0 || in->derivs > 2)
    return fail(DIRECT_ERR_INVALID, "bad sizes");
  if (!(in->dt > 0.0)) return fail(DIRECT_ERR_INVALID, "dt must be positive");
  if (!in->n_seg || !in->bez || !in->T || !out->count || !out->pos) return fail(DIRECT_ERR_INVALID, "null array");
  if (out->cmax && (!in->planes || !in->n_planes || in->p_max <= 0))
    return fail(DIRECT_ERR_INVALID, "the containment audit (cmax) needs planes, n_planes and p_max");
  HIP_TRY(hipSetDevice(h->device));
  if (h->dtype == DIRECT_F64) return sample_t<double>(h, in, out);
  return sample_t<float>(h, in, out);
}

direct_status_t direct_traj_sample_last_ms(direct_ddp_handle_t h, float* ms) {
  if (!h || !ms) return fail(DIRECT_ERR_INVALID, "null argument");
  if (!h->sample_timed) return fail(DIRECT_ERR_INVALID, "no sampling launch to time");
  HIP_TRY(hipEventSynchronize(h->ev3));
  HIP_TRY(hipEventElapsedTime(ms, h->ev2, h->ev3));
  return DIRECT_OK;
}

}  // extern "C"

#if defined(DDP_SPLIT_PROBE)  // register-budget probe (tools only): the two sweeps as kernels of their own at four waves per SIMD
template <typename St, int RPL>
__global__ __launch_bounds__(64, 4) void k_probe_bwd(Batch<St> B) {
  __shared__ WaveLds<Cmp, St, RPL> lds;
  Wave<Cmp, St, RPL> W(B, lds, blockIdx.x);
  W.load_state();
  W.init_tables();
  W.bwd_sweep();
  W.store_state();
}
template <typename St, int RPL>
__global__ __launch_bounds__(64, 4) void k_probe_fwd(Batch<St> B) {
  __shared__ WaveLds<Cmp, St, RPL> lds;
  Wave<Cmp, St, RPL> W(B, lds, blockIdx.x);
  W.load_state();
  W.init_tables();
  W.fwd_pass();
  W.store_state();
}
template __global__ void k_probe_bwd<float, 2>(Batch<float>);
template __global__ void k_probe_fwd<float, 2>(Batch<float>);
#endif
