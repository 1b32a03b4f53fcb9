// Wave-level IPDDP: one 64-lane wavefront owns one trajectory.
//
// Restructured (not translated) from ntu-caokun/DIRECT global_planner/src/ddp_optimizer.cpp ("DDP"):
//   backward sweep   DDP:440-644   -> bwd_sweep()
//   forward pass     DDP:647-778   -> fwd_pass()
//   outer loop       DDP:295-412   -> iterate()
//   setup/first roll DDP:42-286    -> begin()
//   final conversion DDP:414-437   -> finish()
// The reference materialises cx (nc x 9) and cu (nc x 10) and multiplies them densely.  Here every
// constraint row is kept in its Kronecker form  [We[cr][:] (x) n | n . dval[cr]]  (SURVEY.md A.6), so
// cu' D cu etc. collapse to 3x3 accumulators per control point, rows are spread over the 64 lanes,
// and the 10x10 LLT + 10 right-hand sides run column-per-lane with v_readlane broadcasts.
//
// SIMT abstraction: code between LANES{...} is per-lane; everything outside is wave-uniform.
// On gfx950 LANES binds `lane = threadIdx.x`; under DIRECT_EMULATE (tests only, never in the
// product library) it is a 64-iteration loop so the very same source can be debugged on a CPU.
#pragma once
#include <stdint.h>

#include "ddp_tables.h"

#if defined(DIRECT_EMULATE)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
namespace direct {
using std::fabs; using std::fma; using std::fmax; using std::fmin; using std::log; using std::pow; using std::sqrt;
// The emulator also CHECKS what the device code merely assumes: a value passed through DDP_UNIFORM_* (a
// v_readfirstlane on the GPU) must be the same on every lane of the enclosing LANES block, otherwise the
// GPU silently uses lane 0's value for all of them.  emu_lane is the lane being emulated (64 outside blocks).
inline int& emu_lane() { static thread_local int l = 64; return l; }
template <typename T>
inline T* uniform_ptr(T* p) { return p; }
template <typename T>
inline T emu_uniform(T x, int line) {
  static thread_local unsigned char seen[4096][sizeof(double)];
  const int l = emu_lane();
  if (l >= 64) return x;
  unsigned char* ref = seen[line & 4095];
  if (l == 0) {
    std::memcpy(ref, &x, sizeof(T));
  } else if (std::memcmp(ref, &x, sizeof(T)) != 0 && !(x != x)) {  // NaNs of different payloads are not an error
    std::fprintf(stderr, "ddp_wave.h: wave-uniform site %d: the value differs between lanes 0 and %d\n", line, l);
    std::abort();
  }
  return x;
}
}
#define DDP_DEV inline
#define DDP_DEV_NOINLINE inline
#if defined(DDP_EMU_TRACE)  // tools/lds_trace: the tracer is told the lane by a call the optimiser cannot move
extern "C" int ddp_emu_set_lane(int);
#define LANES for (int lane = (direct::emu_lane() = 0, ddp_emu_set_lane(0)); lane < 64; direct::emu_lane() = ++lane, ddp_emu_set_lane(lane))
#else
#define LANES for (int lane = (direct::emu_lane() = 0); lane < 64; direct::emu_lane() = ++lane)
#endif
// LANES_AGAIN(l): a further per-lane block of a phase whose lane index l was declared with DDP_LANE_DECL
#define LANES_AGAIN(l) LANES
#define DDP_LANE_DECL(l) const int l = 0; (void)l
#define PLV(T, name) T name[64]
#define PLA(T, name, n) T name[64][n]
#define LV(name) name[lane]
#define WSYNC() ((void)0)
#define RDLANE(arr, idx, src) (arr[src][idx])
#define RDLANE_M(var, member, src) (var[src].member)
#define RDLANE_V(var, src) (var[src])
#define RDLANE_I(var, src) (var[src])
// value of per-lane array entry arr[idx] on lane `src` of the CALLER'S ROW of 16 lanes / acc -= that value * mul
#define ROW_BCAST(arr, idx, src) (arr[(lane & 48) + (src)][idx])
#define ROW_FNMA(acc, arr, idx, src, mul) ((acc) -= arr[(lane & 48) + (src)][idx] * (mul))
#define ROW_FMA(acc, arr, idx, src, mul) ((acc) += arr[(lane & 48) + (src)][idx] * (mul))
#define ROW_HAZARD(x) ((void)0)
// a predicate accumulated over several LANES blocks of which only lane 0's value is wanted
#define DDP_PRED_DECL(name) int name[64] = {0}
#define DDP_PRED_OR(name, cond) (name[lane] |= (cond) ? 1 : 0)
#define DDP_PRED_LANE0(name) (name[0])
#define DDP_UNIFORM_I(x) direct::emu_uniform((x), __COUNTER__)
#define DDP_UNIFORM_R(x) direct::emu_uniform((x), __COUNTER__)
#define DDP_UNIFORM_PW(x) direct::emu_uniform((x), __COUNTER__)
#define ROW_FMA_V(acc, var, src, mul) ((acc) += var[(lane & 48) + (src)] * (mul))
#define DDP_LAUNDER_S(x) ((void)0)
#define DDP_LOADS_ISSUED() ((void)0)
#define DDP_PIN(x) ((void)0)
#define DDP_OPAQUE_S(x) ((void)0)
#define DDP_UMUL24(a, b) ((a) * (b))
#define DDP_GLOBAL
#if defined(DDP_EMU_TRACE)  // tools/lds_trace: LDS bank-conflict attribution on the emulator (phase marks, 16-byte loads)
extern "C" void ddp_emu_mark(const char*);
extern "C" void ddp_emu_ld2(const void*, int);
#define DDP_MARK(name) ddp_emu_mark(name)
#else
#define DDP_MARK(name)
#endif
#else
#include <hip/hip_runtime.h>
#define DDP_DEV __device__ __forceinline__
#ifdef DDP_NOINLINE_SWEEPS
#define DDP_DEV_NOINLINE __device__ __attribute__((noinline))
#else
#define DDP_DEV_NOINLINE __device__ __forceinline__
#endif
// `lane` is laundered through an empty asm so that LICM cannot hoist the dozens of lane-derived
// indices and loop-invariant LDS table reads of every phase out of the knot loop (that costs >128
// VGPRs and with them the occupancy); re-deriving them per phase is a handful of integer ops.
#define LANES for (int lane = direct::opaque_lane(), lanes_once_ = 1; lanes_once_; lanes_once_ = 0)
// Several per-lane blocks of ONE phase that share one laundered lane index (each LANES block launders its own: a
// v_mov and the re-derivation of everything computed from it)
#define LANES_AGAIN(l) for (int lane = l, lanes_once_ = 1; lanes_once_; lanes_once_ = 0)
#define DDP_LANE_DECL(l) const int l = direct::opaque_lane()
#define PLV(T, name) T name
#define PLA(T, name, n) T name[n]
#define LV(name) name
// Wave-local phase boundary.  One workgroup is one wavefront and the LDS unit executes a wave's DS
// instructions in order, so a ds_write is visible to every later ds_read of the same wave without any
// wait; all that is needed is that the COMPILER does not reorder LDS accesses across the boundary.
// (__syncthreads() would also emit s_waitcnt vmcnt(0) and stall every phase on the outstanding HBM
// prefetch loads and gain stores.)  Global memory is only ever re-read by the lane that wrote it.
#define WSYNC()                                  \
  do {                                           \
    asm volatile("" ::: "memory");               \
    __builtin_amdgcn_wave_barrier();             \
  } while (0)
#define RDLANE(arr, idx, src) direct::readlane_real(arr[idx], src)
#define RDLANE_M(var, member, src) direct::readlane_real(var.member, src)
#define RDLANE_V(var, src) direct::readlane_real(var, src)
#define RDLANE_I(var, src) __builtin_amdgcn_readlane(var, src)
// Row broadcasts on the DP-ALU's DPP path (gfx90a+: 64-bit DPP supports row_newbcast only): lane `src` (a compile-time
// constant, 0..15) of every row of 16 lanes is the operand of all 16 lanes of that row.  ROW_FNMA is ONE instruction,
// v_fmac_f64_dpp: acc -= bcast(arr[idx]) * mul - against two v_readlane plus the FMA of the SGPR-broadcast form.
// The hazard recognizer does not look into inline asm: a VGPR written by the VALU needs two wait states before a DPP
// read (ROW_HAZARD pins the producer before the s_nop; the ROW_BCAST form carries its own).
#define ROW_BCAST(arr, idx, src) direct::row_bcast<src>(arr[idx])
#define ROW_FNMA(acc, arr, idx, src, mul) direct::row_fnma<src>(acc, arr[idx], mul)
#define ROW_FMA(acc, arr, idx, src, mul) direct::row_fma<src>(acc, arr[idx], mul)
#define ROW_HAZARD(x) asm volatile("s_nop 1" : "+v"(x))
// a predicate accumulated over several LANES blocks of which only lane 0's value is wanted: the compare's own lane
// mask, OR-ed on the scalar unit (a per-lane flag costs a v_cndmask per update, or is merged into an fmin / fmax chain)
#define DDP_PRED_DECL(name) unsigned long long name = 0ull
#define DDP_PRED_OR(name, cond) (name |= __ballot(cond))
#define DDP_PRED_LANE0(name) ((int)(name & 1ull))
#define DDP_UNIFORM_I(x) __builtin_amdgcn_readfirstlane(x)
// a wave-uniform real computed by the VALU (or read from LDS) moved to SGPRs: frees its VGPRs
#define DDP_UNIFORM_R(x) direct::uniform_real(x)
// The powers of T: wave-uniform values that every lane computes for itself and keeps in VGPRs.  Moving a double to
// SGPRs costs two v_readfirstlane on top of the multiply that produced it (there is no scalar f64 multiply): three
// VALU instructions per power instead of one.
#define DDP_UNIFORM_PW(x) (x)
#define ROW_FMA_V(acc, var, src, mul) direct::row_fma<src>(acc, var, mul)
// re-materialise a wave-uniform value: stops LICM from hoisting everything derived from it (slab
// pointers, strides) out of the outer iteration loop, where it would stay live across both sweeps
#define DDP_LAUNDER_S(x) asm volatile("" : "+s"(x))
// Compiler-only memory barrier placed after a block of LDS operand loads: the scheduler otherwise
// sinks each load next to its use (3 loads, wait, 2 FMAs, 3 loads, wait ...) and every wait exposes
// a full LDS round trip; with the barrier all loads of the block are in flight before the first wait.
#define DDP_LOADS_ISSUED() asm volatile("" ::: "memory")
// Pins a value's computation at this point of the program.  Without it the compiler sinks a whole
// unrolled recurrence below the operand loads of ALL its steps (into the block that finally stores
// the result), and the operands of every step are live at once.
#define DDP_PIN(x) asm volatile("" : "+v"(x))
// A wave-uniform value the compiler must treat as an opaque SGPR operand.  Without it a lane-dependent
// select between two kernel-argument fields is turned into ONE lane-indexed vector load from the kernarg
// segment, i.e. a full memory round trip (and a vmcnt(0) that also drains the prefetch) in the row loop.
#define DDP_OPAQUE_S(x) asm("" : "+s"(x))
#define DDP_UMUL24(a, b) __umul24(a, b)  // full-rate 24-bit multiply (v_mul_lo_u32 is quarter rate)
// A pointer that went through an empty asm is a generic pointer to the compiler (FLAT loads, which also
// count on lgkmcnt); the sweep's array bases are therefore typed as global-memory pointers.
#define DDP_GLOBAL __attribute__((address_space(1)))

#if defined(DDP_MARKS)  // phase markers in the .s for static instruction accounting (tools/phase_count.py)
#define DDP_MARK(name) asm volatile("; DDP_MARK " name ::: "memory")
#elif defined(DDP_TIMING)  // per-phase cycle accounting with s_memtime (tools/phase_timing.py); debug builds only
#define DDP_MARK(name) direct::phase_tick(name, DDP_TICK_OBJ.tick)
#define DDP_TICK_OBJ (*this)
#else
#define DDP_MARK(name)
#endif
#endif

namespace direct {

// compile-time loop: f(IC<B>{}), f(IC<B+1>{}), ... - the DPP lane selectors must be constant expressions
template <int I>
struct IC {
  static constexpr int v = I;
  constexpr operator int() const { return I; }
};
template <int B, int E, typename F>
DDP_DEV void static_for(F&& f) {
  if constexpr (B < E) {
    f(IC<B>{});
    static_for<B + 1, E>(f);
  }
}
template <int B, int E, typename F>
DDP_DEV void static_for_down(F&& f) {  // B, B-1, ..., E
  if constexpr (B >= E) {
    f(IC<B>{});
    static_for_down<B - 1, E>(f);
  }
}

constexpr int kPLim = 128;  // DIRECT_P_LIMIT: what polyhedronGenerator can emit (128 planes); fourteen row slots per lane hold (64 * 14 - 55) / 6 = 140

#if !defined(DIRECT_EMULATE)
#if defined(DDP_TIMING)
// Per-phase cycle accounting of workgroup 0 (debug builds; tools/phase_timing.py).  Cheap enough not to distort what it
// measures: s_memtime, one scalar subtraction and two fire-and-forget LDS atomics per mark; the running state (last
// time stamp, phase that is open) lives in two wave-uniform members of the Wave object, the totals in LDS until the
// "Z_E" mark at the end of the kernel adds them to g_phase_cycles.  No vmcnt wait: the HBM prefetch stays in flight.
__device__ unsigned long long g_phase_cycles[64];
__device__ __forceinline__ constexpr int phase_id(const char* n) {
  // B_L 0 B_T1 1 B_T2 2 B_R1 3 B_S 4 B_S2 5 B_H 6 B_C 7 B_G 8 B_R2 9 B_END 10 F_L 11 F_D 12 F_T 13 F_R 14 F_END 15
  if (n[0] == 'Z') return 19;
  if (n[0] == 'X') return n[2] == 'T' ? 16 : n[2] == 'G' ? 17 : 18;  // X_T ticket wait, X_G state load + sweep prologue, X_A after the forward pass
  return n[0] == 'B' ? (n[2] == 'L' ? 0 : n[2] == 'T' ? (n[3] == '1' ? 1 : 2) : n[2] == 'R' ? (n[3] == '1' ? 3 : 9)
                        : n[2] == 'S' ? (n[3] == '2' ? 5 : 4) : n[2] == 'H' ? 6 : n[2] == 'C' ? 7 : n[2] == 'G' ? 8 : 10)
                     : (n[2] == 'L' ? 11 : n[2] == 'D' ? 12 : n[2] == 'T' ? 13 : n[2] == 'R' ? 14 : 15);
}
struct TickState {
  unsigned long long tlast;
  int open;  // id of the phase that is running
};
// time since the previous mark is charged to the phase that ENDS here (= the previous mark's phase)
__device__ __forceinline__ void phase_tick(const char* name, TickState& ts) {
  if (blockIdx.x == 0) {
    __shared__ unsigned long long s_acc[32];
    __shared__ unsigned int s_cnt[32];
    const unsigned long long t = __builtin_readcyclecounter();
    if (name[0] == 'Z' && name[2] == '0') {  // first mark of the kernel
      if (threadIdx.x < 32) { s_acc[threadIdx.x] = 0; s_cnt[threadIdx.x] = 0; }
    } else {
      if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&s_acc[ts.open], t - ts.tlast, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&s_cnt[ts.open], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (name[0] == 'Z' && name[2] == 'E') {  // last mark: totals to global memory
        __syncthreads();
        if (threadIdx.x < 32) {
          atomicAdd(&g_phase_cycles[threadIdx.x], s_acc[threadIdx.x]);
          atomicAdd(&g_phase_cycles[32 + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
        }
      }
    }
    ts.tlast = t;
    ts.open = phase_id(name);
  }
}
#endif
__device__ __forceinline__ int opaque_lane() {
  int l = (int)threadIdx.x;
  asm volatile("" : "+v"(l));
  __builtin_assume(l >= 0 && l < 64);  // range for the index arithmetic (24-bit multiplies, unsigned division)
  return l;
}
__device__ __forceinline__ float readlane_real(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ double readlane_real(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
template <int SRC>
__device__ __forceinline__ double row_bcast(double v) {
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(SRC));
  return r;
}
template <int SRC>
__device__ __forceinline__ void row_fma(double& acc, double a, double b) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(SRC));
}
template <int SRC>
__device__ __forceinline__ void row_fnma(double& acc, double a, double b) {
  asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(SRC));
}
__device__ __forceinline__ float uniform_real(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
  const unsigned long long a = (unsigned long long)p;
  return (T*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)a));
}
__device__ __forceinline__ double uniform_real(double v) {
  int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
// Wave reductions on the VALU alone (DPP): __shfl_xor compiles to ds_bpermute, i.e. onto the LDS pipe,
// which is the busiest unit of the sweeps.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double v) {  // lanes without a source read 0
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {
  v += dpp_d<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_d<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_d<0x141, 0xf>(v);  // row_half_mirror
  v += dpp_d<0x140, 0xf>(v);  // row_mirror: every lane holds the sum of its row of 16
  v += dpp_d<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3
  v += dpp_d<0x143, 0xc>(v);  // row_bcast31 into rows 2 and 3: lanes 48..63 hold the total
  return readlane_real(v, 63);
}
__device__ __forceinline__ double wave_max_d(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ int wave_any(int v) { return __any(v) ? 1 : 0; }
// number of lanes below this one with a non-zero flag (exclusive prefix count) and the wave total
__device__ __forceinline__ int wave_prefix_count(int flag, int& total) {
  const unsigned long long m = __ballot(flag);
  total = __popcll(m);
  return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
#define WAVE_SUM_D(name) direct::wave_sum_d((double)(name))
#define WAVE_MAX_D(name) direct::wave_max_d((double)(name))
#define WAVE_SUM_I(name) direct::wave_sum_i(name)
#define WAVE_ANY(name) direct::wave_any(name)
#define WAVE_PREFIX_COUNT(flag, pos, total) pos = direct::wave_prefix_count(flag, total)
#else
template <typename T>
inline double emu_sum_d(const T* v) {
  double vals[64];
  for (int i = 0; i < 64; i++) vals[i] = (double)v[i];
  for (int o = 32; o > 0; o >>= 1)  // same butterfly order as the device
    for (int i = 0; i < o; i++) vals[i] += vals[i + o];
  return vals[0];
}
template <typename T>
inline double emu_max_d(const T* v) {
  double m = (double)v[0];
  for (int i = 1; i < 64; i++) m = std::fmax(m, (double)v[i]);
  return m;
}
inline int emu_sum_i(const int* v) {
  int s = 0;
  for (int i = 0; i < 64; i++) s += v[i];
  return s;
}
inline int emu_any(const int* v) {
  for (int i = 0; i < 64; i++)
    if (v[i]) return 1;
  return 0;
}
#define WAVE_SUM_D(name) direct::emu_sum_d(name)
#define WAVE_MAX_D(name) direct::emu_max_d(name)
#define WAVE_SUM_I(name) direct::emu_sum_i(name)
#define WAVE_ANY(name) direct::emu_any(name)
#define WAVE_PREFIX_COUNT(flag, pos, total)                 \
  do {                                                      \
    total = 0;                                              \
    for (int l_ = 0; l_ < 64; l_++) {                       \
      pos[l_] = total;                                      \
      total += flag[l_] ? 1 : 0;                            \
    }                                                       \
  } while (0)
#endif

// ---- per-solve constants (by-value arguments of polyCurveGeneration, DDPH:275-289) ------------
struct SolveConst {
  double max_vel, max_acc, w_snap, w_term, w_time, reg_base, shift, tol;
  int iter_max, time_power, zero_init, line_init, minvo, fixed_iters, exact_dt;
  int pair_trials;  // scheduling only: evaluate line-search steps 1..10 two per sweep (results are unchanged)
};

// ---- per-trajectory solver state (algParam + the scalar members of fwdPass / bwdPass) ---------
struct TrajState {
  double cost, costq, logcost, err, mu, opterr, stepsize, prev_cost;
  double sumlog, errsum;  // sum log(-c) or sum log(y); sum |c+y| of the CURRENT iterate
  int reg, step, fp_failed, bp_failed;
  int rtn, iter, done, nfilter;
  int infeas, infeas_ref, line_failed, bp_no_upd;
  int no_upd, fwd_passes, cur, viol;
  int neg_time, nseg, nc0, npos;  // npos: rows with c > 0 in the last evaluation sweep (DDP:255-269)
};
// What the gains of a trajectory were formed FROM (one per trajectory, global memory; outside TrajState because the hot
// kernel's LDS has no 24 bytes to spare at twelve waves per CU).  The reference's forward pass steps slacks and duals with
// gains stored by the backward pass (DDP:568-572, 611-614 -> 680-703); the kernels regenerate them from the iterate the
// sweep read (see run_round, phase R) - which is only the CURRENT iterate as long as the sweep completed.  When the retry
// sequence of DDP:297-310 gives up, knots [0, kreach) still hold the gains of the last sweep that reached them: formed from
// buffer `buf` with barrier parameter `mu` (Wave::stale_fwd_pass).
struct GainBase {
  double mu;   // alg.mu of the last COMPLETED backward sweep
  int buf;     // the iterate buffer it read; -1: no sweep has completed yet (the reference's gains are still zero, DDP:154-159)
  int kreach;  // knots [0, kreach) have not been reached by any sweep since then (N right after a completed sweep)
};
constexpr int kRtnStuckPending = -40;  // TrajState::rtn between the hot kernel (backward pass stuck, DDP:297-310) and k_stuck

// ---- shared line search (scheduling only, see Wave::fwd_pass) -----------------------------------
// What one completed line-search trial reports: enough for the filter test and for the state an accepted trial
// leaves behind.  Wave-uniform; a helper wave hands it to the trajectory's owner through HelpSlot::res.
struct TrialRes {
  int alive, step, neg, viol;
  double stepsize, qsum, cost, sumlog, errsum;
  double pad;
};
// One per trajectory.  The owner of a trajectory opens its line search to other waves after step 0 failed; waves
// that are waiting for this very trajectory (k_iterate_dyn) claim rounds of it instead of sleeping.
struct HelpSlot {
  int gen;         // 0: closed; else the tag of the open line search (release-published by the owner)
  int next_round;  // next unclaimed round, 1..last_round; owner and helpers draw from it
  int active;      // helpers inside the protocol; the owner closes with gen = 0 and waits for 0
  int cancel;      // an earlier step was accepted: running rounds stop at the next knot
  int cur;         // the nominal iterate's buffer
  int last_round;  // 5: rounds are the step pairs (1,2) .. (9,10); 10: rounds are the single steps 1 .. 10
  double mu;
  int done[12];    // done[r] == gen: round r's results are in res[], one record per step (release-published by its runner)
  int pad[4];
  TrialRes res[12];
};
// Polls after which a wait inside a launch (for a helper's record, a helper to leave, a trajectory's previous chunk:
// direct_ddp.hip, next_work) counts as a scheduling error instead of hanging.  Round 6: 2^27 (two minutes or more) instead of
// 2^22 (1 - 5 s) - with two handles' kernels running next to each other, one launch in ~5000 saw a seventeenth of its waves
// stand still for 10 - 64 s and then go on (not a lost update, not a deadlock: with longer limits every one of 64 000 stress
// launches completed bit-identically, nine groups of four in 10 .. 64 s instead of 0.3).
#ifndef DDP_SPIN_LIMIT_LOG2
#define DDP_SPIN_LIMIT_LOG2 27
#endif
constexpr int kSpinLimit = 1 << DDP_SPIN_LIMIT_LOG2;
constexpr int kMaxBuf = 12;  // iterate buffers: `cur` + one per concurrently evaluated step, 0 .. 10 (3 without helpers)

// ---- helper-assisted backward sweep (scheduling only, see Wave::bwd_sweep_t) -----------------------
// One per trajectory.  A third of a backward knot does not depend on the value function (control values, constraint rows,
// the constraint part of the condensed system: phases T2, R1 rows, S, S2 and the front half of H): the owner of a sweep
// publishes it, and waves that wait for this trajectory's next ticket compute that part ("front") for knots of their
// own, from knot 0 upwards, into hand-over records in HBM, while the owner sweeps down from knot N - 1; where the two
// meet the owner goes on with the value-dependent rest ("back") alone.
struct BwdShare {
  unsigned word;             // (tag << kBsCountBits) | helpers inside the protocol; tag 0: no sweep open.  Only ever changed by atomic RMWs
  int cur, infeas, seq;      // the sweep's iterate buffer and mode; seq: sweeps opened so far (owner only: the tag source)
  double mu;
  unsigned long long claim;  // free knots [low, high): low in bits 0..31 (helpers take from below), high + 2^30 in bits 32..63 (the owner from above)
};
// The count field holds every wave that can be resident at once (13 bits: 8191; a device holds 4096 waves of 64 lanes at
// most), so entrants that are about to back out can never carry into the tag - and only kBsMaxHelpers are let in at all: a
// sweep of N knots has N / kHelpChunk claims to hand out, while with few trajectories on many waves EVERY waiting wave polls
// the one open sweep (iter_max of a few hundred at B = 1 used to overflow an 8-bit count: ADVICE r05).
constexpr int kBsCountBits = 13;
constexpr unsigned kBsCountMask = (1u << kBsCountBits) - 1u, kBsTagMask = (1u << (32 - kBsCountBits)) - 1u;
constexpr int kBsMaxHelpers = 64;
constexpr int kRecDoubles = 384;  // one knot's hand-over record: six doubles per lane, stored as three 16-byte halves per lane
#ifndef DDP_OWN_CHUNK
#define DDP_OWN_CHUNK 2
#endif
#ifndef DDP_HELP_CHUNK
#define DDP_HELP_CHUNK 4
#endif
constexpr int kOwnChunk = DDP_OWN_CHUNK, kHelpChunk = DDP_HELP_CHUNK;  // knots per claim (the owner takes little: a knot it runs fused costs it twice a knot from a record)
constexpr unsigned long long kClaimBias = 1ull << 30;

// ---- device-resident batch (all pointers are device memory) -----------------------------------
template <typename Real>
struct Batch {
  int B, nmax, pmax, ncs;  // B: trajectories of THIS launch; ncs = row stride of S/Y/KS/KY (>= 6*pmax+55)
  // Row-slot classes (direct_ddp.hip): a launch covers the trajectories idx[0 .. B) of the handle's batch, the ones whose
  // widest polytope fits this instantiation's row slots; null: trajectories 0 .. B - 1 themselves.  Every per-trajectory
  // array stays indexed by the trajectory's own number.
  const int32_t* idx;
  int fcap, nbuf, help_early, tail_thresh;  // nbuf: iterate buffers in use (3, or kMaxBuf with the shared line search);
                                     // help_early: single-step searches are open to helpers from step 0 on
  const int32_t* n_seg;
  const Real* x0;
  const Real* xd;
  const Real* T0;
  const int32_t* n_planes;
  const Real* planes;
  const Real* init_bez;
  const Real* init_poly;
  const Real* seeds;     // [B][nmax][3] Polytope.seed_coord (line-init only)
  const uint8_t* infeas_in;
  // Iterate buffers: `cur` and the trial buffers (step t of a line search writes buffer trial_buf(cur, t))
  Real* X[kMaxBuf];   // [B][nmax+1][x_stride<Real>()]: x_k (9), u_k (10) (+ with float storage their 19 low words), pad
  Real* S[kMaxBuf];   // [B][nmax][ncs]
  Real* Y[kMaxBuf];   // [B][nmax][ncs]
  Real* KU;     // [B][nmax][100]: ku (10), Ku (10x9 row-major)
  Real* KS;     // [B][nmax][ncs]
  Real* KY;     // [B][nmax][ncs]
  double* filt; // [B][fcap][2]
  TrajState* st;
  GainBase* gbase;  // [B]
  HelpSlot* help;  // [B], or null: every line search stays with its owner
  int* sched_err;  // the launch's sticky error flag (spin limits of the shared line search)
  int* live;  // trajectories of the launch still in their outer loop (ticket scheduler; null otherwise): when no more
              // than tail_thresh are left the line searches switch to single steps open from step 0 - few trajectories on
              // many waves, the tail of a natural-exit launch (scheduling only)
  unsigned long long* visits;  // [4] knots executed by backward sweeps / by forward trials, -, accepted line searches (observability; may be null)
  // helper-assisted backward sweep: null / 0 = every sweep stays with its owner
  BwdShare* bshare;  // [B]
  int* bflag;        // [B][nmax]: tag of the sweep whose record of this knot is complete
  double* brec;      // [B][nmax][kRecDoubles]
  int bforce;        // tests: the owner itself runs the helpers' half first (every knot but its first claim goes through a record)
  const void* self;  // this very struct in device memory: what the out-of-line halves of a shared sweep (front_cold, back_cold) are handed
  unsigned long long* bvisits;  // [1] knots whose front half a helper (or the forced split) computed (observability; may be null)
#if defined(DDP_SCHED_DEBUG)   // development builds: where every persistent wave of the launch currently is (direct_ddp.hip, next_work's time-out record)
  int* mark;                   // [grid]
#define DDP_DBG_MARK(Bq, v) do { if ((Bq).mark != nullptr && threadIdx.x == 0) (Bq).mark[blockIdx.x] = (v); } while (0)
#else
#define DDP_DBG_MARK(Bq, v) ((void)0)
#endif
#if defined(DDP_TIMELINE)  // debug builds (tools/timeline.py): wall-clock stamps per trajectory and outer iteration
  unsigned long long* tl;  // [B][kTimelineDepth][4]: start, end of the backward sweeps, end (100 MHz), knots that came through records
#endif
  SolveConst k;
};

#if defined(DDP_TIMELINE)
#ifndef DDP_TIMELINE_DEPTH
#define DDP_TIMELINE_DEPTH 32
#endif
constexpr int kTimelineDepth = DDP_TIMELINE_DEPTH;  // outer iterations stamped per trajectory
#endif
constexpr int kXS = 24;  // LDS knot record: x (9), u (10), pad
// Knot record stride of X in HBM.  With float storage EVERY entry of the iterate (x, u) is an unevaluated hi + lo float
// pair (words a and 19 + a): the filter line search demands a STRICT decrease of a log-cost of ~2e6 that moves by 1e-10
// per iteration near a stall, and on a single-float lattice (6e-8 relative) small steps are rounded away - float
// storage then left the fp64 iterates at the stagnation exits of 11 % of BASELINE config 2's problems
// (profiles/r03_n100_parity.json).  Gains, slacks and duals stay single floats: they shape the step, not the iterate.
template <typename St>
constexpr int x_stride() { return sizeof(St) < sizeof(double) ? 40 : 24; }
typedef double Acc;      // accumulator type of the condensed system (see WaveLds)

// state of one line-search trial of the forward pass (LDS)
template <typename Real>
struct FwdTrial {
  Real tpn[8], zn[kXS], dz[kXS], xn[12], xnx[12], valn[48], qp[12], G[48];
};

// ---- LDS (one per wave) ------------------------------------------------------------------------
// RowT (the storage type) is kept for the template's signature; the per-row D / g staging arrays are double for both
// storage types (float staging cost a conversion per row and plane in phase S and two per row in phase R1).
template <typename Real, typename RowT, int RPL>
struct WaveLds {
  typedef FwdTrial<Real> FwdT;
  static constexpr int kPMax = (64 * RPL - 55) / 6 > kPLim ? kPLim : (64 * RPL - 55) / 6;  // planes per knot
  TrajState st;
  // value / d-dT base tables with Ek_inv folded in (fixed for the launch).  Rows 15..17: [F|G] and its
  // T-derivative, rows 18..20: the jerk Gram matrix R acting on u and its T-derivative (see ctrl_off()).
  Real WbE[126], WdE[126];
  Real Rc[10];            // jerk Gram coefficients: R[a][a'] = Rc * T^(a+a'+1); Rc[9] = 0
  // Per-lane role descriptors of the backward sweep's assembly phases, packed (init_tables; fixed for the
  // launch).  They depend on the lane alone, but `lane` is laundered per phase (see LANES) so that the indices are
  // not hoisted into a hundred live registers - which made every knot re-derive them with ~40 integer
  // instructions per phase.  One ds_read_b32 and a few bit-field extracts replace that.
  int lt[8][64];                  // rows 0..3: phase H; 4..6: the three VZ passes of phase R1; 7: phase T2
  unsigned short lt16[5][64];     // 0: value recursion of phase R2; 1: phase S; 2..4: phase H (operand offsets)
  Real ones[5];           // 1.0: the neutral second factor of phase S's g rows (addressed like a plane component)
  // per knot, both sweeps
  Real tp[8];             // powers of T
  Real z[kXS];
  alignas(16) Real pl[4 * kPMax + 20];  // the knot's planes, then five pseudo-planes for the velocity / acceleration / T rows
  Real val[48], G[48];
  union {
    struct {  // ---- backward sweep only
      // We = WbE o T-powers for rows 0..17; rows 15..17 are Z = [F|G] itself
      alignas(16) Real We[112];  // layout we_idx(): [i][row]; [108..111]: dump slot of phase T2's masked stores
      Real dval[48];
      // The condensed 19x19 system, its Cholesky and the value-function recursion are kept in double
      // (Acc) whatever Real is: cu'Dcu with D = s/c and the Vxx update cancel catastrophically in
      // fp32 over ~100 knots (DESIGN.md "Precision").
      Acc fT[12], Ru[9], Rpu[9], Rppu[9];
      // V is dead between phase R1 (VZ = V Z) and phase R2 (which rewrites it): two assembly operands of phases S .. H
      // live in its storage meanwhile
      union {
        alignas(16) Acc V[81];
        struct {
          alignas(16) Acc Sd[54];  // layout ch_idx(): [axis][control row]
          alignas(16) Acc dl[30];  // layout dl_idx(): [axis][velocity / acceleration control point]
        };
      };
      Acc Vx[12];
      // The condensed system.  Hxx (9x9): the upper triangle is what the value recursion reads.  HR: rows 9..18 of the
      // 19x19 matrix WITH the right-hand side as a twentieth column - HR[a] = [Hux[a][0..8] | Huu[a][0..9] | Hu[a]], row
      // stride kHRS = 21: phase C's read `a` of all twenty column lanes is then one run of twenty consecutive doubles
      // (conflict-free; the 9x10 + 10x10 + separate Hz layout of r04 put three address families on the same banks), and
      // phase H's scattered stores see two-way bank conflicts at worst instead of three-way (tools/lds_trace; the
      // offset of HR behind Hxx is part of that tuning).  Hzx = Hx; hdump: the slot of phase H's masked stores.
      // Everything is addressed from Hxx[0] (phase H stores through one base pointer).  Phase C reads HR into registers
      // and never again, so the gains it produces share that storage.
      // The per-row weights D = s/c and g of phase R1 are consumed by phase S, before phase H writes the
      // condensed system, and the system (and the gains) of a knot are dead when the next knot's R1 runs:
      // the two share storage.
      union {
        struct {
          Acc Hxx[81], hpad[7];
          union {
            Acc HR[10 * 21];
            Acc KU[100];
          };
          Acc Hzx[9], hdump[1];
        };
        struct {
          Acc drow[64 * RPL], grow[64 * RPL];
        };
      };
      union {
        struct {  // operands of the assembly: dead once phase H is done
          alignas(16) Acc Sp[36];  // layout sp_idx(): [entry of the symmetric 3x3][control point]
          alignas(16) Acc hh[54];  // layout ch_idx()
          Acc last[4];
          Acc VZ[176];  // Vxx * Z
        };
        struct {
          // rows of [L^T | y | Y]: UY[k][j] = L[j][k] (j < 10), UY[k][10 + c] = (L^-1 [Hu | Hux])[k][c];
          // written one row per elimination step by all column lanes at once (phases C, R2).  Stride 20,
          // 64 slots per row so that the idle lanes' stores need no masking.
          Acc UY[10 * 20 + 44];
        };
      };
    };
    struct {  // ---- forward pass / evaluation sweep only
      Real KUr[100];  // gains of the knot as stored in HBM
      FwdT ft[2];     // two line-search trials (step sizes alpha and alpha / 2) share a sweep
    };
  };
};

// Reciprocal, reciprocal square root and mantissa/exponent split.  On gfx950: the hardware seed
// (v_rcp / v_rsq, ~24 good bits in f64) plus two Newton steps -- 5 instructions instead of the ~12 of
// the IEEE division expansion; results are within 1-2 ulp, far inside every stated tolerance.
#if defined(DIRECT_EMULATE)
DDP_DEV double frcp(double x) { return 1.0 / x; }
DDP_DEV float frcp(float x) { return 1.0f / x; }
DDP_DEV double frsq(double x) { return 1.0 / std::sqrt(x); }
DDP_DEV float frsq(float x) { return 1.0f / std::sqrt(x); }
DDP_DEV double split_mant(double x, int& e) { return std::frexp(x, &e); }
DDP_DEV float split_mant(float x, int& e) { return std::frexp(x, &e); }
#else
DDP_DEV double frcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
DDP_DEV float frcp(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  return fmaf(fmaf(-x, r, 1.0f), r, r);
}
DDP_DEV double frsq(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}
DDP_DEV float frsq(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  return y * fmaf(-0.5f * x * y, y, 1.5f);
}
DDP_DEV double split_mant(double x, int& e) {
  e = __builtin_amdgcn_frexp_exp(x);
  return __builtin_amdgcn_frexp_mant(x);
}
DDP_DEV float split_mant(float x, int& e) {
  e = __builtin_amdgcn_frexp_expf(x);
  return __builtin_amdgcn_frexp_mantf(x);
}
#endif

// sum of logarithms as the logarithm of a running product: mantissa in [0.5,1) and an integer
// exponent per lane, one log() per lane at the end instead of one per constraint row.
template <typename Real>
struct LogProd {
  Real m;
  int e;
  DDP_DEV void init() { m = (Real)1; e = 0; }
  // x > 0 (a negative or NaN factor poisons m, a zero one makes it 0, exactly like log(x) would).  The
  // product is only renormalised by norm(), which the sweeps call once per knot: at most RPL factors,
  // each within a few hundred binades of 1, accumulate in between.
  DDP_DEV void mul(Real x) {
    if (x < (Real)0) x = (Real)NAN;  // log of a negative number is NaN (ddp_optimizer.cpp:731, quirk Q7)
    m *= x;
  }
  DDP_DEV void norm() {
    int e2;
    m = split_mant(m, e2);
    e += e2;
  }
  DDP_DEV Real value() const { return log(m) + (Real)e * (Real)0.6931471805599453094; }
};

template <typename Real>
DDP_DEV Real powi(Real T, int e) {  // T^e for 0 <= e <= 7 without a register-array index
  Real r = (Real)1;
#pragma unroll
  for (int q = 0; q < 7; q++) r = (q < e) ? r * T : r;
  return r;
}

// Two neighbouring elements with ONE load: 16 bytes for doubles, i.e. ds_read_b128, which reaches its rate from one
// wave per SIMD where 8-byte reads need four (MI355X_MICROARCH.md, LDS) - the sweeps run at three.  p must be aligned
// to two elements; the LDS layouts below (Wave::we_idx ...) are chosen so that it is.
#if defined(DIRECT_EMULATE)
template <typename T>
DDP_DEV void ld2(const T* p, T& a, T& b) {
#if defined(DDP_EMU_TRACE)
  ddp_emu_ld2(p, (int)(2 * sizeof(T)));
#endif
  a = p[0];
  b = p[1];
}
#else
template <typename T>
DDP_DEV void ld2(const T* p, T& a, T& b) {
  typedef T __attribute__((ext_vector_type(2))) V2;  // a struct of two would be split into two scalar loads again
  const V2 v = *reinterpret_cast<const V2*>(p);
  a = v.x;
  b = v.y;
}
#endif
// element of an array at a BYTE offset (the per-lane role tables hold ready-made byte offsets)
template <typename T>
DDP_DEV T* byte_at(T* base, int byte_off) { return (T*)((char*)base + byte_off); }
template <typename T>
DDP_DEV const T* byte_at(const T* base, int byte_off) { return (const T*)((const char*)base + byte_off); }

// T^e for 0 <= e <= 7 by binary powering from T, T^2, T^4 (3 selects + 2 multiplies)
template <typename Real>
DDP_DEV Real pow3(Real T, Real T2, Real T4, int e) {
  Real r = (e & 1) ? T : (Real)1;
  r *= (e & 2) ? T2 : (Real)1;
  r *= (e & 4) ? T4 : (Real)1;
  return r;
}

// Exponent offset of a row of the value table: entry i carries T^(i - o).  Rows 0..14 are the position /
// velocity / acceleration control points; rows 15..17 are the rows of [F | G] (o = row) and rows 18..20
// the rows of the jerk Gram matrix R acting on u (o = 2 - row): the dynamics and the running cost have
// the same "weight * T^(i-o) * coefficient" structure and run through the same code (forward pass, phase T).
DDP_DEV int ctrl_off(int cr) { return cr < 6 ? 0 : (cr < 11 ? 1 : (cr < 15 ? 2 : (cr < 18 ? cr - 15 : 20 - cr))); }

// One constraint row as seen by (slot, lane).  Rows are dealt to lanes BY KIND so that a slot runs one
// kind of code: slots 0..RPL-2 hold position rows r = lane + 64*slot (r < 6P); the last slot holds the
// 55 velocity / acceleration / T_min rows in lanes 0..54 (r = 6P + lane) and, in lanes 55..63, the few
// position rows beyond 64*(RPL-1).  HBM arrays stay indexed by r (DDP:1181-1188, 1236-1238, 1274-1279).
// Every row is evaluated by ONE branch-free formula,  c = n . A[a0..a0+2] + o - shift :
//   position row:  n = plane normal, o = plane offset, a0 = 3 * (control point)
//   other rows:    n = (0, 0, +/-1), a0 = (index of the bounded value) - 2, o = -max_vel | -max_acc | 0.3
//                  (A[45] carries T for the T_min row; the two zero products are exact)
// Rows that do not exist (r < 0) alias position row 0: their arithmetic is harmless and callers mask the
// stores and the reductions.
template <typename Real>
struct RowK {
  int r, a0;
  Real n0, n1, n2, o;
};

// The two halves of a SHARED backward sweep that are not the fused knot (Wave::bwd_front_run, Wave::bwd_back_run) are
// real function calls, compiled on their own: inlined next to the fused sweep and the forward rounds - one function
// of 14 k instructions at the register limit of three waves per SIMD - they cost the FUSED path 8 .. 20 % through
// register allocation alone (spills moved into the forward rounds; same-box A/B, round 5), although a launch that
// shares nothing never executes them.  The callee rebuilds its own Wave from the device copy of the Batch
// (Batch::self) and the wave's LDS; what it exchanges with the caller goes through private memory.
#if defined(DIRECT_EMULATE)
#define DDP_COLD inline
template <typename T>
using LdsPtr = T*;
#else
#define DDP_COLD __device__ __attribute__((noinline))
template <typename T>
using LdsPtr = __attribute__((address_space(3))) T*;
#endif
template <typename Real, typename St, int RPL, bool SHARE = false>
struct Wave;
template <typename Real, typename St, int RPL>
DDP_COLD void front_cold(const Batch<St>* Bg, LdsPtr<WaveLds<Real, St, RPL>> lds, int b, int N, int tag, int cur, int infeas, double mu);
template <typename Real, typename St, int RPL>
DDP_COLD void back_cold(const Batch<St>* Bg, LdsPtr<WaveLds<Real, St, RPL>> lds, int b, int N, void* io);

// ------------------------------------------------------------------------------------------------
// Real = arithmetic type of the per-row / roll-out work, St = storage type of everything in HBM
// (the dtype of the C-ABI), Acc (double) = type of the condensed system.
// SHARE: the instantiation can share its backward sweeps with helper waves (bwd_knot MODE 1 / 2, the claim protocol).  The
// hot kernels exist WITH and WITHOUT it (direct_ddp.hip): the kernel sits on the register cliff of three waves per
// SIMD, and the mere presence of the protocol's few live values costs the fused path 7 % (spills move into the forward
// rounds; same-box A/B, round 5) - launches that have no idle waves run the instantiation without it.
template <typename Real, typename St, int RPL, bool SHARE>
struct Wave {
  typedef WaveLds<Real, St, RPL> Lds;
  const Batch<St>& B;
  Lds& L;
  TrajState& st;
  int b;  // trajectory
  int N;  // segments
#if defined(DDP_TIMING) && !defined(DIRECT_EMULATE)
  TickState tick;
#endif
#if defined(DDP_TIMELINE) && !defined(DIRECT_EMULATE)
  int tl_split_ = 0;
#endif

  DDP_DEV Wave(const Batch<St>& batch, Lds& lds, int traj) : B(batch), L(lds), st(lds.st), b(traj), N(0) {}

  // LDS layouts of the assembly operands of the backward sweep.  The index a consumer's inner loop runs over is the
  // fastest one, so that two neighbours come with one 16-byte load (ld2).
  static constexpr int kHRS = 21, kHR0 = 88, kHzx = kHR0 + 10 * kHRS, kHdump = kHzx + 9;  // WaveLds: HR, Hzx, hdump in doubles from Hxx[0]
  static constexpr int we_idx(int row, int i) { return i * 18 + row; }   // We: 15 control rows + 3 rows of Z, 6 coefficients
  static constexpr int sp_idx(int cp, int e) { return e * 6 + cp; }      // Sp: 6 position control points, 6 entries of the symmetric 3x3
  static constexpr int dl_idx(int cp9, int d) { return d * 10 + cp9; }   // dl: 9 velocity / acceleration control points, 3 axes
  static constexpr int ch_idx(int row, int d) { return d * 18 + row; }   // Sd, hh: 15 control rows, 3 axes; stride 18: the 16-byte row loads of the three axes (phase H) fall on disjoint banks

  // Knot (b, k) of the [B][nmax(+1)] arrays; k may differ between lanes.
  DDP_DEV St* Xp(int buf, int k) const { return B.X[buf] + ((size_t)b * (B.nmax + 1) + k) * x_stride<St>(); }
  DDP_DEV St* Sp_(St* base, int k) const { return base + ((size_t)b * B.nmax + k) * B.ncs; }
  DDP_DEV const St* planes_(int k) const { return B.planes + ((size_t)b * B.nmax + k) * B.pmax * 4; }
  DDP_DEV int np_(int k) const { return B.n_planes[(size_t)b * B.nmax + k]; }
  DDP_DEV St* KUp(int k) const { return B.KU + ((size_t)b * B.nmax + k) * 100; }
  // The same for a WAVE-UNIFORM k (the sweeps).  The 32-bit row index is forced into an SGPR so that
  // every 64-bit address product stays on the scalar unit (under SGPR pressure `b` ends up in a VGPR and
  // each address would otherwise cost several quarter-rate v_mad_u64_u32).
  // The array bases themselves are read from the kernel arguments ONCE per sweep (set_sweep_ptrs) and made
  // opaque: the iterate buffers are selected by a run-time index (cur, cur + 1, cur + 2 mod 3), which otherwise costs a
  // scalar load from the kernarg segment - and an lgkmcnt(0) stall that also drains the LDS queue - in
  // every knot.  An opaque SGPR pair can at worst be spilled to a VGPR lane (two v_readlane to restore).
  typedef DDP_GLOBAL St GSt;
  typedef DDP_GLOBAL const St GCSt;
  typedef DDP_GLOBAL const int32_t GCInt;
  struct SweepPtrs {
    GSt *X[3], *S[3], *Y[3];  // [0] = buffer `cur`, [1], [2] = the trial buffers of the round being evaluated
    GSt *KS, *KY, *KU;
    GCSt* planes;
    GCInt* n_planes;
  };
  SweepPtrs sp;
  // buffer written by step `step` of a line search around buffer `cur`: any two consecutive steps differ
  DDP_DEV int trial_buf(int cur, int step) const {
    const int bi = cur + 1 + step % (B.nbuf - 1);
    return bi >= B.nbuf ? bi - B.nbuf : bi;
  }
  DDP_DEV void set_sweep_ptrs(int cur, int t0 = -1, int t1 = -1) {
    for (int i = 0; i < 3; i++) {
      int bi = i == 0 ? cur : (i == 1 ? t0 : t1);
      if (bi < 0) bi = cur;
      bi = DDP_UNIFORM_I(bi);
      sp.X[i] = (GSt*)uniform_ptr(B.X[bi]);
      sp.S[i] = (GSt*)uniform_ptr(B.S[bi]);
      sp.Y[i] = (GSt*)uniform_ptr(B.Y[bi]);
      DDP_OPAQUE_S(sp.X[i]);
      DDP_OPAQUE_S(sp.S[i]);
      DDP_OPAQUE_S(sp.Y[i]);
    }
    // (uniform_ptr: in the out-of-line halves of a shared sweep the Batch is read from memory, and the compiler does not
    // take every one of these loads for wave-uniform; on kernel arguments it folds away)
    sp.KS = (GSt*)uniform_ptr(B.KS); sp.KY = (GSt*)uniform_ptr(B.KY); sp.KU = (GSt*)uniform_ptr(B.KU);
    sp.planes = (GCSt*)uniform_ptr(B.planes);
    sp.n_planes = (GCInt*)uniform_ptr(B.n_planes);
    DDP_OPAQUE_S(sp.KS);
    DDP_OPAQUE_S(sp.KY);
    DDP_OPAQUE_S(sp.KU);
    DDP_OPAQUE_S(sp.planes);
    DDP_OPAQUE_S(sp.n_planes);
  }
  DDP_DEV size_t rowU(int k) const { return (size_t)(unsigned)DDP_UNIFORM_I(b * B.nmax + k); }
  // sel: 0 = the current iterate buffer, 1 / 2 = the trial buffers
  DDP_DEV GSt* XpU(int sel, int k) const { return sp.X[sel] + (size_t)(unsigned)DDP_UNIFORM_I(b * (B.nmax + 1) + k) * x_stride<St>(); }
  DDP_DEV GSt* SpU(GSt* base, int k) const { return base + rowU(k) * B.ncs; }
  DDP_DEV GCSt* planesU(int k) const { return sp.planes + rowU(k) * (B.pmax * 4); }
  DDP_DEV int npU(int k) const { return sp.n_planes[rowU(k)]; }
  DDP_DEV GSt* KUpU(int k) const { return sp.KU + rowU(k) * 100; }
  // Knot-record access.  With float storage every entry is an unevaluated hi + lo pair (words a and 19 + a), see
  // x_stride().  pair_round() is the value such a pair holds: what "rounded to the storage type" means for the iterate.
  template <typename Ptr>
  DDP_DEV Real ldx(Ptr rec, int a) const {
    Real v = (Real)rec[a];
    if (sizeof(St) < sizeof(double)) v += (Real)rec[19 + a];
    return v;
  }
  template <typename Ptr>
  DDP_DEV void stx(Ptr rec, int a, Real v) const {
    const St hi = (St)v;
    rec[a] = hi;
    if (sizeof(St) < sizeof(double)) rec[19 + a] = (St)(v - (Real)hi);
  }
  DDP_DEV Real pair_round(Real v) const {
    if (sizeof(St) >= sizeof(double)) return v;
    const St hi = (St)v;
    return (Real)hi + (Real)(St)(v - (Real)hi);
  }

  // Software prefetch of the next knot's HBM data into registers: the serial knot recursion would
  // otherwise expose one full memory latency per knot (nothing else is in flight in this wave).
  // Nothing here may WAIT on a load it has just issued (no arithmetic on the loaded words, no
  // load-dependent branch): every access is an unconditional load from a clamped, always valid index
  // and the words stay raw until commit().
  // Per row both sweeps read the slack s (and the dual y in infeasible mode) and nothing else: the slack / dual
  // gains ks, ky, Ks, Ky of the reference (DDP:565-575, 610-614) never reach HBM on the hot path - see fwd R.
  // Float storage halves the prefetch registers: its instantiations can afford to gather the operands of all rows (and
  // the gains of phase D) in ONE batch of loads; with double storage the same batches spill inside the sweeps, and a
  // spill reload waits for every outstanding load, the HBM prefetch included
  static constexpr bool kWide = sizeof(St) < sizeof(double) && RPL <= 2;
  // Field widths of the packed row descriptor: rows fit eight bits up to four row slots per lane (6 P + 55 <= 255);
  // the kernels with five to eight slots (polytopes of 34 .. 76 planes: 6 P + 55 <= 511) take a ninth bit from a0 (< 64),
  // those with ten to fourteen (up to 128 planes: 6 P + 55 <= 823) a tenth.
  static constexpr int kRB = RPL > 8 ? 10 : (RPL > 4 ? 9 : 8);
  // Row slots a knot with P planes really uses: position rows fill slots 0 .. ceil(6 P / 64) - 1, the last slot always
  // holds the 55 velocity / acceleration / T_min rows.  The kernels with many slots skip the others (a wave-uniform
  // branch per slot): a corridor whose widest polytope has 60 planes mostly consists of polytopes with 15 - 20
  // (DESIGN.md section 7d), and every slot costs its loads and ~40 instructions per row phase.
  static DDP_DEV bool slot_on(int slot, int P) { return RPL <= 2 || slot == RPL - 1 || 64 * slot < 6 * P; }
  static constexpr int kRMask = (1 << kRB) - 1, kAMask = (1 << (16 - kRB)) - 1;
  static constexpr int kNPL = (4 * Lds::kPMax + 63) / 64 > 2 ? (4 * Lds::kPMax + 63) / 64 : 2;  // plane words per lane
  struct Pre {  // held in the storage type: half the registers in DIRECT_F32
    St zh, zl, pl[kNPL], s[RPL], y[RPL], ku[2];
  };
  DDP_DEV void prefetch(Pre& p, int* pkv, int pk_valid, int lane, int buf, int k, int P, bool fwd, int infeas) const {
    (void)buf;

    GCSt* rec = XpU(0, k);
    p.zh = rec[lane < 19 ? lane : 18];
    p.zl = (sizeof(St) < sizeof(double)) ? rec[19 + (lane < 19 ? lane : 18)] : (St)0;  // see ldx()
    GCSt* pk = planesU(k);
    const int pend = 4 * P - 1;
    for (int i = 0; i < kNPL; i++) p.pl[i] = pk[lane + 64 * i < pend ? lane + 64 * i : pend];
    GCSt* sk = SpU(sp.S[0], k);
    GCSt* yk = SpU(sp.Y[0], k);
    for (int i = 0; i < RPL; i++) {
      // the knot's row descriptors: computed once, carried to its row phases; they only depend on P,
      // so a run of knots with the same plane count (every free-space corridor) reuses them
      if (!pk_valid) pkv[i] = row_pack(i, lane, P);
      if (!slot_on(i, P)) continue;
      const int r = (pkv[i] & kRMask) - 1;
      const int rc = r >= 0 ? r : 0;  // rows that do not exist read row 0; their results are masked
      p.s[i] = sk[rc];
      if (infeas) p.y[i] = yk[rc];
    }
    if (fwd) {
      GCSt* ku = KUpU(k);
      p.ku[0] = ku[lane];
      p.ku[1] = ku[lane + 64 < 100 ? lane + 64 : 99];
    }
  }
  DDP_DEV void commit(const Pre& p, int lane, int P, bool fwd) {
    Real z = (Real)p.zh;
    if (sizeof(St) < sizeof(double)) z += (Real)p.zl;
    if (lane < 19) L.z[lane] = z;
    for (int i = 0; i < kNPL; i++)
      if (lane + 64 * i < 4 * P) L.pl[lane + 64 * i] = (Real)p.pl[i];
    if (fwd) {
      L.KUr[lane] = (Real)p.ku[0];
      if (lane + 64 < 100) L.KUr[lane + 64] = (Real)p.ku[1];
    }
  }

  // The prefetched words are waited for HERE (the compiler places the s_waitcnt vmcnt where a loaded register is first
  // read, and the empty asm is such a read): called BEFORE a knot's global stores are issued.  Loads and stores retire
  // through one in-order counter: a wait for the prefetch that comes after the stores - at the top of the next knot, where
  // the registers are consumed - is a wait for the stores' acknowledgements as well, every knot (DDP_PREFETCH_PIN=0: off).
  DDP_DEV void pin_prefetch(Pre& p, bool fwd, int infeas) const {
#if !defined(DIRECT_EMULATE) && !defined(DDP_NO_PREFETCH_PIN)
    DDP_PIN(p.zh);
    if (sizeof(St) < sizeof(double)) DDP_PIN(p.zl);
#pragma unroll
    for (int i = 0; i < kNPL; i++) DDP_PIN(p.pl[i]);
#pragma unroll
    for (int i = 0; i < RPL; i++) DDP_PIN(p.s[i]);
    // (the dual rows of infeasible mode are not pinned: a conditional pin makes the compiler wait right behind the load,
    // an unconditional one pushes spills into the forward rounds - infeasible sweeps keep their wait at the top of the knot)
    (void)infeas;
    if (fwd) {
      DDP_PIN(p.ku[0]);
      DDP_PIN(p.ku[1]);
    }
#else
    (void)p; (void)fwd; (void)infeas;
#endif
  }

  DDP_DEV void load_state() {
    const TrajState* g = &B.st[b];
    LANES {
      const int* src = (const int*)g;
      int* dst = (int*)&L.st;
      for (int e = lane; e < (int)(sizeof(TrajState) / 4); e += 64) dst[e] = src[e];
    }
    WSYNC();
    N = DDP_UNIFORM_I(L.st.nseg);
  }
  DDP_DEV void store_state() {
    WSYNC();
    TrajState* g = &B.st[b];
    LANES {
      const int* src = (const int*)&L.st;
      int* dst = (int*)g;
      for (int e = lane; e < (int)(sizeof(TrajState) / 4); e += 64) dst[e] = src[e];
    }
  }

  // one-time LDS tables
  DDP_DEV void init_tables() {
    const int mv = B.k.minvo ? 1 : 0;
    LANES {
#pragma unroll 1
      for (int e = lane; e < 90; e += 64) {
        int cr = e / 6, i = e % 6;
        double eps = (i == 2) ? 0.5 : 1.0;  // Ek_inv = {1,1,1/2} (DDP:101-103), 1 for the u part
        double v = kValueTab[mv][cr][i];
        double d = kDtTab[cr][i];
        if (B.k.exact_dt) d = (double)(i - ctrl_off(cr)) * v;  // non-parity: exact d/dT
        L.WbE[e] = (Real)(v * eps);
        L.WdE[e] = (Real)(d * eps);
      }
      if (lane < 18) {  // [F|G] (DDP:862-871) and [F'|G'] (DDP:930-935): coefficient and exponent of T
        int c = lane / 6, i = lane % 6;
        double hc, hpc;
        int he, hpe;
        if (i < 3) {
          int e = i - c;
          hc = (e < 0) ? 0.0 : (e == 2 ? 0.5 : 1.0);
          he = e < 0 ? 0 : e;
          hpc = (e <= 0) ? 0.0 : 1.0;
          hpe = e <= 1 ? 0 : e - 1;
        } else {
          int a = i - 3, e = 3 + a - c;
          double g0 = (c == 0) ? 1.0 : (c == 1 ? (double)(3 + a) : (double)((3 + a) * (2 + a)));
          hc = g0;
          he = e;
          hpc = g0 * (double)e;
          hpe = e - 1;
        }
        (void)he;
        (void)hpe;  // the exponents are i - row and i - row - 1: the table's own structure
        L.WbE[90 + lane] = (Real)hc;   // rows 15..17 of the value table: [F | G]
        L.WdE[90 + lane] = (Real)hpc;  // and [F' | G']
      }
      if (lane < 9) {  // DDP:991-999: Rc[a][a'] = c_a c_a' / (a+a'+1), c_a = (a+1)(a+2)(a+3)
        int a = lane / 3, a2 = lane % 3;
        double ca = (a + 1) * (a + 2) * (a + 3), cb = (a2 + 1) * (a2 + 2) * (a2 + 3);
        L.Rc[lane] = (Real)(ca * cb / (double)(a + a2 + 1));
        L.Rc[9] = (Real)0;
        L.WbE[108 + a * 6 + a2] = (Real)0;  // rows 18..20: [0 0 0 | Rc[a][:]], exponent i - (2 - a) = a + a2 + 1
        L.WbE[108 + a * 6 + 3 + a2] = (Real)(ca * cb / (double)(a + a2 + 1));
        L.WdE[108 + a * 6 + a2] = (Real)0;  // d/dT: (a + a2 + 1) Rc, one power less
        L.WdE[108 + a * 6 + 3 + a2] = (Real)(ca * cb / (double)(a + a2 + 1)) * (Real)(a + a2 + 1);
      }
      {  // lt[0..3], lt[6..8]: phase H (18x18 block of the condensed system); lt[4]: value recursion of phase R2
        // All fields are BYTE offsets, ready for the address: a bit-field extract each, no shift, no multiply.
        const int l62 = lane < 63 ? lane : 62;  // lane 63 redoes lane 62
        const int pr = l62 / 3, d = l62 % 3;
        const int i = (pr >= 6) + (pr >= 11) + (pr >= 15) + (pr >= 18) + (pr >= 20);
        const int i2 = i + pr - (6 * i - (i * (i - 1)) / 2);
        const int p = 3 * i + d;
        const int hasq = i >= 3 ? 1 : 0;
        const int e8 = (int)sizeof(Real), a8 = (int)sizeof(Acc);
        // byte offsets of We[.][i], We[.][i2], dl[.][d]; index of the jerk Gram coefficient (Rc[9] = 0 for pairs without one)
        L.lt[0][lane] = (we_idx(0, i) * e8) | ((we_idx(0, i2) * e8) << 10) | ((dl_idx(0, d) * a8) << 20) |
                        ((hasq ? (i - 3) * 3 + (i2 - 3) : 9) << 28);
        for (int t = 0; t < 3; t++) {  // the lane walks the column axes d2 = d, d + 1, d + 2 (mod 3): the first one is its own
          const int d2 = (d + t) % 3;
          const int lo = d < d2 ? d : d2, hi = d < d2 ? d2 : d;
          const int sidx = lo * 3 - (lo * (lo - 1)) / 2 + (hi - lo);  // xx,xy,xz,yy,yz,zz
          const int q = 3 * i2 + d2;
          // entry (p, q), p <= q < 18: Hxx keeps its upper triangle only; rows >= 9 live in HR, both triangles (phase C
          // reads whole rows), and the Hxu block as its transpose inside those rows (see WaveLds)
          int o1 = q < 9 ? p * 9 + q : (p < 9 ? kHR0 + (q - 9) * kHRS + p : kHR0 + (p - 9) * kHRS + q);
          int o2 = (q < 9 || p < 9) ? o1 : kHR0 + (q - 9) * kHRS + p;
          // i == i2: the lower triangle of the diagonal block belongs to the lane of the other axis - this lane's copy
          // goes to a slot nobody reads
          if (p > q) o1 = o2 = kHdump;
          // bits 12..14 (t == 0): the power of T of the jerk Gram term; bits 28..31: Sp[.][sidx] in units of 16 bytes
          L.lt[1 + t][lane] = (o1 * a8) | ((t == 0 && hasq ? i + i2 - 5 : 0) << 12) | ((o2 * a8) << 16) |
                              ((sp_idx(0, sidx) * a8 / 16) << 28);
          L.lt16[2 + t][lane] = (unsigned short)((d * 19 + q) * a8);
        }
        const int l45 = lane < 45 ? lane : 44;  // upper triangle (a, c2 >= a) of the 9x9 value matrix
        int a = 0, rem = l45;
        while (rem >= 9 - a) {
          rem -= 9 - a;
          a++;
        }
        L.lt16[0][lane] = (unsigned short)(a | ((a + rem) << 4));
        {  // lt[5]: phase S roles (lanes >= 54 redo lane 53)
          const int l54 = lane < 54 ? lane : 53;
          const bool isS = l54 < 36;
          // consecutive lanes take consecutive control points: their stores fall on consecutive LDS words (entry- / axis-
          // fastest roles stored with a stride of 6 / 16 doubles, a three-way bank conflict on the store path)
          const int e = l54 / 6, j = isS ? l54 % 6 : (l54 - 36) % 6;
          const int d0 = isS ? ((e < 3) ? 0 : (e < 5 ? 1 : 2)) : (l54 - 36) / 6;
          const int d1 = isS ? ((e < 3) ? e : (e < 5 ? e - 2 : 2)) : 0;
          const int dst = isS ? sp_idx(j, e) : (int)(&L.hh[0] - &L.Sp[0]) + ch_idx(j, d0);  // hh follows Sp in the same struct
          L.lt16[1][lane] = (unsigned short)(d0 | (d1 << 2) | ((isS ? 1 : 0) << 4) | (j << 5) | (dst << 8));
        }
        if (lane < 5) L.ones[lane] = (Real)1;
        if (lane < 5) L.z[19 + lane] = (Real)0;  // read (against zero weights) by phase T2's clamp-free term loop
        {  // lt[7]: phase T2 (lane 63 redoes lane 62)
          const int l62 = lane < 63 ? lane : 62;
          const int cr = l62 / 3, d = l62 % 3, o = ctrl_off(cr);
          // bits 19..28: byte offset of We[cr][o], the first of the 6 - o entries of We this lane ALSO produces (the
          // products weight x power of T of rows 0..17: phase T2 stores them, there is no separate table phase); bits
          // 29..31: how many of them the lane stores (axis 0 of rows 0..17 only)
          const int nst = (d == 0 && cr < 18) ? 6 - o : 0;
          L.lt[7][lane] = ((cr * 6 + o) * (int)sizeof(Real)) | (((3 * o + d) * (int)sizeof(Real)) << 10) | (o << 17) |
                          ((we_idx(cr < 18 ? cr : 17, o) * (int)sizeof(Real)) << 19) | (nst << 29);
        }
        for (int pass = 0; pass < 3; pass++) {  // lt[4..6]: the VZ passes of phase R1
          const int e = (lane + 64 * pass < 162) ? lane + 64 * pass : 161;
          const int a = e / 18, q = e % 18;
          const int i = q / 3, d = q % 3;
          L.lt[4 + pass][lane] = ((a * 9 + d) * (int)sizeof(Acc)) | ((we_idx(15, i) * (int)sizeof(Real)) << 10) |
                                 (((a * 19 + q) * (int)sizeof(Acc)) << 20);
        }
      }
      if (lane < 5) {  // pseudo-planes (n, o) of the non-plane rows: +/-v - vmax, +/-a - amax, -T + 0.3 (DDP:1237-1279)
        Real* q = &L.pl[4 * Lds::kPMax + 4 * lane];
        q[0] = (Real)0;
        q[1] = (Real)0;
        q[2] = (lane == 0 || lane == 2) ? (Real)1 : (Real)-1;
        q[3] = lane < 2 ? -(Real)B.k.max_vel : (lane < 4 ? -(Real)B.k.max_acc : (Real)0.3);
      }
    }
    WSYNC();
  }

  // ---- shared pieces ---------------------------------------------------------------------------
  // Row descriptor of (slot, lane) for a knot with P planes, packed into one word so that it can be
  // computed ONCE per knot and carried in a register: (r + 1) | a0 << 8 | (plane index in L.pl) << 16.
  // Non-plane rows point at the pseudo-planes behind the knot's planes, so every row is a "plane row".
  DDP_DEV int row_pack(int slot, int lane, int P) const {
    const bool last = (slot == RPL - 1);  // a constant once the slot loop is unrolled
    const int rp = last ? 64 * (RPL - 1) + lane - 55 : lane + 64 * slot;
    const bool pv = rp >= 0 && rp < 6 * P;
    const int rq = pv ? rp : 0;
    const int j = (int)(DDP_UMUL24((unsigned)rq, kInvP[P]) >> kInvPShift);  // rq / P: the control point
    const int q = rq - (int)DDP_UMUL24((unsigned)j, (unsigned)P);   // the plane
    int pk = (pv ? rp + 1 : 0) | ((3 * j) << kRB) | (q << 16);
    if (last) {  // lanes 0..54: velocity (30), acceleration (24), T_min (1) rows  (DDP:1236-1238, 1274-1279)
      const bool isv = lane < 30, isa = lane < 54;
      const int l2 = lane - 30;
      const int a0v = 16 + (lane < 15 ? lane : lane - 15);  // 18 + . - 2: the bounded value is read through n2
      const int a0a = 31 + (l2 < 12 ? l2 : l2 - 12);         // 33 + . - 2
      const int a0 = isv ? a0v : (isa ? a0a : 43);
      const int pp = isv ? (lane < 15 ? 0 : 1) : (isa ? (l2 < 12 ? 2 : 3) : 4);
      const int po = (6 * P + lane + 1) | (a0 << kRB) | ((Lds::kPMax + pp) << 16);
      // lanes 55..63 without a position row of their own alias the T_min row (lane 54's operands: the same LDS addresses,
      // i.e. a broadcast) instead of position row 0, whose val[0..2] and plane 0 sit one bank period away from this slot's
      // val[32..34] and pseudo-plane 4 - a bank conflict in every row-operand load of the slot (tools/lds_trace)
      const int pal = (43 << kRB) | ((Lds::kPMax + 4) << 16);
      pk = lane < 55 ? po : (pv ? pk : pal);
    }
    return pk;
  }
  DDP_DEV RowK<Real> row_unpack(int pk) const {
    RowK<Real> k;
    const Real* n = &L.pl[4 * (pk >> 16)];
    k.n0 = n[0];
    k.n1 = n[1];
    k.n2 = n[2];
    k.o = n[3];
    k.r = (pk & kRMask) - 1;
    k.a0 = (pk >> kRB) & kAMask;
    return k;
  }
  DDP_DEV RowK<Real> row_slot(int slot, int lane, int P) const { return row_unpack(row_pack(slot, lane, P)); }
  // The row phases of the sweeps gather the operands of ALL their rows first (plane, three entries of the vector the
  // row multiplies) and compute afterwards: one LDS round trip per phase instead of two or three per row.
  struct Row3 {
    Real a, b, c;
  };
  DDP_DEV RowK<Real> row_unpack2(int pk) const {  // the plane as two 16-byte loads
    RowK<Real> k;
    const Real* n = &L.pl[4 * (pk >> 16)];
    if (kWide) {
      ld2(n, k.n0, k.n1);
      ld2(n + 2, k.n2, k.o);
    } else {  // (16-byte loads want four consecutive registers: one more constraint where there are none to spare)
      k.n0 = n[0];
      k.n1 = n[1];
      k.n2 = n[2];
      k.o = n[3];
    }
    k.r = (pk & kRMask) - 1;
    k.a0 = (pk >> kRB) & kAMask;
    return k;
  }
  DDP_DEV Row3 row_ops(const Real* A, int pk) const {
    const Real* p = A + ((pk >> kRB) & kAMask);
    Row3 v;
    v.a = p[0];
    v.b = p[1];
    v.c = p[2];
    return v;
  }
  // explicit fused operations in a fixed order: left to the compiler, the contraction of  n0 a + n1 b + n2 c  differs
  // between instantiations of the same source (static and ticket-scheduled kernel), and with it the last bit
  DDP_DEV Real row_dot(const RowK<Real>& k, const Row3& v) const { return fma(k.n2, v.c, fma(k.n1, v.b, k.n0 * v.a)); }
  // A_r . w
  DDP_DEV Real row_lin(const Real* A, const RowK<Real>& k) const {
    return k.n0 * A[k.a0] + k.n1 * A[k.a0 + 1] + k.n2 * A[k.a0 + 2];
  }
  // c_r (val[45] must hold T), shifted by 2e-4 unless minvo (DDP:1281-1283)
  DDP_DEV Real row_c(const Real* val, const RowK<Real>& k) const {
    return row_lin(val, k) + k.o - (Real)B.k.shift;
  }

  // control values val[cr][d] = sum_i W[cr][i] T^(i-o) C_i[d] from a knot record zz with powers tpw
  DDP_DEV Real ctrl_val(const Real* zz, const Real* tpw, int cr, int d) const {
    const int o = ctrl_off(cr);
    Real v = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
      int e = i - o;
      v += L.WbE[cr * 6 + i] * tpw[e < 0 ? 0 : e] * zz[3 * i + d];
    }
    return v;
  }
  // d val[cr][d] / dT = sum_i Wd[cr][i] T^(i-o-1) C_i[d]: the T column of the constraint Jacobian (DDP:1543-1561, quirk Q1:
  // Wd holds the MINVO tables whatever basis W holds)
  DDP_DEV Real ctrl_dval(const Real* zz, const Real* tpw, int cr, int d) const {
    const int o = ctrl_off(cr);
    Real v = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
      int e = i - o - 1;
      v += L.WdE[cr * 6 + i] * tpw[e < 0 ? 0 : e] * zz[3 * i + d];
    }
    return v;
  }
  // x+ component a of (F (x) I) x + (G (x) I) u  (DDP:1062-1067)
  DDP_DEV Real next_x(const Real* zz, const Real* tpw, int a) const {
    const int c = a / 3, d = a % 3;
    Real acc = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) acc += L.WbE[90 + c * 6 + i] * tpw[i < c ? 0 : i - c] * zz[3 * i + d];
    return acc;
  }
  // u_a[d] * (R u)_a[d]; the nine of them sum to u'Ru  (DDP:1294-1305)
  DDP_DEV Real jerk_part(const Real* zz, const Real* tpw, int a9) const {
    const int a = a9 / 3, d = a9 % 3;
    Real acc = 0;
#pragma unroll
    for (int a2 = 0; a2 < 3; a2++) acc += L.Rc[a * 3 + a2] * tpw[a + a2 + 1] * zz[9 + 3 * a2 + d];
    return acc * zz[9 + a9];
  }
  DDP_DEV double knot_cost(Real T, const Real* qp) const {  // q from the nine partial products in qp
    Real acc = 0;
#pragma unroll
    for (int a = 0; a < 9; a++) acc += qp[a];
    Real q = (Real)0.5 * (Real)B.k.w_snap * acc;
    if (B.k.time_power == 2) q += (Real)0.5 * T * (Real)B.k.w_time * T;
    else q += (Real)0.5 * (Real)B.k.w_time * T;
    return (double)q;
  }
  DDP_DEV double terminal_sq() {  // |x_N - x_d|^2 with x_N - x_d in L.z[0..8]
    Real acc = 0;
#pragma unroll
    for (int a = 0; a < 9; a++) acc += L.z[a] * L.z[a];
    return (double)acc;
  }

  // observability (direct_ddp_last_launch_info): knots executed by a backward sweep (which = 0) / by the trials of a
  // forward round (which = 1); one atomic per sweep
  DDP_DEV void count_visits(int which, int n) const {
#if !defined(DIRECT_EMULATE)
    if (B.visits != nullptr && threadIdx.x == 0) atomicAdd(&B.visits[which], (unsigned long long)n);
#else
    (void)which; (void)n;
#endif
  }

  // ---- evaluation sweep over one iterate buffer: costs, log / error sums, violation count.
  // With do_roll it also propagates x (initialroll, DDP:1608-1620).
  DDP_DEV void eval_sweep(int buf, bool do_roll) {
    PLV(LogProd<Real>, plog);
    PLV(Real, slog);
    PLV(Real, serr);
    PLV(int, nviol);
    PLV(int, npos);
    LANES { LV(plog).init(); LV(serr) = 0; LV(nviol) = 0; LV(npos) = 0; }
    const int infeas = DDP_UNIFORM_I(st.infeas);
    double qsum = 0.0;
    int neg = 0;
    // The next knot's record and planes are loaded a knot ahead (registers), the dual rows of a knot at its top: the
    // sweep used to expose a global-memory round trip per knot, and with do_roll it re-read from HBM the state it had
    // just written (k_begin was 1.7 % of the benchmark's step).
    PLV(Real, zr);
    PLA(St, plr, kNPL);
    int Pn = np_(0);
    LANES {
      LV(zr) = ldx(Xp(buf, 0), lane < 19 ? lane : 18);
      const int pend = 4 * Pn - 1;
      for (int i = 0; i < kNPL; i++) LV(plr)[i] = planes_(0)[lane + 64 * i < pend ? lane + 64 * i : pend];
    }
    for (int k = 0; k < N; k++) {
      const int P = Pn;
      PLA(Real, yv, RPL);
      LANES {
        // x_k of a roll is what the previous knot produced - rounded as the store / load pair would round it
        const bool rolled = do_roll && k > 0 && lane < 9;
        if (lane < 19) L.z[lane] = rolled ? pair_round(L.ft[0].xnx[lane]) : LV(zr);
        for (int i = 0; i < kNPL; i++)
          if (lane + 64 * i < 4 * P) L.pl[lane + 64 * i] = (Real)LV(plr)[i];
        if (infeas) {
          const St* yk = Sp_(B.Y[buf], k);
          for (int i = 0; i < RPL; i++) {
            if (!slot_on(i, P)) continue;
            const int r = (row_pack(i, lane, P) & kRMask) - 1;
            LV(yv)[i] = (Real)yk[r >= 0 ? r : 0];
          }
        }
      }
      if (k + 1 < N) {
        Pn = np_(k + 1);
        LANES {
          LV(zr) = ldx(Xp(buf, k + 1), lane < 19 ? lane : 18);
          const int pend = 4 * Pn - 1;
          for (int i = 0; i < kNPL; i++) LV(plr)[i] = planes_(k + 1)[lane + 64 * i < pend ? lane + 64 * i : pend];
        }
      }
      WSYNC();
      const Real T = L.z[18];
      if (T < 0) neg = 1;
      LANES { if (lane < 8) L.tp[lane] = powi(T, lane); }
      WSYNC();
      LANES {
        if (lane < 45) L.val[lane] = ctrl_val(L.z, L.tp, lane / 3, lane % 3);
        else if (lane < 54) { if (do_roll) L.ft[0].xnx[lane - 45] = next_x(L.z, L.tp, lane - 45); }
        else if (lane < 63) L.ft[0].qp[lane - 54] = jerk_part(L.z, L.tp, lane - 54);
        else L.val[45] = T;
      }
      WSYNC();
      qsum += knot_cost(T, L.ft[0].qp);
      LANES {
        for (int i = 0; i < RPL; i++) {
          if (!slot_on(i, P)) continue;
          const RowK<Real> rk = row_slot(i, lane, P);
          const int r = rk.r;
          if (r >= 0) {
            Real c = row_c(L.val, rk);
            if (infeas) {
              Real y = LV(yv)[i];
              LV(plog).mul(y);
              LV(serr) += fabs(c + y);
            } else {
              LV(plog).mul(-c);
            }
            if (c >= (Real)2.0e-4) LV(nviol)++;
            if (c > (Real)0) LV(npos)++;
          }
        }
        LV(plog).norm();
        if (do_roll && lane < 9) stx(Xp(buf, k + 1), lane, L.ft[0].xnx[lane]);
      }
      WSYNC();
    }
    LANES {
      if (lane < 9) L.z[lane] = ldx(Xp(buf, N), lane) - (Real)B.xd[(size_t)b * 9 + lane];
    }
    WSYNC();
    const double pterm = terminal_sq();  // DDP:1289-1292
    WSYNC();
    LANES { LV(slog) = LV(plog).value(); }
    st.costq = qsum;
    st.cost = qsum + 0.5 * B.k.w_term * pterm;
    st.sumlog = WAVE_SUM_D(slog);
    st.errsum = WAVE_SUM_D(serr);
    st.viol = WAVE_SUM_I(nviol);
    st.npos = WAVE_SUM_I(npos);
    st.neg_time = neg;
  }

  // resetfilter (DDP:1636-1662) from the sums of the current iterate
  DDP_DEV void reset_filter() {
    double logcost = st.cost - st.mu * st.sumlog;
    double err = 0.0;
    if (st.infeas) {
      err = st.errsum;
      if (err < B.k.tol) err = 0.0;
    }
    st.logcost = logcost;
    st.err = err;
    double* f = B.filt + (size_t)b * B.fcap * 2;
    f[0] = logcost;  // wave-uniform store: every lane writes the same value
    f[1] = err;
    st.nfilter = 1;
    st.step = 0;
    st.fp_failed = 0;
  }

  // ---- line initialisation (DDP:194-248): per segment the quintic that joins the polytope seeds at
  // rest, its duration doubled (at most 5 times) until every constraint of the segment is negative.
  DDP_DEV void line_init() {
    const St* T0 = B.T0 + (size_t)b * B.nmax;
    for (int l = 0; l < N; l++) {
      const int P = np_(l);
      Real pa[3], pn[3];
      for (int d = 0; d < 3; d++) {
        pa[d] = (l == 0) ? (Real)B.x0[(size_t)b * 9 + d] : (Real)B.seeds[((size_t)b * B.nmax + l) * 3 + d];
        pn[d] = (l == N - 1) ? (Real)B.xd[(size_t)b * 9 + d] : (Real)B.seeds[((size_t)b * B.nmax + l + 1) * 3 + d];
      }
      LANES { for (int e = lane; e < 4 * P; e += 64) L.pl[e] = planes_(l)[e]; }
      Real Tk = (Real)T0[l];
      int vio = 1, cnt = 0;
      while (vio && cnt <= 4) {
        const Real Tk2 = Tk * Tk, Tk3 = Tk2 * Tk, Tk4 = Tk3 * Tk, Tk5 = Tk4 * Tk;
        // u = G^-1 (x_next - F x) with x = [pa; 0; 0], x_next = [pn; 0; 0]: only the first column of G^-1 acts
        const Real g0 = (Real)10.0 / Tk3, g1 = (Real)-15.0 / Tk4, g2 = (Real)6.0 / Tk5;
        LANES {
          if (lane < 19) {
            const int a = lane / 3, d = lane % 3;
            Real v = 0;
            if (lane < 3) v = pa[d];
            else if (lane >= 9 && lane < 18) v = (a == 3 ? g0 : (a == 4 ? g1 : g2)) * (pn[d] - pa[d]);
            else if (lane == 18) v = Tk;
            L.z[lane] = v;
          }
          if (lane < 8) L.tp[lane] = powi(Tk, lane);
        }
        WSYNC();
        LANES {
          if (lane < 45) L.val[lane] = ctrl_val(L.z, L.tp, lane / 3, lane % 3);
          else if (lane == 63) L.val[45] = Tk;
        }
        WSYNC();
        PLV(int, bad);
        LANES {
          LV(bad) = 0;
          for (int i = 0; i < RPL; i++) {
            if (!slot_on(i, P)) continue;
            const RowK<Real> rk = row_slot(i, lane, P);
            if (rk.r >= 0 && !(row_c(L.val, rk) < (Real)0)) LV(bad) = 1;
          }
        }
        if (WAVE_ANY(bad)) {
          Tk = (Real)2 * Tk;
          cnt++;
        } else {
          vio = 0;
        }
        WSYNC();
      }
      // the coefficients of the last trial stay, the duration is the (possibly doubled once more) Tk
      LANES {
        if (lane >= 9 && lane < 18) stx(Xp(0, l), lane, L.z[lane]);
        if (lane == 18) stx(Xp(0, l), 18, Tk);
      }
      WSYNC();
    }
  }

  // ---- setup (DDP:104-286) -----------------------------------------------------------------------
  DDP_DEV void begin() {
    // Sizes are validated HERE, on the device: with device-resident inputs (DIRECT_MEM_DEVICE) the host
    // never sees them, and an n_seg / n_planes outside the handle's configuration would index past every
    // per-trajectory array (and past kInvP / L.pl).  A bad row is finished at once with DIRECT_RTN_INVALID.
    N = B.n_seg[b];
    int bad_sizes = (N < 1 || N > B.nmax) ? 1 : 0;
    if (!bad_sizes) {
      PLV(int, badp);
      LANES {
        LV(badp) = 0;
        for (int k = lane; k < N; k += 64) {
          const int np = np_(k);
          if (np < 1 || np > B.pmax || np > Lds::kPMax) LV(badp) = 1;
        }
      }
      bad_sizes = WAVE_ANY(badp);
    }
    if (bad_sizes) {
      N = 0;
      LANES {
        int* dst = (int*)&L.st;
        for (int e = lane; e < (int)(sizeof(TrajState) / 4); e += 64) dst[e] = 0;
      }
      WSYNC();
      st.rtn = -100;  // DIRECT_RTN_INVALID
      st.done = 1;
      st.line_failed = 1;
      return;
    }
    st.nseg = N;
    st.cur = 0;
    st.infeas = B.infeas_in ? (int)B.infeas_in[b] : 0;
    st.infeas_ref = st.infeas;
    st.line_failed = 1;
    const St* T0 = B.T0 + (size_t)b * B.nmax;
    LANES {
      if (lane < 9) stx(Xp(0, 0), lane, (Real)B.x0[(size_t)b * 9 + lane]);
      // u: zero init (DDP:126-127) or the tail of the Bezier->poly row (DDP:167-193)
      for (int k = lane; k < N; k += 64) {
        St* rec = Xp(0, k);
        Real T = T0[k];
        if (B.k.zero_init || (B.init_bez == nullptr && B.init_poly == nullptr)) {
          for (int a = 0; a < 9; a++) stx(rec, 9 + a, (Real)0);
        } else if (B.init_poly != nullptr) {  // extension: monomial warm start, u = [c3; c4; c5]
          const St* row = B.init_poly + ((size_t)b * B.nmax + k) * 18;
          for (int a = 0; a < 9; a++) stx(rec, 9 + a, (Real)row[9 + a]);
        } else {
          const St* row = B.init_bez + ((size_t)b * B.nmax + k) * 18;  // [x0..x5,y0..y5,z0..z5]
          for (int i = 3; i < 6; i++)
            for (int d = 0; d < 3; d++) {
              Real acc = 0;
              for (int l = 0; l < 6; l++) acc += (Real)kBez2Mono[l][i] * (T * (Real)row[d * 6 + l]);
              stx(rec, 9 + (i - 3) * 3 + d, acc / powi(T, i));
            }
        }
        stx(rec, 18, T);
      }
    }
    if (!B.k.zero_init && B.k.line_init) line_init();
    for (int k = 0; k < N; k++) {  // s = 0.1, y = 0.01 (DDP:150-151)
      const int nc = 6 * np_(k) + 55;
      LANES {
        for (int i = 0; i < RPL; i++) {
          int r = lane + 64 * i;
          if (r < nc) {
            Sp_(B.S[0], k)[r] = (St)0.1;
            Sp_(B.Y[0], k)[r] = (St)0.01;
          }
        }
      }
    }
    WSYNC();
    eval_sweep(0, true);
    if (B.k.line_init && st.npos == 0 && st.infeas) {  // DDP:255-269: the straight lines are already feasible
      st.infeas = 0;
      eval_sweep(0, false);  // the barrier sums of the feasible mode (the roll itself is unchanged)
    }
    st.prev_cost = st.cost;
    st.nc0 = 6 * np_(0) + 55;
    st.mu = st.cost / (double)N / (double)st.nc0;  // DDP:281 (quirk Q6)
    reset_filter();
    st.reg = B.k.line_init ? 10 : 0;  // DDP:283-286
    st.bp_failed = 0;
    st.opterr = 0.0;
    st.stepsize = 0.0;
    st.rtn = 0;
    st.iter = 0;
    st.done = 0;
    st.bp_no_upd = 0;
    st.no_upd = 0;
    st.fwd_passes = 0;
    LANES {
      if (lane == 0) {  // no gains yet: bp.ku .. bp.Ky are zero (DDP:154-159)
        GainBase* g = &B.gbase[b];
        g->mu = 0.0;
        g->buf = -1;
        g->kreach = N;
      }
    }
  }

  // ---- backward sweep (DDP:440-644).  Returns 1 on success, 0 when the LLT failed. ---------------
  // kGains: also form the slack / dual gains ks, ky per row (DDP:568-571, 611) and store them (phases G and the row
  // part of R2).  Only the stepwise entry point direct_ddp_backward_pass wants them (per-pass parity tests read
  // DIRECT_FIELD_KS / KY): the forward pass eliminates them algebraically (see run_round, phase R), so the hot kernels
  // neither compute nor store nor load them - like cx, cu, Ks and Ky they are never materialised.
  //
  // One knot of the sweep is written ONCE (bwd_knot) and instantiated three ways:
  //   MODE 0  fused: the whole knot, everything between its phases in LDS / registers - what every sweep runs when it
  //           has no helper;
  //   MODE 1  front: the half that does not depend on the value function (phases L, T2, the rows of R1, S, S2 and the
  //           constraint half of H), ending in a hand-over record of six doubles per lane (BRec) instead of the condensed
  //           system - run by HELPER waves for knots of their own (bwd_front_run), and by nobody else;
  //   MODE 2  back: the value-dependent rest (VZ of R1, the Z'VZ half of H, C, R2) from such a record - run by the
  //           sweep's owner for the knots helpers have prepared.
  // The split is scheduling only: front + back perform the very operations of the fused knot in the same order (the
  // record carries the front half's values at the point where the fused code combines them with the back half's), so the
  // results are bitwise those of the fused sweep whoever computed what (tests/test_gpu_fullsize.py, tests/test_emu_parity.py).
  struct BRec {  // one knot's hand-over record as ONE LANE holds it (roles of phase H):
    Acc r[6];    //  r[0..2] the lane's three entries of the 18 x 18 block: quu -/+ A'DA (+ the velocity / acceleration rows)
  };             //  r[3]    lanes 0..17 T column, 18..35 Hz (constraint halves); 36..53 Z = [F | G] (row c, coefficient i at
                 //          36 + 6 c + i); 54..62 fT; 63 the knot's max |r|, |c + y| (opterr)
                 //  r[4]    lanes 0..53 the lane's term of the (T,T) entry's wave sum; 54: quu -/+ D_last; 55: qz - g_last
                 //  r[5]    lanes 0..53 the lane's term of Hz[T]'s wave sum
  struct BwdCtx {  // what a run of knots carries from one knot to the next
    int regi, buf, infeas, klo;  // klo: no prefetch below this knot (0; the bottom of a helper's chunk)
    Acc lam, sig;
    Real mu, wsn;
    int Pn, Pnn;   // plane counts of the next knot and the one after (see the prefetch)
    Acc emu_u;     // back: running maximum of the row errors of the knots that came through records
    PLV(Real, e_mu);  // running max |r|, |c + y| of the lane's rows (DDP:641); front: of the knot
    PLV(Acc, e_qu);   // running max |Hu[j]| of lane j < 10 (DDP:633, quirk Q10: Qu after the condensation correction)
    PLV(Pre, pre);
    PLA(int, pkn, RPL);  // row descriptors of the prefetched knot
    PLV(BRec, rc);       // back: the knot's record; front: the record being formed
  };
#ifndef DDP_KSPLIT
#define DDP_KSPLIT (SHARE && RPL <= 4)
#endif
  static constexpr bool kSplit = DDP_KSPLIT;  // the helper-assisted sweep is built for the narrow classes only (long trajectories: N >= 80)

  // -- the share protocol (wave-uniform; lane 0 performs the atomics).  Every transition of BwdShare::word is an atomic
  // RMW, so owner and helpers need no fences between them: open = add (tag << kBsCountBits), enter = add 1 and look at the
  // old tag (and back out when kBsMaxHelpers are inside already), close = and kBsCountMask and look at the old count.  Records and flags are written through (sc1) and drained (vmcnt(0))
  // before the flag, and read with sc1 loads issued only after the flag has been seen (MI355X_MICROARCH.md, hand-off forms).
  DDP_DEV BwdShare* bshare_slot() const { return (kSplit && B.bshare && B.self) ? &B.bshare[b] : nullptr; }
  DDP_DEV double* rec_ptr(int k) const { return B.brec + ((size_t)(unsigned)DDP_UNIFORM_I(b * B.nmax + k)) * kRecDoubles; }
  DDP_DEV int* flag_ptr(int k) const { return B.bflag + (size_t)(unsigned)DDP_UNIFORM_I(b * B.nmax + k); }
#if !defined(DIRECT_EMULATE)
  typedef unsigned int U4 __attribute__((ext_vector_type(4)));
  typedef double D2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  DDP_DEV void rec_store(int k, const BRec& r, int lane) const {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(rec_ptr(k), 0, kRecDoubles * 8, 0x00020000);
#pragma unroll
    for (int p = 0; p < 3; p++) {
      D2 v;
      v.x = r.r[2 * p];
      v.y = r.r[2 * p + 1];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(U4, v), rs, (p * 64 + lane) * 16, 0, 16 /* sc1 */);
    }
  }
  DDP_DEV void rec_load(int k, BRec& r, int lane) const {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(rec_ptr(k), 0, kRecDoubles * 8, 0x00020000);
#pragma unroll
    for (int p = 0; p < 3; p++) {
      const D2 v = __builtin_bit_cast(D2, __builtin_amdgcn_raw_buffer_load_b128(rs, (p * 64 + lane) * 16, 0, 16 /* sc1 */));
      r.r[2 * p] = v.x;
      r.r[2 * p + 1] = v.y;
    }
  }
  DDP_DEV void flag_set(int k, int tag) const {
    drain_stores();
    a_store(flag_ptr(k), tag);
  }
  // the flag of knot k as a per-lane value (every lane loads the same word: one transaction, nothing waits here)
  DDP_DEV int flag_peek(int k) const { return __hip_atomic_load(flag_ptr(k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  DDP_DEV int flag_wait(int k, int tag) const {
    int spins = 0;
    while (a_load(flag_ptr(k)) != tag) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > kSpinLimit) {
        proto_error();
        return 0;
      }
    }
    return 1;
  }
  static __device__ __forceinline__ unsigned long long a_add64(unsigned long long* p, unsigned long long v) {
    unsigned long long o = 0;
    if (threadIdx.x == 0) o = __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(o >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)o);
  }
  // owner: open the sweep around iterate buffer `cur`; knots [kfloor, N) are the owner's first claim.  Returns the tag.
  DDP_DEV int bs_open(BwdShare* bs, int cur, int infeas, double mu, int kfloor) {
    int seq = 0;
    if (threadIdx.x == 0) {
      seq = bs->seq + 1;
      bs->seq = seq;
      __hip_atomic_store(&bs->cur, cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&bs->infeas, infeas, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store((unsigned long long*)&bs->mu, (unsigned long long)__double_as_longlong(mu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&bs->claim, ((kClaimBias + (unsigned long long)kfloor) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    seq = __builtin_amdgcn_readfirstlane(seq) & (int)kBsTagMask;
    if (seq == 0) seq = 1;
    drain_stores();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&bs->word, (unsigned)seq << kBsCountBits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return seq;
  }
  // owner: close it and wait until no helper is inside (they finish the chunk they hold; nothing of theirs is awaited)
  DDP_DEV void bs_close(BwdShare* bs) {
    unsigned o = 0;
    if (threadIdx.x == 0) o = __hip_atomic_fetch_and(&bs->word, kBsCountMask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int inside = __builtin_amdgcn_readfirstlane((int)(o & kBsCountMask));
    int spins = 0;
    while (inside) {
      __builtin_amdgcn_s_sleep(2);
      inside = a_load((int*)&bs->word) & (int)kBsCountMask;
      if (++spins > kSpinLimit) {
        proto_error();
        break;
      }
    }
  }
  DDP_DEV int bs_tag(BwdShare* bs) const { return (int)((unsigned)a_load((int*)&bs->word) >> kBsCountBits); }
  // helper: join the open sweep of trajectory b, if there is one
  DDP_DEV int bs_enter(BwdShare* bs, int& tag, int& cur, int& infeas, double& mu) {
    unsigned o = 0;
    if (threadIdx.x == 0) o = __hip_atomic_fetch_add(&bs->word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tag = __builtin_amdgcn_readfirstlane((int)(o >> kBsCountBits));
    if (!tag || __builtin_amdgcn_readfirstlane((int)(o & kBsCountMask)) >= kBsMaxHelpers) {  // closed, or full
      tag = 0;
      if (threadIdx.x == 0) __hip_atomic_fetch_add(&bs->word, ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the iterate: released by whoever wrote it before the owner's ticket was ready
    int cv = 0, iv = 0, ml = 0, mh = 0;
    if (threadIdx.x == 0) {
      cv = __hip_atomic_load(&bs->cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      iv = __hip_atomic_load(&bs->infeas, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long m = __hip_atomic_load((unsigned long long*)&bs->mu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ml = (int)m;
      mh = (int)(m >> 32);
    }
    cur = __builtin_amdgcn_readfirstlane(cv);
    infeas = __builtin_amdgcn_readfirstlane(iv);
    mu = __hiloint2double(__builtin_amdgcn_readfirstlane(mh), __builtin_amdgcn_readfirstlane(ml));
    return 1;
  }
  DDP_DEV void bs_leave(BwdShare* bs) {
    drain_stores();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&bs->word, ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // helper: the next kHelpChunk free knots from below -> [k0, k1), empty when none are left
  DDP_DEV void bs_claim_low(BwdShare* bs, int& k0, int& k1) {
    const unsigned long long o = a_add64(&bs->claim, (unsigned long long)kHelpChunk);
    const long long lo = (long long)(o & 0xffffffffull), hi = (long long)(o >> 32) - (long long)kClaimBias;
    k0 = (int)lo;
    k1 = (int)(lo + kHelpChunk < hi ? lo + kHelpChunk : hi);
  }
  // owner: take kOwnChunk more knots from above; the answer (the old claim word) is looked at a chunk later
  struct Pend {
    unsigned lo, hi;  // per-lane halves of lane 0's returned word
  };
  DDP_DEV void bs_claim_high_issue(BwdShare* bs, Pend& pd) {
    unsigned long long o = 0;
    if (threadIdx.x == 0) o = __hip_atomic_fetch_add(&bs->claim, 0ull - ((unsigned long long)kOwnChunk << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pd.lo = (unsigned)o;
    pd.hi = (unsigned)(o >> 32);
  }
  DDP_DEV int bs_claim_high_low(const Pend& pd) const {  // how many knots the helpers had taken from below at that moment
    return __builtin_amdgcn_readfirstlane((int)pd.lo);
  }
#else  // the emulator runs one wave: the same bookkeeping on plain memory (forced split only)
  DDP_DEV void rec_store(int k, const BRec& r, int lane) const {
    double* p = rec_ptr(k);
    for (int q = 0; q < 3; q++) {
      p[(q * 64 + lane) * 2] = r.r[2 * q];
      p[(q * 64 + lane) * 2 + 1] = r.r[2 * q + 1];
    }
  }
  DDP_DEV void rec_load(int k, BRec& r, int lane) const {
    const double* p = rec_ptr(k);
    for (int q = 0; q < 3; q++) {
      r.r[2 * q] = p[(q * 64 + lane) * 2];
      r.r[2 * q + 1] = p[(q * 64 + lane) * 2 + 1];
    }
  }
  DDP_DEV void flag_set(int k, int tag) const { *flag_ptr(k) = tag; }
  DDP_DEV int flag_peek(int k) const { return *flag_ptr(k); }
  DDP_DEV int flag_wait(int k, int tag) const { return *flag_ptr(k) == tag; }
  DDP_DEV int bs_open(BwdShare* bs, int cur, int infeas, double mu, int kfloor) {
    bs->seq++;
    bs->cur = cur;
    bs->infeas = infeas;
    bs->mu = mu;
    bs->claim = (kClaimBias + (unsigned long long)kfloor) << 32;
    bs->word += (unsigned)bs->seq << kBsCountBits;
    return bs->seq;
  }
  DDP_DEV void bs_close(BwdShare* bs) { bs->word &= kBsCountMask; }
  DDP_DEV int bs_tag(BwdShare* bs) const { return (int)(bs->word >> kBsCountBits); }
  DDP_DEV int bs_enter(BwdShare*, int&, int&, int&, double&) { return 0; }
  DDP_DEV void bs_leave(BwdShare*) {}
  DDP_DEV void bs_claim_low(BwdShare* bs, int& k0, int& k1) {
    const unsigned long long o = bs->claim;
    bs->claim += (unsigned long long)kHelpChunk;
    const long long lo = (long long)(o & 0xffffffffull), hi = (long long)(o >> 32) - (long long)kClaimBias;
    k0 = (int)lo;
    k1 = (int)(lo + kHelpChunk < hi ? lo + kHelpChunk : hi);
  }
  struct Pend {
    unsigned lo, hi;
  };
  DDP_DEV void bs_claim_high_issue(BwdShare* bs, Pend& pd) {
    const unsigned long long o = bs->claim;
    bs->claim -= (unsigned long long)kOwnChunk << 32;
    pd.lo = (unsigned)o;
    pd.hi = (unsigned)(o >> 32);
  }
  DDP_DEV int bs_claim_high_low(const Pend& pd) const { return (int)pd.lo; }
#endif

  DDP_DEV int bwd_sweep() { return bwd_sweep_t<false>(); }
  // one backwardpass() of the stepwise interface (direct_ddp_backward_pass): with the slack / dual gains ks, ky
  // (DIRECT_FIELD_KS / KY), which the hot kernels never form, and the GainBase bookkeeping of iterate_once / fwd_pass
  DDP_DEV int backward_pass_stepwise() {
    const int ok = bwd_sweep_t<true>();
    if (ok) note_completed_sweep(st.cur);
    else note_failed_sweep();
    return ok;
  }

  // ---- one knot of the backward sweep (see above for MODE) -----------------------------------------------------
  // INF: the mode of the sweep as a compile-time constant (0 feasible, 1 infeasible; -1: read from the context).  The fused
  // sweep is instantiated once per mode: the feasible one carries no dual rows (registers the kernel does not have).
  template <int MODE, bool kGains, int INF = -1>
  DDP_DEV int bwd_knot(BwdCtx& C, int k_) {
    constexpr bool kF = MODE != 2, kB = MODE != 1;  // the knot's front / back half is part of this instantiation
    {
      // the knot index is re-materialised every trip: as a visible induction variable it makes loop
      // strength reduction keep one 64-bit pointer PER ARRAY live (and spilled) across the whole body
      int k = k_;
      if constexpr (MODE == 2) k = DDP_UNIFORM_I(k_);  // a loop behind a loop with several exits: the counter may not be provably uniform
      DDP_LAUNDER_S(k);
      const int regi = C.regi, infeas = INF < 0 ? C.infeas : INF, buf = C.buf;
      const Acc lam = C.lam, sig = INF < 0 ? C.sig : (INF ? (Acc)1 : (Acc)-1);
      const Real mu = C.mu, wsn = C.wsn;
      const int P = C.Pn;
      const int nc = 6 * P + 55;
      (void)nc; (void)mu; (void)buf; (void)lam; (void)wsn; (void)sig;
      PLA(Real, rs, RPL);
      PLA(Real, ry, RPL);
      PLA(Real, rc, RPL);
      PLA(Real, rr, RPL);  // r (feasible) or rhat (infeasible)
      PLA(int, pkc, RPL);  // row descriptors of the knot being processed
      // Per-lane role descriptors (init_tables), each loaded ONE PHASE AHEAD of its use: a descriptor that is read in
      // the phase that needs it puts a second LDS round trip (descriptor, then the operands it addresses) on the
      // phase's critical path, and with three waves per SIMD those round trips are what the waves park on
      PLV(int, tw_t2);
      PLA(int, tw_r1, 3);
      PLV(int, tw_s);
      PLV(int, tw_h0);
      PLA(int, tw_ho, 3);
      PLA(int, tw_hv, 3);
      PLV(int, tw_r2);
      DDP_MARK("B_L");
      Real T = (Real)0;
      (void)T;
      if constexpr (kF) {
        // ---- L: this knot's data from the prefetch registers; issue the loads of the next knot
        LANES {
          LV(tw_t2) = L.lt[7][lane];
          commit(LV(C.pre), lane, P, false);
          for (int i = 0; i < RPL; i++) {
            LV(pkc)[i] = LV(C.pkn)[i];
            if (!slot_on(i, P)) continue;
            LV(rs)[i] = (Real)LV(C.pre).s[i];
            LV(ry)[i] = infeas ? (Real)LV(C.pre).y[i] : (Real)1;
          }
          if constexpr (MODE == 1) LV(C.e_mu) = 0;  // front: the knot's own maximum goes into its record
        }
        // the segment time straight from lane 18's prefetch registers (hi + lo with float storage, see ldx()):
        // the T-dependent tables then need no LDS round trip and share this phase
        T = (sizeof(St) < sizeof(double)) ? (Real)RDLANE_M(C.pre, zh, 18) + (Real)RDLANE_M(C.pre, zl, 18) : (Real)RDLANE_M(C.pre, zh, 18);
        if (k > C.klo) {
          C.Pn = DDP_UNIFORM_I(C.Pnn);
          C.Pnn = npU(k > 1 ? k - 2 : 0);
          const int same = (C.Pn == P) ? 1 : 0;
          LANES { prefetch(LV(C.pre), LV(C.pkn), same, lane, buf, k - 1, C.Pn, false, infeas); }
        }
      } else {
        // ---- L (back): Z = [F | G] and fT of the knot from its record; the row errors' maximum
        LANES {
#pragma unroll
          for (int pass = 0; pass < 3; pass++) LV(tw_r1)[pass] = L.lt[4 + pass][lane];
          const Acc v = LV(C.rc).r[3];
          if (lane >= 36 && lane < 54) {
            const int j = lane - 36;
            L.We[we_idx(15 + j / 6, j % 6)] = (Real)v;
          }
          if (lane >= 54 && lane < 63) L.fT[lane - 54] = v;
        }
        C.emu_u = fmax(C.emu_u, (Acc)RDLANE_M(C.rc, r[3], 63));
      }
      WSYNC();
      if constexpr (kF) {
        DDP_MARK("B_T2");
        // ---- T2: control values and their d/dT, fT, Ru / R'u / R''u (all lanes run all roles, clamped)
        const Real T2 = DDP_UNIFORM_R(T * T), T4 = DDP_UNIFORM_R(T2 * T2);
        Real pw[6];  // T^j as wave-uniform operands
#pragma unroll
        for (int j = 0; j < 6; j++) pw[j] = (j == 0) ? (Real)1 : DDP_UNIFORM_R(pow3(T, T2, T4, j));
        LANES {
          // One code path for 21 table rows x 3 axes (lanes 0..62, see ctrl_off()): rows 0..14 give the control
          // values and their d/dT; rows 15..17 give fT = (F' (x) I) x + (G' (x) I) u as the d/dT value
          // (DDP:1332); rows 18..20 give R u, R'u and R''u (DDP:1349-1355) as value, first and second derivative.
          // summed over the exponent j = i - o (see fwd_pass, phase T): uniform powers, no table reads.  lt[7]: byte
          // offsets of the row's first live weight WbE[cr][o] and of z[3 o + d], and o; terms past the end of the
          // row (j + o > 5) read a zero weight (WbE[108] = WdE[108] = 0) against a finite z (z[19..23] = 0)
          const int w2 = LV(tw_t2);
          if constexpr (kB) {
#pragma unroll
            for (int pass = 0; pass < 3; pass++) LV(tw_r1)[pass] = L.lt[4 + pass][lane];
          }
          const int wbb = w2 & 1023, o = (w2 >> 17) & 3;
          const Real* zb = byte_at(L.z, (w2 >> 10) & 127);
          Real v = 0, dv = 0, ddv = 0, z6[6], wb6[6], wd6[6];
#pragma unroll
          for (int j = 0; j < 6; j++) {
            const int wa = (j < 4 || j + o < 6) ? wbb + j * (int)sizeof(Real) : 108 * (int)sizeof(Real);
            wb6[j] = *byte_at(L.WbE, wa);
            wd6[j] = *byte_at(L.WdE, wa);
            z6[j] = zb[3 * j];
          }
          DDP_LOADS_ISSUED();
          // the lane's products weight x power of T ARE entries of We (rows 0..17, axis-0 lanes): stored on the way, the
          // others go to the dump slot behind the table; the powers of T themselves go to L.tp (read by phase H)
          Real* const wst = byte_at(L.We, ((unsigned)w2 >> 19) & 1023);
          const int nst = (int)((unsigned)w2 >> 29);
          L.tp[lane & 7] = pow3(T, T2, T4, lane & 7);
#pragma unroll
          for (int j = 0; j < 6; j++) {
            const Real pj = wb6[j] * pw[j];  // We[cr][o + j]
            *(j < nst ? wst + 18 * j : &L.We[108]) = pj;
            v += pj * z6[j];
            dv += wd6[j] * pw[j < 1 ? 0 : j - 1] * z6[j];
            if (j >= 2) ddv += (wd6[j] * (Real)(j - 1)) * pw[j - 2] * z6[j];  // only read for rows 18..20
          }
          if (lane < 45) {
            L.val[lane] = v;
            L.dval[lane] = dv;
          } else if (lane < 54) {
            L.fT[lane - 45] = dv;
          } else if (lane < 63) {
            L.Ru[lane - 54] = v;
            L.Rpu[lane - 54] = dv;
            L.Rppu[lane - 54] = ddv;
          } else {
            L.val[45] = T;
          }
        }
        WSYNC();
      }
      DDP_MARK("B_R1");
      // ---- R1: constraint rows -> D, g (front) ; VZ = Vxx * Z (back)
      LANES {
        if constexpr (kF) {
          LV(tw_s) = L.lt16[1][lane];
          RowK<Real> rk1[RPL];
          Row3 ov1[RPL];
#pragma unroll
          for (int i = 0; i < RPL; i++) {
            if (!slot_on(i, P)) continue;
            rk1[i] = row_unpack2(LV(pkc)[i]);
            ov1[i] = row_ops(L.val, LV(pkc)[i]);
          }
          DDP_LOADS_ISSUED();
#pragma unroll
          for (int i = 0; i < RPL; i++) {
            if (!slot_on(i, P)) continue;
            // every lane runs the row arithmetic (empty slots alias row 0); only the stores and the
            // running maxima are masked
            const RowK<Real>& rk = rk1[i];
            const int r = rk.r;
            const bool in = r >= 0;
            Real c = row_dot(rk, ov1[i]) + rk.o - (Real)B.k.shift, s = LV(rs)[i], y = LV(ry)[i];
            Real D, g, rv, frcp_reuse = (Real)0;
            if (infeas) {  // DDP:535-539, 554
              Real rm = s * y - mu;
              rv = s * (c + y) - rm;  // rhat
              Real yinv = frcp(y);
              D = s * yinv;
              g = s + yinv * rv;
              LV(C.e_mu) = fmax(LV(C.e_mu), in ? fmax(fabs(rm), fabs(c + y)) : (Real)0);
            } else {  // DDP:583-587, 601
              rv = s * c + mu;
              Real cinv = frcp(c);
              frcp_reuse = cinv;
              D = s * cinv;
              g = -mu * cinv;  // s - r/c
              LV(C.e_mu) = fmax(LV(C.e_mu), in ? fabs(rv) : (Real)0);
            }
            if constexpr (kGains) {
              // for phase R2: infeasible mode needs c and rhat; feasible mode only the two quotients r / c and s / c, so
              // that the slack gain ks = -(r + s cu ku) / c costs no second reciprocal there
              LV(rc)[i] = infeas ? c : D;
              LV(rr)[i] = infeas ? rv : rv * frcp_reuse;
            }
            if (in) {
              L.drow[r] = (Acc)D;
              L.grow[r] = (Acc)g;
            }
          }
        } else {
          // (back) the role descriptors of phase H, a phase ahead as in the fused knot's phase S2
          LV(tw_h0) = L.lt[0][lane];
#pragma unroll
          for (int t = 0; t < 3; t++) {
            LV(tw_ho)[t] = L.lt[1 + t][lane];
            LV(tw_hv)[t] = L.lt16[2 + t][lane];
          }
        }
        if constexpr (kB) {
#pragma unroll
          for (int pass = 0; pass < 3; pass++) {  // VZ[a][q], q < 18: 162 three-term entries (idle lanes redo the last)
            const int wz = LV(tw_r1)[pass];  // byte offsets: V[a][d] | We[row 15][i] << 10 | VZ[a][q] << 20 (loaded a phase ahead)
            const Acc* vp = byte_at(L.V, wz & 1023);
            const Real* hp = byte_at(L.We, (wz >> 10) & 1023);
            Acc v3[3];
            Real h3[3];
#pragma unroll
            for (int c = 0; c < 3; c++) v3[c] = vp[3 * c];
            h3[0] = hp[0];
            ld2(hp + 1, h3[1], h3[2]);  // rows 16, 17
            DDP_LOADS_ISSUED();
            Acc acc = 0;
#pragma unroll
            for (int c = 0; c < 3; c++) acc += v3[c] * h3[c];
            *byte_at(L.VZ, (wz >> 20) & 2047) = acc;
          }
          if (lane >= 34 && lane < 43) {  // the T column VZ[a][18] = V[a][:] . fT on lanes the third pass leaves idle
            const int a = lane - 34;
            Acc v9[9], f9[9];
#pragma unroll
            for (int c = 0; c < 9; c++) {
              v9[c] = L.V[a * 9 + c];
              f9[c] = L.fT[c];
            }
            DDP_LOADS_ISSUED();
            Acc acc = 0;
#pragma unroll
            for (int c = 0; c < 9; c++) acc += v9[c] * f9[c];
            L.VZ[a * 19 + 18] = acc;
          }
        }
      }
      WSYNC();
      if constexpr (kF) {
        DDP_MARK("B_S");
        // ---- S: 3x3 accumulators per control row.  Every lane runs every role on a clamped index: idle
        // lanes recompute (and re-store) a neighbour's value, which costs no extra instruction in SIMT and
        // removes the exec-mask bookkeeping of role branches.
        LANES {
          {  // lanes 0..35: S_j[d0][d1] = sum_q D_(jP+q) n_q[d0] n_q[d1]; lanes 36..53: h_j[d0] = sum_q g_(jP+q) n_q[d0] * 1
            const int ws = LV(tw_s);  // (loaded in phase R1) d0 | d1 << 2 | isS << 4 | j << 5 | destination (doubles from Sp[0]) << 8
            const bool isS = (ws >> 4) & 1;
            const Acc* w = (isS ? L.drow : L.grow) + ((ws >> 5) & 7) * P;
            const Real* n0 = &L.pl[ws & 3];
            const Real* nf = isS ? &L.pl[(ws >> 2) & 3] : &L.ones[0];
            const int fstep = isS ? 4 : 0;
            Acc acc = 0;
            int q = 0;
#pragma unroll 1
            for (; q + 6 <= P; q += 6) {  // six planes per trip: eighteen loads in flight before the first FMA (P >= 6 for every
                                          // corridor the planner produces: a polytope of its decomposition has at least six faces)
              Acc w6[6];
              Real a6[6], f6[6];
#pragma unroll
              for (int u = 0; u < 6; u++) {
                w6[u] = w[q + u];
                a6[u] = n0[4 * (q + u)];
                f6[u] = nf[fstep * (q + u)];
              }
              DDP_LOADS_ISSUED();
#pragma unroll
              for (int u = 0; u < 6; u++) acc += w6[u] * a6[u] * f6[u];
            }
#pragma unroll 2
            for (; q < P; q++) acc += w[q] * n0[4 * q] * nf[fstep * q];
            (&L.Sp[0])[(ws >> 8) & 255] = acc;
          }
          {  // velocity / acceleration rows: +/- pairs
            const int lq = lane < 27 ? lane : 26;
            const int d9 = lq / 9, cp9 = lq - 9 * d9;  // lq = 9 axis + control point: consecutive lanes store consecutive words
            const int l27 = 3 * cp9 + d9;                     // the row's number among the 27: 3 (control point) + axis
            const int rp = 6 * P + (l27 < 15 ? l27 : 15 + l27);  // 6P + l | 6P + 30 + (l - 15)
            const int rm = rp + (l27 < 15 ? 15 : 12);
            L.dl[dl_idx(cp9, d9)] = L.drow[rp] + L.drow[rm];
            L.hh[ch_idx(6 + cp9, d9)] = L.grow[rp] - L.grow[rm];
          }
          L.last[0] = L.drow[nc - 1];
          L.last[1] = L.grow[nc - 1];
        }
        WSYNC();
        DDP_MARK("B_S2");
        // ---- S2: Sd = S_cr * dval[cr]
        LANES {
          LV(tw_h0) = L.lt[0][lane];
#pragma unroll
          for (int t = 0; t < 3; t++) {
            LV(tw_ho)[t] = L.lt[1 + t][lane];
            LV(tw_hv)[t] = L.lt16[2 + t][lane];
          }
          const int lq = lane < 45 ? lane : 44;
          const int d = lq / 15, cr = lq - 15 * d;  // row-fastest roles: consecutive lanes store consecutive words of Sd
          const int l45 = 3 * cr + d;
          const int crp = cr < 6 ? cr : 5;
          const Acc* S = &L.Sp[sp_idx(crp, 0)];  // entries xx,xy,xz,yy,yz,zz of control point crp, six apart
          const Real* dv = &L.dval[crp * 3];
          // row d of the symmetric 3x3: (0,1,2) | (1,3,4) | (2,4,5)
          const int i0 = d, i1 = d == 0 ? 1 : (d == 1 ? 3 : 4), i2 = d == 2 ? 5 : (d == 1 ? 4 : 2);
          const Acc pos = S[6 * i0] * dv[0] + S[6 * i1] * dv[1] + S[6 * i2] * dv[2];
          const Acc oth = L.dl[dl_idx(cr < 6 ? 0 : cr - 6, d)] * L.dval[l45];
          L.Sd[ch_idx(cr, d)] = cr < 6 ? pos : oth;
        }
        WSYNC();
      }
      DDP_MARK("B_H");
      // ---- H: assemble the 19x19 system  Hzz = Z'VZ + quu -/+ A'DA,  Hz = qz + Z'Vx + A'g
      // The 18x18 block, p <= q.  A lane owns one (i, i2 >= i) pair of control points and one axis d and
      // walks the three axes d2 of the column: the twelve We operands, their products and the three H
      // operands are loaded once for three entries, and the velocity / acceleration rows (which only touch
      // d2 == d) need no second pass over the stored block.
      // Every entry is  (value half: Z'VZ, back)  +  (constraint half: quu -/+ A'DA ..., front)  with the constraint
      // half complete BEFORE the one addition that joins them: that value is what a record hands over.
      LANES {
        const int w0 = LV(tw_h0);  // (loaded a phase ahead) byte offsets of We[.][i] | We[.][i2] << 10 | dl[.][d] << 20; Rc index << 28
        const Real* Wi = byte_at(L.We, w0 & 1023);
        Acc ww[6], adv = 0;
        Real hh3[3];
        if constexpr (kF) {
          const Real* Wi2 = byte_at(L.We, (w0 >> 10) & 1023);
          const Acc* dld = byte_at(L.dl, (w0 >> 20) & 255);
          {
            Real w1[6], w2[6];
#pragma unroll
            for (int cr = 0; cr < 6; cr += 2) {
              ld2(Wi + cr, w1[cr], w1[cr + 1]);
              ld2(Wi2 + cr, w2[cr], w2[cr + 1]);
            }
            if constexpr (kB) {
              hh3[0] = Wi[15];
              ld2(Wi + 16, hh3[1], hh3[2]);
            }
            DDP_LOADS_ISSUED();
#pragma unroll
            for (int cr = 0; cr < 6; cr++) {
              ww[cr] = w1[cr] * w2[cr];
              DDP_PIN(ww[cr]);
            }
          }
#pragma unroll
          for (int part = 0; part < 3; part++) {  // the nine velocity / acceleration control points in batches of 4, 4, 1
            Real w1[4], w2[4];
            Acc dl4[4];
            if (part < 2) {
#pragma unroll
              for (int c2 = 0; c2 < 4; c2 += 2) {
                ld2(Wi + 6 + 4 * part + c2, w1[c2], w1[c2 + 1]);
                ld2(Wi2 + 6 + 4 * part + c2, w2[c2], w2[c2 + 1]);
                ld2(dld + 4 * part + c2, dl4[c2], dl4[c2 + 1]);
              }
            } else {
              w1[0] = Wi[14];
              w2[0] = Wi2[14];
              dl4[0] = dld[8];
            }
            DDP_LOADS_ISSUED();
#pragma unroll
            for (int c2 = 0; c2 < (part < 2 ? 4 : 1); c2++) adv += w1[c2] * w2[c2] * dl4[c2];
            DDP_PIN(adv);  // or the FMAs sink below the later batches' loads and all operands stay live
          }
          adv *= sig;
        } else {
          hh3[0] = Wi[15];
          ld2(Wi + 16, hh3[1], hh3[2]);
        }
        Acc* Hb = L.Hxx;  // Hxx | HR | Hzx | hdump are consecutive members: byte offsets from Hxx[0] (see init_tables)
#pragma unroll
        for (int t = 0; t < 3; t++) {  // column axis d2 = (d + t) mod 3: t == 0 is the lane's own axis
          const int wo = LV(tw_ho)[t], wv = LV(tw_hv)[t];
          Acc sp6[6], vz3[3];
          Real rc1 = 0, tp1 = 0;
          if constexpr (kF) {
            const Acc* sps = byte_at(L.Sp, ((unsigned)wo >> 24) & 0xf0);
#pragma unroll
            for (int cr = 0; cr < 6; cr += 2) ld2(sps + cr, sp6[cr], sp6[cr + 1]);
          }
          if constexpr (kB) {
            const Acc* vzq = byte_at(L.VZ, wv);
#pragma unroll
            for (int c = 0; c < 3; c++) vz3[c] = vzq[3 * c * 19];
          }
          if constexpr (kF) {
            // quu (DDP:1349-1355): w_snap Rc T^(i + i2 - 5) on the lane's own axis; Rc[9] = 0 where the pair has none
            if (t == 0) {
              rc1 = L.Rc[(unsigned)w0 >> 28];
              tp1 = L.tp[(wo >> 12) & 7];
            }
          }
          DDP_LOADS_ISSUED();
          Acc hc;  // the constraint half of the entry
          if constexpr (kF) {
            Acc ada = 0;
#pragma unroll
            for (int cr = 0; cr < 6; cr++) ada += ww[cr] * sp6[cr];
            const Acc quu = (t == 0) ? wsn * rc1 * tp1 : (Acc)0;
            hc = fma(sig, ada, quu);
            if (t == 0) hc = hc + adv;  // the velocity / acceleration rows only couple equal axes
            DDP_PIN(hc);
            if constexpr (MODE == 1) LV(C.rc).r[t] = hc;
          } else {
            hc = LV(C.rc).r[t];
          }
          if constexpr (kB) {
            Acc zvz = 0;
#pragma unroll
            for (int c = 0; c < 3; c++) zvz += hh3[c] * vz3[c];
            const Acc v = zvz + hc;
            *byte_at(Hb, wo & 0xfff) = v;
            *byte_at(Hb, (wo >> 16) & 0xfff) = v;
          }
        }
        if (lane < 36) {  // T column (lanes 0..17, against Sd) and Hz (lanes 18..35, against hh)
          const int p = lane < 18 ? lane : lane - 18;
          const int i = p / 3, d = p % 3;
          const Real* Wc = &L.We[we_idx(0, i)];
          Acc hT;  // the constraint half
          Real z3[3];  // Z[.][i]: rows 15..17
          if constexpr (kF) {
            const Acc* vec = (lane < 18 ? L.Sd : L.hh) + ch_idx(0, d);
            Acc acc = 0;
            Real w1[16];
            Acc v15[16];
#pragma unroll
            for (int cr = 0; cr < 14; cr += 2) {
              ld2(Wc + cr, w1[cr], w1[cr + 1]);
              ld2(vec + cr, v15[cr], v15[cr + 1]);
            }
            ld2(Wc + 14, w1[14], z3[0]);  // rows 14 and 15
            ld2(Wc + 16, z3[1], z3[2]);
            v15[14] = vec[14];
            const Acc rq = (i >= 3) ? (lane < 18 ? L.Rpu[p - 9] : L.Ru[p - 9]) : (Acc)0;
            DDP_LOADS_ISSUED();
#pragma unroll
            for (int cr = 0; cr < 15; cr++) acc += w1[cr] * v15[cr];
            hT = (lane < 18) ? fma(sig, acc, wsn * rq) : (wsn * rq + acc);
            DDP_PIN(hT);
            if constexpr (MODE == 1) LV(C.rc).r[3] = hT;
          } else {
            z3[0] = Wc[15];
            ld2(Wc + 16, z3[1], z3[2]);
            hT = LV(C.rc).r[3];
          }
          if constexpr (kB) {
            if (lane < 18) {
              Acc zvz = 0;
#pragma unroll
              for (int c = 0; c < 3; c++) zvz += z3[c] * L.VZ[(3 * c + d) * 19 + 18];
              const Acc v = zvz + hT;
              L.HR[9 * kHRS + p] = v;                      // row 18 (T), column p
              if (p >= 9) L.HR[(p - 9) * kHRS + 18] = v;   // and its mirror in the rows of u
            } else {
              Acc zv = 0;
#pragma unroll
              for (int c = 0; c < 3; c++) zv += z3[c] * L.Vx[3 * c + d];
              *(p < 9 ? &L.Hzx[p] : &L.HR[(p - 9) * kHRS + 19]) = zv + hT;
            }
          }
        }
      }
      {  // (T,T) entry and Hz[T]: 45 + 9 + 9 products each, one per lane, then a wave sum on the VALU
        PLV(Acc, ptt);
        PLV(Acc, pzt);
        Acc rtt = 0, rzt = 0;  // quu -/+ D_last, qz - g_last: the parts that are no sum over lanes
        LANES {
          const int a = lane < 45 ? 0 : (lane < 54 ? lane - 45 : 8);
          Acc tt, zt;
          if constexpr (kF) {
            const int tq = lane < 45 ? lane : 44;
            const int td = tq / 15, tr = tq - 15 * td;  // tq = 15 axis + row (row-fastest: Sd / hh are [axis][row])
            const Acc dv = L.dval[3 * tr + td], sd = L.Sd[ch_idx(tr, td)], hv = L.hh[ch_idx(tr, td)];
            const Acc zu = L.z[9 + a], r2 = L.Rppu[a], r1 = L.Rpu[a];
            DDP_LOADS_ISSUED();
            tt = lane < 45 ? sig * (dv * sd) : (Acc)0.5 * wsn * (zu * r2);
            zt = lane < 45 ? dv * hv : (Acc)0.5 * wsn * (zu * r1);
            DDP_PIN(tt);
            DDP_PIN(zt);
          } else {
            tt = LV(C.rc).r[4];
            zt = LV(C.rc).r[5];
          }
          if constexpr (kB) {
            const Acc ft = L.fT[a], vzt = L.VZ[a * 19 + 18], vxa = L.Vx[a];
            DDP_LOADS_ISSUED();
            LV(ptt) = lane < 45 ? tt : (lane < 54 ? fma(ft, vzt, tt) : (Acc)0);
            LV(pzt) = lane < 45 ? zt : (lane < 54 ? fma(ft, vxa, zt) : (Acc)0);
          } else {
            LV(ptt) = lane < 54 ? tt : (Acc)0;
            LV(pzt) = lane < 54 ? zt : (Acc)0;
          }
        }
        if constexpr (kF) {
          const Acc quu = (B.k.time_power == 2) ? (Acc)B.k.w_time : (Acc)0;
          const Acc qz = (B.k.time_power == 2) ? (Acc)B.k.w_time * T : (Acc)0.5 * (Acc)B.k.w_time;
          rtt = fma(sig, L.last[0], quu);
          rzt = qz - L.last[1];
        } else {
          rtt = (Acc)RDLANE_M(C.rc, r[4], 54);
          rzt = (Acc)RDLANE_M(C.rc, r[4], 55);
        }
        if constexpr (kB) {
          const Acc stt = (Acc)WAVE_SUM_D(ptt), szt = (Acc)WAVE_SUM_D(pzt);
          L.HR[9 * kHRS + 18] = stt + rtt;  // wave-uniform stores
          L.HR[9 * kHRS + 19] = szt + rzt;
        } else {
          // ---- (front) the rest of the record: Z and fT as phase T2 left them, the knot's row errors, the two sums' terms
          const Acc emu = (Acc)WAVE_MAX_D(C.e_mu);
          LANES {
            BRec& R = LV(C.rc);
            if (lane >= 36 && lane < 54) {
              const int j = lane - 36;
              R.r[3] = (Acc)L.We[we_idx(15 + j / 6, j % 6)];
            } else if (lane >= 54 && lane < 63) {
              R.r[3] = L.fT[lane - 54];
            } else if (lane == 63) {
              R.r[3] = emu;
            }
            R.r[4] = lane == 54 ? rtt : (lane == 55 ? rzt : LV(ptt));
            R.r[5] = LV(pzt);
          }
        }
      }
      WSYNC();
      if constexpr (!kB) return 1;
      if constexpr (kB) {
      DDP_MARK("B_C");
      // ---- C: LLT of Huu + lam I and the 10 right-hand sides [Hu | Hux], one column per lane in registers.
      // Columns live in ROWS OF 16 LANES so that every multiplier is a DP-ALU DPP row broadcast folded into the FMA
      // (ROW_FNMA: one v_fmac_f64_dpp instead of two v_readlane + FMA; no LDS round trip and no SGPR hop on the
      // serial chain): row 0 = matrix columns 0..9 + right-hand sides 10..15, row 1 = a second copy of the matrix
      // columns + right-hand sides 16..19; rows 2 and 3 repeat rows 0 and 1 (their results are never read).
      PLA(Acc, m, 10);
      PLA(Acc, ls, 10);  // scaled entries: ls[k] of matrix lane j is L[j][k]; of a right-hand-side lane (L^-1 [Hu | Hux])[k][.]
      PLA(Acc, xs, 10);  // MINUS the back-substituted columns (right-hand-side lanes): the gains themselves
      PLA(Acc, rd, 10);  // 1 / L_kk
      PLV(int, colv);
      DDP_PRED_DECL(bad);
      DDP_LANE_DECL(lane_c);
      LANES_AGAIN(lane_c) {
        const int l5 = lane & 31;
        const int col = l5 < 16 ? l5 : (l5 < 26 ? l5 - 16 : (l5 < 30 ? l5 - 10 : 19));
        LV(colv) = col;
        // column `col` of [Huu | Hu | Hux]: entry a sits in row a of HR (matrix columns behind the nine of Hux, Hu last)
        const Acc* src = &L.HR[col < 10 ? 9 + col : (col == 10 ? 19 : col - 11)];
#pragma unroll
        for (int a = 0; a < 10; a++) LV(m)[a] = src[a * kHRS];
        const Acc hu = L.HR[(lane < 10 ? lane : 9) * kHRS + 19];  // Hu, one entry per lane: max |Qu| for the optimality error
        DDP_LOADS_ISSUED();
        LV(C.e_qu) = fmax(LV(C.e_qu), fabs(hu));
        if (regi > 0) {  // lam = base^reg - 1 is exactly 0 at reg = 0 (the common case)
#pragma unroll
          for (int a = 0; a < 10; a++) LV(m)[a] += ((a == col) ? lam : (Acc)0);
        }
      }
      // Elimination.  After row kk of a column has been scaled by 1/L_kk it IS the multiplier of that
      // column's index (Huu + lam I and its Schur complements are symmetric: M[i][kk] = M[kk][i]): the
      // multiplier of row i is lane i's scaled entry.  The scaled rows are also written to LDS, where
      // phase R2 reads [y | Y] = L^-1 [Hu | Hux] (columns 10..19).
      static_for<0, 10>([&](auto KK) {
        constexpr int kk = KK;
        LANES_AGAIN(lane_c) {
          const Acc piv = ROW_BCAST(m, kk, kk);
          DDP_PRED_OR(bad, piv <= (Acc)0);  // Eigen LLT: NumericalIssue iff pivot <= 0 (NaN passes)
          const Acc rinv = frsq(piv);
          LV(rd)[kk] = rinv;
          LV(ls)[kk] = LV(m)[kk] * rinv;
          ROW_HAZARD(LV(ls)[kk]);
          L.UY[kk * 20 + LV(colv)] = LV(ls)[kk];
          static_for<kk + 1, 10>([&](auto I) {
            constexpr int i = I;
            ROW_FNMA(LV(m)[i], ls, kk, i, LV(ls)[kk]);
          });
        }
      });
      const int ok = DDP_PRED_LANE0(bad) ? 0 : 1;  // every row of lanes has seen the same ten pivots
      if (!ok) return 0;  // DDP:546-551, 595-600 (the sweep reports it)
      // Back substitution L^T X = [y | Y] in the right-hand-side lanes: L[j][i] (j > i) is entry i of matrix lane j,
      // again a row broadcast folded into the FMA.  The lanes carry Z = -X, the gains themselves ([ku | Ku] = -X,
      // DDP:561-564, 607-609):  x_i = (y_i - sum L_ji x_j) / L_ii  <=>  z_i = (y_i + sum L_ji z_j) * (-1 / L_ii).
      // Column-oriented: as soon as z_j is final it is folded into EVERY unfinished row i < j (nine, eight, ...
      // independent FMAs), so the serial chain is one FMA and one multiply per row instead of the whole row sum -
      // the order Eigen's own triangular solve uses.
      LANES_AGAIN(lane_c) {
#pragma unroll
        for (int i = 0; i < 10; i++) LV(xs)[i] = LV(ls)[i];
      }
      static_for_down<9, 0>([&](auto JJ) {
        constexpr int j = JJ;
        LANES_AGAIN(lane_c) {
          LV(xs)[j] = LV(xs)[j] * (-LV(rd)[j]);
          static_for<0, j>([&](auto I) {
            constexpr int i = I;
            ROW_FMA(LV(xs)[i], ls, i, j, LV(xs)[j]);
          });
        }
      });
      LANES_AGAIN(lane_c) {
        if (LV(colv) >= 10) {
          const int c = LV(colv) - 10;
#pragma unroll
          for (int a = 0; a < 10; a++) {
            const int idx = (c == 0) ? a : 10 + a * 9 + (c - 1);
            L.KU[idx] = LV(xs)[a];
          }
        }
      }
      WSYNC();
      DDP_MARK("B_G");
      // ---- G: cu*ku per control row (kGains only)
      LANES {
        LV(tw_r2) = L.lt16[0][lane];
        if constexpr (kGains) {
          if (lane < 45) {
            int cr = lane / 3, d = lane % 3;
            Acc acc = L.dval[lane] * L.KU[9];
#pragma unroll
            for (int i = 3; i < 6; i++) acc += L.We[we_idx(cr, i)] * L.KU[(i - 3) * 3 + d];
            L.G[lane] = (Real)acc;
          } else if (lane == 45) {
            L.G[45] = (Real)L.KU[9];  // the T_min row: A_r . ku = -ku_T
          }
        }
      }
      if constexpr (kGains) WSYNC();
      DDP_MARK("B_R2");
      // ---- R2: (kGains: slack / dual gains per row;) V recursion; gains ku, Ku to HBM
      LANES {
        if constexpr (kGains) {
          GSt* ksg = SpU(sp.KS, k);
          GSt* kyg = SpU(sp.KY, k);
          RowK<Real> rk2[RPL];
          Row3 og2[RPL];
#pragma unroll
          for (int i = 0; i < RPL; i++) {
            if (!slot_on(i, P)) continue;
            rk2[i] = row_unpack2(LV(pkc)[i]);
            og2[i] = row_ops(L.G, LV(pkc)[i]);
          }
          DDP_LOADS_ISSUED();
#pragma unroll
          for (int i = 0; i < RPL; i++) {
            if (!slot_on(i, P)) continue;
            const RowK<Real>& rk = rk2[i];
            const int r = rk.r;
            const Real cuku = row_dot(rk, og2[i]);
            const Real s = LV(rs)[i], c = LV(rc)[i], rv = LV(rr)[i];
            if (infeas) {  // DDP:568, 571
              const Real y = LV(ry)[i];
              const St ks = (St)((rv + s * cuku) * frcp(y));
              const St ky = (St)(-(c + y) - cuku);
              if (r >= 0) {
                ksg[r] = ks;
                kyg[r] = ky;
              }
            } else {  // DDP:611: -(r + s cu ku) / c with rv = r / c and c = s / c carried from phase R1
              const St ks = (St)(-(rv + c * cuku));
              if (r >= 0) ksg[r] = ks;
            }
          }
        }
        if constexpr (MODE == 0) {
          pin_prefetch(LV(C.pre), false, infeas);
          DDP_PIN(C.Pnn);
        }
        KUpU(k)[lane] = (St)L.KU[lane];
        {
          const int e2 = lane + 64 < 100 ? lane + 64 : 99;
          KUpU(k)[e2] = (St)L.KU[e2];
        }
        // Value-function recursion (DDP:626-628).  With [y | Y] = L^-1 [Hu | Hux], LL' = Huu + lam I and
        // [ku | Ku] = -(Huu + lam I)^-1 [Hu | Hux] the reference's
        //   Vxx = Hxx + Hxu Ku + (Hxu Ku)' + Ku' Huu Ku,   Vx = Hx + Ku'Hu + Ku'Huu ku + Hxu ku
        // are identically  Vxx = Hxx - Y'Y - lam Ku'Ku,   Vx = Hx - Y'y - lam Ku'ku   (10-term dots instead
        // of two 10x10x9 products).  V / Vx are dead since phases R1 / H: overwritten in place.
        {
          const int wv = LV(tw_r2);
          const int a = wv & 15, c2 = wv >> 4;
          const int aa = lane < 45 ? 0 : (lane < 54 ? lane - 45 : 8);
          // lanes 0..44: (colA, colB) = Y columns 1+a, 1+c2;  lanes 45..53: Y column 1+aa against y (column 0)
          const int cA = lane < 45 ? 1 + a : 1 + aa, cB = lane < 45 ? 1 + c2 : 0;
          const Acc h0 = lane < 45 ? L.Hxx[a * 9 + c2] : L.Hzx[aa];
          Acc yy = 0, kk2 = 0;
#pragma unroll
          for (int half = 0; half < 2; half++) {  // two batches of operands: 40 live doubles would spill
            Acc ya[5], yb[5];
#pragma unroll
            for (int j = 0; j < 5; j++) {
              const int k2 = 5 * half + j;
              ya[j] = L.UY[k2 * 20 + 10 + cA];
              yb[j] = L.UY[k2 * 20 + 10 + cB];
            }
            DDP_LOADS_ISSUED();
#pragma unroll
            for (int j = 0; j < 5; j++) yy += ya[j] * yb[j];
          }
          if (regi > 0) {  // lam = base^reg - 1 is exactly 0 at reg = 0 (the common case): the Ku'Ku term vanishes
#pragma unroll
            for (int half = 0; half < 2; half++) {
              Acc ka[5], kb[5];
#pragma unroll
              for (int j = 0; j < 5; j++) {
                const int k2 = 5 * half + j;
                ka[j] = (cA == 0) ? L.KU[k2] : L.KU[10 + k2 * 9 + (cA - 1)];
                kb[j] = (cB == 0) ? L.KU[k2] : L.KU[10 + k2 * 9 + (cB - 1)];
              }
              DDP_LOADS_ISSUED();
#pragma unroll
              for (int j = 0; j < 5; j++) kk2 += ka[j] * kb[j];
            }
          }
          const Acc vnew = h0 - yy - lam * kk2;
          if (lane < 45) {
            L.V[a * 9 + c2] = vnew;
            L.V[c2 * 9 + a] = vnew;
          } else if (lane < 54) {
            L.Vx[aa] = vnew;
          }
        }
      }
      WSYNC();
      }
    }
    return 1;
  }

  // ---- the helpers' half of a shared sweep: the front halves of the knots this wave claims, from below, each into its
  // record.  Also run by the owner itself in the forced split of the tests (Batch::bforce), before its own sweep.
  DDP_DEV void bwd_front_run(BwdShare* bs, int tag, int cur, int infeas, double mu_d) {
    BwdCtx C;
    C.regi = 0; C.buf = DDP_UNIFORM_I(cur); C.infeas = DDP_UNIFORM_I(infeas); C.klo = 0;
    C.lam = (Acc)0;
    C.sig = infeas ? (Acc)1 : (Acc)-1;
    C.mu = DDP_UNIFORM_R((Real)mu_d);
    C.wsn = (Real)B.k.w_snap;
    C.emu_u = (Acc)0;
    set_sweep_ptrs(C.buf);
    LANES {
      LV(C.e_mu) = 0;
      LV(C.e_qu) = 0;
      L.We[lane] = (Real)0;  // the entries of We with a zero weight (see bwd_sweep_t)
      if (lane + 64 < 112) L.We[lane + 64] = (Real)0;
    }
    WSYNC();
    int done = 0;
#pragma unroll 1
    while (true) {
      if (bs_tag(bs) != tag) break;  // the owner has closed the sweep (its LLT failed): nobody waits for more records
      int k0, k1;
      bs_claim_low(bs, k0, k1);
      if (k0 >= k1) break;
      C.klo = k0;
      C.Pn = DDP_UNIFORM_I(npU(k1 - 1));
      C.Pnn = npU(k1 - 2 > 0 ? k1 - 2 : 0);
      LANES { prefetch(LV(C.pre), LV(C.pkn), 0, lane, C.buf, k1 - 1, C.Pn, false, C.infeas); }
#pragma unroll 1
      for (int k_ = k1 - 1; k_ >= k0; k_--) {
        bwd_knot<1, false>(C, k_);
        LANES { rec_store(k_, LV(C.rc), lane); }
        flag_set(k_, tag);
      }
      done += k1 - k0;
    }
#if !defined(DIRECT_EMULATE)
    if (B.bvisits != nullptr && threadIdx.x == 0 && done) atomicAdd(B.bvisits, (unsigned long long)done);
#endif
  }
  // owner: publish the backward sweep that is about to run (iterate_once; bwd_sweep_t picks the tag up and closes it)
  int sh_tag = 0, sh_kfloor = 0;
  DDP_DEV void bwd_share_open() {
    BwdShare* bs = bshare_slot();
    sh_tag = 0;
    if (bs == nullptr) return;
    sh_kfloor = N > kOwnChunk ? N - kOwnChunk : 0;
    sh_tag = bs_open(bs, DDP_UNIFORM_I(st.cur), DDP_UNIFORM_I(st.infeas), st.mu, sh_kfloor);
  }

  // ---- the owner's half of a shared sweep below the knots it ran fused: the back halves of knots ks .. 0 from the
  // helpers' records.  The flag of a knot is asked for two knots ahead and its record one knot ahead (only once the flag
  // has been seen: the loads must not overtake it); a record that is late is waited for.
  struct BackIO {  // what bwd_sweep_t hands over and gets back
    int regi, buf, infeas, tag, ks, ok, kfail;
    Acc lam, emu_u;
    PLV(Acc, e_qu);
  };
  DDP_DEV void bwd_back_run(BackIO& io) {
    BwdCtx C;
    C.regi = DDP_UNIFORM_I(io.regi); C.buf = DDP_UNIFORM_I(io.buf); C.infeas = DDP_UNIFORM_I(io.infeas); C.klo = 0;
    C.lam = DDP_UNIFORM_R(io.lam);
    C.sig = C.infeas ? (Acc)1 : (Acc)-1;
    C.mu = (Real)0;
    C.wsn = (Real)B.k.w_snap;
    C.emu_u = DDP_UNIFORM_R(io.emu_u);
    C.Pn = 0; C.Pnn = 0;
    const int tag = DDP_UNIFORM_I(io.tag), ks = DDP_UNIFORM_I(io.ks);
    set_sweep_ptrs(C.buf);
    LANES {
      LV(C.e_mu) = 0;
      LV(C.e_qu) = LV(io.e_qu);
    }
    int ok = 1, kfail = 0;
    int have = flag_wait(ks, tag);
    if (have) {
      LANES { rec_load(ks, LV(C.rc), lane); }
    }
    PLV(int, fl);
    LANES { LV(fl) = ks > 0 ? flag_peek(ks - 1) : 0; }
    PLV(BRec, rn);
#pragma unroll 1
    for (int k_ = ks; k_ >= 0 && have; k_--) {
      int next = 0;
      if (k_ > 0) {
        if (RDLANE_I(fl, 0) == tag) {
          LANES { rec_load(k_ - 1, LV(rn), lane); }
          next = 1;
        }
        LANES { LV(fl) = k_ > 1 ? flag_peek(k_ - 2) : 0; }
      }
      ok = bwd_knot<2, false>(C, k_);
      if (!ok) {
        kfail = k_;
        break;
      }
      if (k_ > 0) {
        if (!next) {
          have = flag_wait(k_ - 1, tag);
          if (have) {
            LANES { rec_load(k_ - 1, LV(rn), lane); }
          }
        }
        LANES { LV(C.rc) = LV(rn); }
      }
    }
    if (!have) {  // a record never came (the protocol's error flag is up): the sweep counts as failed
      ok = 0;
      kfail = 0;
    }
    io.ok = ok;
    io.kfail = kfail;
    io.emu_u = C.emu_u;
    LANES { LV(io.e_qu) = LV(C.e_qu); }
  }

  // the owner's fused knots, from N - 1 down to the knot where the helpers' claims begin (ks; -1: the whole sweep)
  template <bool kGains, int INF>
  DDP_DEV int bwd_fused_run(BwdCtx& C, BwdShare* bs, Pend& pend, int& kfloor, int& ks, int& kfail) {
#pragma unroll 1
    for (int k_ = N - 1; k_ >= 0; k_--) {
      if (k_ < kfloor) {  // the owner's claim is used up: what did the next one get?
        const int low = bs_claim_high_low(pend);
        if (low >= kfloor) {  // the helpers have taken everything below
          ks = k_;
          break;
        }
        kfloor = kfloor - kOwnChunk > low ? kfloor - kOwnChunk : low;
        if (kfloor > 0) bs_claim_high_issue(bs, pend);
      }
      if (!bwd_knot<0, kGains, INF>(C, k_)) {
        kfail = k_;
        return 0;
      }
    }
    return 1;
  }

  template <bool kGains>
  DDP_DEV_NOINLINE int bwd_sweep_t() {
    DDP_LAUNDER_S(b);
    DDP_LAUNDER_S(N);
    {  // regulariser schedule (DDP:452-474)
      int reg = st.reg;
      if (st.fp_failed || st.bp_failed) reg += 1;
      else if (st.step == 0) reg -= 1;
      else if (st.step > 3) reg += 1;
      st.reg = reg < 0 ? 0 : (reg > 24 ? 24 : reg);
    }
    BwdCtx C;
    C.regi = DDP_UNIFORM_I(st.reg);
    double lam_d = 1.0;
    for (int q = 0; q < C.regi; q++) lam_d *= B.k.reg_base;
    C.lam = DDP_UNIFORM_R((Acc)(lam_d - 1.0));  // DDP:529
    C.buf = DDP_UNIFORM_I(st.cur);
    C.infeas = DDP_UNIFORM_I(st.infeas);
    C.mu = DDP_UNIFORM_R((Real)st.mu);
    C.wsn = (Real)B.k.w_snap;
    C.sig = C.infeas ? (Acc)1 : (Acc)-1;
    C.klo = 0;
    C.emu_u = (Acc)0;
    const int buf = C.buf, infeas = C.infeas;

    // Shared sweep: iterate_once has published it (bwd_share_open: the top kOwnChunk knots are the owner's from the start)
    BwdShare* bs = (kGains || sh_tag == 0) ? nullptr : bshare_slot();
    int kfloor = bs != nullptr ? sh_kfloor : 0;
    const int tag = sh_tag;
    sh_tag = 0;
    Pend pend;
    pend.lo = 0; pend.hi = 0;
    set_sweep_ptrs(buf);

    // terminal derivatives (DDP:1318-1323)
    LANES {
#pragma unroll 1
      for (int e = lane; e < 81; e += 64) L.V[e] = (e / 9 == e % 9) ? (Acc)B.k.w_term : (Acc)0;
      if (lane < 9) L.Vx[lane] = (Acc)B.k.w_term * (Acc)(ldx(XpU(0, N), lane) - (Real)B.xd[(size_t)b * 9 + lane]);
    }
    WSYNC();
    // opterr = max(|Qu|, |r|, |c + y|) over the sweep (DDP:641): one running maximum per lane is enough
    LANES {
      LV(C.e_mu) = 0;
      LV(C.e_qu) = 0;
    }

    // plane counts run two knots ahead of the sweep (the prefetch of knot k-1 needs P(k-1) for its
    // addresses: loading it on the spot would expose one HBM round trip per knot)
    C.Pn = DDP_UNIFORM_I(npU(N - 1));
    C.Pnn = npU(N > 1 ? N - 2 : 0);
    LANES {
      prefetch(LV(C.pre), LV(C.pkn), 0, lane, buf, N - 1, C.Pn, false, infeas);
      pin_prefetch(LV(C.pre), false, infeas);  // (on every path into the loop, or the wait comes back at its top)
      DDP_PIN(C.Pnn);
    }
    // We = WbE o T-powers (rows 0..14: control points, rows 15..17: Z = [F | G]; read by phases R1, H, G) is a by-product
    // of phase T2, which forms exactly these products on its way to the control values; the entries with a zero weight
    // (coefficient index below the row's exponent offset) never change: cleared once per sweep (the forward pass uses
    // the same LDS).
    LANES {
      L.We[lane] = (Real)0;
      if (lane + 64 < 112) L.We[lane + 64] = (Real)0;
    }
    WSYNC();
    if (bs != nullptr && kfloor > 0) bs_claim_high_issue(bs, pend);
    int ok = 1, kfail = 0;
    int ks = -1;  // knots ks .. 0 are the helpers': through records
    // (one instantiation for both modes: per-mode copies of the fused loop measured no faster, and the compiler contracts
    // the infeasible rows' arithmetic differently in a copy of its own - the bits would depend on who ran the knot)
    ok = bwd_fused_run<kGains, -1>(C, bs, pend, kfloor, ks, kfail);
#if defined(DDP_TIMELINE) && !defined(DIRECT_EMULATE)
    tl_split_ += ks + 1;
#endif
    if constexpr (kSplit && !kGains) {
      if (ok && ks >= 0) {  // the knots helpers have prepared: out of line (see front_cold / back_cold)
        BackIO io;
        io.regi = C.regi; io.buf = buf; io.infeas = infeas; io.tag = tag; io.ks = ks; io.ok = 1; io.kfail = 0;
        io.lam = C.lam; io.emu_u = C.emu_u;
        LANES { LV(io.e_qu) = LV(C.e_qu); }
        back_cold<Real, St, RPL>((const Batch<St>*)B.self, (LdsPtr<Lds>)&L, b, N, (void*)&io);
        ok = DDP_UNIFORM_I(io.ok);
        kfail = DDP_UNIFORM_I(io.kfail);
        C.emu_u = DDP_UNIFORM_R(io.emu_u);
        LANES { LV(C.e_qu) = LV(io.e_qu); }
      }
    }
    if (bs != nullptr) bs_close(bs);
    if (!ok) {  // DDP:546-551, 595-600
      st.bp_failed = 1;
      st.opterr = INFINITY;
      count_visits(0, N - kfail);
      kfail_last = kfail;  // (note_failed_sweep)
      return 0;
    }
    DDP_MARK("B_END");
    count_visits(0, N);
    const double mu_err = fmax((double)WAVE_MAX_D(C.e_mu), (double)C.emu_u);
    const double qu_err = WAVE_MAX_D(C.e_qu);
    st.bp_failed = 0;
    st.opterr = fmax((double)qu_err, mu_err);  // DDP:641
    return 1;
  }

  // ---- forward pass (DDP:647-778) ---------------------------------------------------------------
  // The filter test and update of one completed trial (DDP:737-757), one entry per lane: a serial scan would chain one
  // HBM round trip per entry.  Returns 1 and appends (logcost, err) when no entry dominates the trial.
  DDP_DEV int filter_accept(double* filt, int nfilter, double logcost, double err, int& nkeep) {
    int rejected = 0;
    for (int base = 0; base < nfilter && !rejected; base += 64) {
      PLV(int, rej);
      LANES {
        const int idx = base + lane;
        const bool valid = idx < nfilter;
        const double f0 = valid ? filt[2 * idx] : 0.0, f1 = valid ? filt[2 * idx + 1] : 0.0;
        LV(rej) = (valid && logcost >= f0 && err >= f1) ? 1 : 0;
      }
      rejected = WAVE_ANY(rej);
    }
    if (rejected) return 0;
    nkeep = 0;
    for (int base = 0; base < nfilter; base += 64) {
      PLV(int, keep);
      PLV(int, pos);
      PLV(double, e0);
      PLV(double, e1);
      LANES {
        const int idx = base + lane;
        const bool valid = idx < nfilter;
        LV(e0) = valid ? filt[2 * idx] : 0.0;
        LV(e1) = valid ? filt[2 * idx + 1] : 0.0;
        LV(keep) = (valid && (logcost > LV(e0) || err > LV(e1))) ? 1 : 0;
      }
      int total;
      WAVE_PREFIX_COUNT(keep, pos, total);
      LANES {
        if (LV(keep)) {  // compaction in place: nkeep + pos <= base + lane
          filt[2 * (nkeep + LV(pos))] = LV(e0);
          filt[2 * (nkeep + LV(pos)) + 1] = LV(e1);
        }
      }
      nkeep += total;
    }
    filt[2 * nkeep] = logcost;  // wave-uniform stores
    filt[2 * nkeep + 1] = err;
    return 1;
  }

  // what an accepted trial leaves behind
  struct Accept {
    int accepted, nkeep, step, neg, buf, viol;
    double stepsize, cost, costq, logcost, err, sumlog, errsum;
  };

  // ---- shared line search: the owner's and the helpers' halves of the HelpSlot protocol (device only) ----------
  // All of it is wave-uniform: lane 0 performs the atomic, the value is broadcast.  Agent scope throughout: owner and
  // helper may sit on different XCDs.
#if !defined(DIRECT_EMULATE)
  static __device__ __forceinline__ int a_load(int* p) {
    int v = 0;
    if (threadIdx.x == 0) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_amdgcn_readfirstlane(v);
  }
  static __device__ __forceinline__ void a_store(int* p, int v) {
    if (threadIdx.x == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  static __device__ __forceinline__ int a_add(int* p, int v) {
    int o = 0;
    if (threadIdx.x == 0) o = __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_amdgcn_readfirstlane(o);
  }
  DDP_DEV void proto_error() const { a_store(B.sched_err, 1); }
  // owner: open the line search around buffer `cur` to helpers.  Everything a helper reads (gains, the nominal
  // iterate, the row cache) was written by this wave before the release fence.
  DDP_DEV void share_open(HelpSlot* hs, int cur, double mu, int tag, int last_round, int first_round) {
    if (threadIdx.x == 0) {
      hs->cur = cur;
      __hip_atomic_store(&hs->last_round, last_round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // polled by next_work
      hs->mu = mu;
      __hip_atomic_store(&hs->next_round, first_round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&hs->cancel, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    a_store(&hs->gen, tag);
  }
  // owner: close it.  After this no helper reads the gains or writes a trial buffer of this trajectory.
  // (store gen / fence / load active here, add active / fence / load gen in help_enter: one of the two sees the other)
  DDP_DEV void share_close(HelpSlot* hs) {
    a_store(&hs->cancel, 1);
    a_store(&hs->gen, 0);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    int spins = 0;
    while (a_load(&hs->active) != 0) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > kSpinLimit) {
        proto_error();
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  DDP_DEV int share_claim(HelpSlot* hs) { return a_add(&hs->next_round, 1); }
  DDP_DEV int share_cancelled(HelpSlot* hs) { return a_load(&hs->cancel); }
  // helper: join the open line search of trajectory `b`, if there is one
  DDP_DEV int help_enter(HelpSlot* hs, int& cur, double& mu, int& tag, int& last_round) {
    const int g = a_load(&hs->gen);
    if (!g) return 0;
    a_add(&hs->active, 1);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    const int g2 = a_load(&hs->gen), c = a_load(&hs->cancel);
    if (g2 != g || c) {
      a_add(&hs->active, -1);
      return 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    int cv = 0, ml = 0, mh = 0, lv = 0;
    if (threadIdx.x == 0) {
      cv = hs->cur;
      lv = hs->last_round;
      ml = __double2loint(hs->mu);
      mh = __double2hiint(hs->mu);
    }
    cur = __builtin_amdgcn_readfirstlane(cv);
    last_round = __builtin_amdgcn_readfirstlane(lv);
    mu = __hiloint2double(__builtin_amdgcn_readfirstlane(mh), __builtin_amdgcn_readfirstlane(ml));
    tag = g;
    return 1;
  }
  DDP_DEV void help_leave(HelpSlot* hs) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the trial buffers this wave wrote
    a_add(&hs->active, -1);
  }
  // runner of round r (steps step0 .. step0 + nt - 1): hand the results to the owner
  DDP_DEV void post_results(HelpSlot* hs, int r, int step0, int nt, const TrialRes* res, int tag) {
    if (threadIdx.x == 0) {
#pragma unroll
      for (int t = 0; t < 2; t++) {  // field by field: the records stay in registers
        if (t >= nt) break;
        TrialRes* d = &hs->res[step0 + t];
        d->alive = res[t].alive; d->step = res[t].step; d->neg = res[t].neg; d->viol = res[t].viol;
        d->stepsize = res[t].stepsize; d->qsum = res[t].qsum; d->cost = res[t].cost;
        d->sumlog = res[t].sumlog; d->errsum = res[t].errsum;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    a_store(&hs->done[r], tag);
  }
  // owner: wait for round r (claimed by a helper, which always completes it unless the search is cancelled)
  DDP_DEV int fetch_results(HelpSlot* hs, int r, int step0, int nt, int tag, TrialRes* out) {
    int spins = 0;
    while (a_load(&hs->done[r]) != tag) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > kSpinLimit) {
        proto_error();
        return 0;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int* src = (const int*)&hs->res[step0];  // two records = 32 words, one per lane
    const int v = src[threadIdx.x & 31];
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int o = 16 * t;
      out[t].alive = __builtin_amdgcn_readlane(v, o);
      out[t].step = __builtin_amdgcn_readlane(v, o + 1);
      out[t].neg = __builtin_amdgcn_readlane(v, o + 2);
      out[t].viol = __builtin_amdgcn_readlane(v, o + 3);
      out[t].stepsize = __hiloint2double(__builtin_amdgcn_readlane(v, o + 5), __builtin_amdgcn_readlane(v, o + 4));
      out[t].qsum = __hiloint2double(__builtin_amdgcn_readlane(v, o + 7), __builtin_amdgcn_readlane(v, o + 6));
      out[t].cost = __hiloint2double(__builtin_amdgcn_readlane(v, o + 9), __builtin_amdgcn_readlane(v, o + 8));
      out[t].sumlog = __hiloint2double(__builtin_amdgcn_readlane(v, o + 11), __builtin_amdgcn_readlane(v, o + 10));
      out[t].errsum = __hiloint2double(__builtin_amdgcn_readlane(v, o + 13), __builtin_amdgcn_readlane(v, o + 12));
    }
    if (nt < 2) out[1].alive = 0;  // a one-step round: the second record belongs to the next round
    return 1;
  }
  DDP_DEV HelpSlot* help_slot() const { return B.help ? &B.help[b] : nullptr; }
#else  // the emulator runs one wave: nothing to share
  DDP_DEV void share_open(HelpSlot*, int, double, int, int, int) {}
  DDP_DEV void share_close(HelpSlot*) {}
  DDP_DEV int share_claim(HelpSlot*) { return 0; }
  DDP_DEV int share_cancelled(HelpSlot*) { return 0; }
  DDP_DEV int help_enter(HelpSlot*, int&, double&, int&, int&) { return 0; }
  DDP_DEV void help_leave(HelpSlot*) {}
  DDP_DEV void post_results(HelpSlot*, int, int, int, const TrialRes*, int) {}
  DDP_DEV int fetch_results(HelpSlot*, int, int, int, int, TrialRes*) { return 0; }
  DDP_DEV HelpSlot* help_slot() const { return nullptr; }
#endif

  // One round of the line search: NT trials (step indices step0 .. step0 + NT - 1) in one sweep over the knots.
  // NT = 1 is the plain forward roll; NT = 2 shares everything that belongs to the old iterate between two trials.
  // Leaves res[t] (wave-uniform) and the trial iterates in buffers trial_buf(cur, step0 + t).  `poll` != 0 (helpers):
  // the round is abandoned when the owner cancels the search.
  template <int NT>
  DDP_DEV void run_round(int step0, int cur, int infeas_rt, Real omt, Real mu, TrialRes* res, int poll, HelpSlot* hs) {
    // Trials are only ever paired in feasible mode (fwd_pass): the two-trial instantiation carries no dual rows at all -
    // registers the kernel does not have (98 -> 81 spilled VGPRs, no scratch access left inside the rounds; config 2
    // 34.4 -> 33.8 ms same-box).  (A feasible-only copy of the single-trial round as well measured no further gain.)
    const int infeas = (NT == 2) ? 0 : infeas_rt;
    set_sweep_ptrs(cur, trial_buf(cur, step0), trial_buf(cur, step0 + NT - 1));
    Real alpha[NT], oma[NT], amu[NT];  // step size, 1 - alpha (exact: alpha = 2^-step), alpha * mu
    TrialRes* tr = res;
#pragma unroll
    for (int t = 0; t < NT; t++) {
      tr[t].alive = 1;
      tr[t].step = step0 + t;
      tr[t].neg = 0;
      tr[t].viol = 0;
      tr[t].qsum = 0.0;
      tr[t].cost = 0.0;
      tr[t].sumlog = 0.0;
      tr[t].errsum = 0.0;
      tr[t].stepsize = 1.0;
      for (int q = 0; q < tr[t].step; q++) tr[t].stepsize *= 0.5;  // DDP:670
      alpha[t] = DDP_UNIFORM_R((Real)tr[t].stepsize);
      oma[t] = DDP_UNIFORM_R((Real)1 - alpha[t]);
      amu[t] = DDP_UNIFORM_R(alpha[t] * mu);
    }
#if !defined(DIRECT_EMULATE)
    int cancel_v = 0;  // the cancel flag as of one knot ago (the load is issued a knot ahead, like the prefetch)
#endif
    PLA(LogProd<Real>, plog, NT);
    PLA(Real, serr, NT);
    PLA(int, nviol, NT);
    LANES {
#pragma unroll
      for (int t = 0; t < NT; t++) {
        LV(plog)[t].init(); LV(serr)[t] = 0; LV(nviol)[t] = 0;
        if (lane < 9) L.ft[t].xn[lane] = ldx(XpU(0, 0), lane);
        if (lane < 5) {  // read against zero weights by phase T's clamp-free term loop
          L.ft[t].zn[19 + lane] = (Real)0;
          L.ft[t].dz[19 + lane] = (Real)0;
        }
      }
    }
    WSYNC();
    int Pn = DDP_UNIFORM_I(npU(0));
    int Pnn = npU(N > 1 ? 1 : 0);
    int n_visits = 0;  // trial-knots this round executes (observability)
    PLV(Pre, pre);
    PLA(int, pkn, RPL);
    PLA(int, pkc, RPL);
    LANES {
      prefetch(LV(pre), LV(pkn), 0, lane, cur, 0, Pn, true, infeas);
      pin_prefetch(LV(pre), true, infeas);  // (on every path into the loop, or the wait comes back at its top)
      DDP_PIN(Pnn);
    }
#pragma unroll 1
    for (int k_ = 0; k_ < N; k_++) {
      int k = DDP_UNIFORM_I(k_);  // see bwd_sweep(); the loop has several exits and the counter may not be provably uniform
      DDP_LAUNDER_S(k);
      const int P = Pn;
      DDP_MARK("F_L");
#if !defined(DIRECT_EMULATE)
      if (poll) {
        if (__builtin_amdgcn_readfirstlane(cancel_v)) {
#pragma unroll
          for (int t = 0; t < NT; t++) tr[t].alive = 0;
          break;
        }
        cancel_v = __hip_atomic_load(&hs->cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#endif
#pragma unroll
      for (int t = 0; t < NT; t++) n_visits += tr[t].alive;
      PLA(Real, rs, RPL);
      PLA(Real, ry, RPL);
      PLV(int, tw_ft);
      LANES {
        if (kWide) LV(tw_ft) = L.lt[7][lane];
        commit(LV(pre), lane, P, true);
        for (int i = 0; i < RPL; i++) {
          LV(pkc)[i] = LV(pkn)[i];
          if (!slot_on(i, P)) continue;
          LV(rs)[i] = (Real)LV(pre).s[i];
          LV(ry)[i] = infeas ? (Real)LV(pre).y[i] : (Real)1;
        }
      }
      // the old T straight from lane 18's prefetch register (no LDS round trip)
      const Real To = (sizeof(St) < sizeof(double)) ? (Real)RDLANE_M(pre, zh, 18) + (Real)RDLANE_M(pre, zl, 18) : (Real)RDLANE_M(pre, zh, 18);
      if (k + 1 < N) {
        Pn = DDP_UNIFORM_I(Pnn);
        Pnn = npU(k + 2 < N ? k + 2 : k + 1);
        const int same = (Pn == P) ? 1 : 0;
        LANES { prefetch(LV(pre), LV(pkn), same, lane, cur, k + 1, Pn, true, infeas); }
      }
      WSYNC();
    DDP_MARK("F_D");
      // ---- D: dx, du = alpha ku + Ku dx, u+ = u + du (DDP:689 / 695); powers of the new T
      const Real To2 = DDP_UNIFORM_R(To * To), To4 = DDP_UNIFORM_R(To2 * To2);
      Real pwo[6];  // T^j, j = 0..5, of the old iterate as wave-uniform operands (same values as the tables)
#pragma unroll
      for (int j = 0; j < 6; j++) pwo[j] = (j == 0) ? (Real)1 : DDP_UNIFORM_R(pow3(To, To2, To4, j));
      Real Tn[NT];
      Real pwn[NT][6];
#pragma unroll
      for (int t = 0; t < NT; t++) {
        Tn[t] = (Real)0;
        if (tr[t].alive) {
          typename Lds::FwdT& F = L.ft[t];
          PLV(Real, unew);
          PLV(Real, dxl);
          // x lanes: 0..8, and a second copy in lanes 16..24 of the ROW OF 16 that holds the ten u lanes (16..25): every
          // dx[c] then reaches the u lanes as a DPP row broadcast folded into the FMA, not through two v_readlane
          PLA(Real, kr, 9);
          PLV(Real, zlv);
          PLV(Real, kfv);
          LANES {
            const int lx = lane & 15;
            const int l9 = lx < 9 ? lx : 8;
            const Real xv = F.xn[l9];
            const Real zv = L.z[l9];
            // the u lanes' gains and old controls in the same batch of loads (lanes outside 16..25 read lane 25's)
            const int a = lane < 16 ? 0 : (lane < 26 ? lane - 16 : 9);
            if (kWide) {
#pragma unroll
              for (int c = 0; c < 9; c++) LV(kr)[c] = L.KUr[10 + a * 9 + c];
              LV(zlv) = L.z[9 + a];
              LV(kfv) = L.KUr[a];
            }
            DDP_LOADS_ISSUED();
            LV(dxl) = xv - zv;
            if (lane < 9) {
              F.dz[lane] = LV(dxl);
              F.zn[lane] = xv;
            }
            ROW_HAZARD(LV(dxl));
          }
          LANES {
            LV(unew) = (Real)0;
            if (lane >= 16 && lane < 26) {
              const int a = lane - 16;
              if (!kWide) {  // double storage: no registers to carry them across the block boundary
#pragma unroll
                for (int c = 0; c < 9; c++) LV(kr)[c] = L.KUr[10 + a * 9 + c];
                LV(zlv) = L.z[9 + a];
                LV(kfv) = L.KUr[a];
                DDP_LOADS_ISSUED();
              }
              Real acc = 0;
              static_for<0, 9>([&](auto C) {
                constexpr int c = C;
                ROW_FMA_V(acc, dxl, c, LV(kr)[c]);
              });
              // du = alpha ku + Ku dx: the whole step of the controls.  The rows of phase R multiply [dx; du], which
              // is how the slack / dual gains are eliminated (see there)
              const Real du = fma(alpha[t], LV(kfv), acc);
              F.dz[9 + a] = du;
              // every new quantity is rounded to the storage type BEFORE it is used, so that the recorded
              // cost / log-barrier belong exactly to the iterate that is stored (DESIGN.md "Precision")
              const Real un = pair_round(LV(zlv) + du);
              F.zn[9 + a] = un;
              LV(unew) = un;
            }
          }
          Tn[t] = RDLANE_V(unew, 25);
          const Real Tn2 = DDP_UNIFORM_PW(Tn[t] * Tn[t]), Tn4 = DDP_UNIFORM_PW(Tn2 * Tn2);
#pragma unroll
          for (int j = 0; j < 6; j++) pwn[t][j] = (j == 0) ? (Real)1 : DDP_UNIFORM_PW(pow3(Tn[t], Tn2, Tn4, j));
          if (Tn[t] < 0) tr[t].neg = 1;
        }
      }
      WSYNC();
    DDP_MARK("F_T");
      // ---- T: control values at the old and at the new iterate, A * [dx; du], x+, jerk cost
#pragma unroll
      for (int t = 0; t < NT; t++) {
        if (tr[t].alive) {
          typename Lds::FwdT& F = L.ft[t];
          LANES {
            {  // lanes 0..44: control values; 45..53: x+ (rows of [F|G]); 54..62: jerk-cost products (rows of R)
              const int l62 = lane < 63 ? lane : 62;
              // lt[7] (loaded in phase L): byte offsets of the row's first live weight WbE[cr][o] and of entry 3 o + d of
              // a knot record, and o - see phase T2 of the backward sweep; terms past the end of the row read a zero
              // weight against entries 19..23 of the records, which the round's prologue has zeroed
              const int w2 = kWide ? LV(tw_ft) : L.lt[7][lane];  // (double storage: no register to carry it from phase L)
              const int wbb = w2 & 1023, o = (w2 >> 17) & 3;
              const Real* zop = byte_at(L.z, (w2 >> 10) & 127);
              const Real* dzp = byte_at(F.dz, (w2 >> 10) & 127);
              const Real* znp = byte_at(F.zn, (w2 >> 10) & 127);
              Real dvo = 0, vn = 0, gf = 0, vo = 0;
              // Summed over the exponent j = i - o instead of the coefficient index i (same terms, same
              // order: the i < o terms have zero weight): the power T^j is then the same for every lane and
              // comes from a uniform register instead of three LDS table reads per term.
#pragma unroll
              for (int half = 0; half < 2; half++) {  // two batches of 15 operands
                Real wb6[3], wd6[3], zo6[3], dz6[3], zn6[3];
#pragma unroll
                for (int jj = 0; jj < 3; jj++) {
                  const int j = 3 * half + jj;
                  const int wa = (j < 4 || j + o < 6) ? wbb + j * (int)sizeof(Real) : 108 * (int)sizeof(Real);
                  wb6[jj] = *byte_at(L.WbE, wa);
                  wd6[jj] = *byte_at(L.WdE, wa);
                  zo6[jj] = zop[3 * j];
                  dz6[jj] = dzp[3 * j];
                  zn6[jj] = znp[3 * j];
                }
                DDP_LOADS_ISSUED();
#pragma unroll
                for (int jj = 0; jj < 3; jj++) {
                  const int j = 3 * half + jj;
                  const Real w = wb6[jj] * pwo[j];
                  vo += w * zo6[jj];  // the old control values (every alive trial writes the same ones)
                  gf += w * dz6[jj];
                  dvo += wd6[jj] * pwo[j < 1 ? 0 : j - 1] * zo6[jj];
                  vn += wb6[jj] * pwn[t][j] * zn6[jj];
                }
              }
              const Real un = F.zn[9 + (lane < 54 ? 0 : l62 - 54)], dT = F.dz[18];
              if (lane < 45) F.G[lane] = gf + dvo * dT;
              if (lane < 45) L.val[lane] = vo;
              Real* dst = lane < 45 ? &F.valn[l62] : (lane < 54 ? &F.xnx[l62 - 45] : &F.qp[l62 - 54]);
              *dst = lane < 54 ? vn : vn * un;  // u_a[d] * (R u)_a[d]: the nine of them sum to u'Ru (DDP:1294-1305)
            }
            if (lane == 63) {
              F.valn[45] = F.zn[18];
              F.G[45] = F.dz[18];
              L.val[45] = L.z[18];
            }
          }
        }
      }
      WSYNC();
#pragma unroll
      for (int t = 0; t < NT; t++)
        if (tr[t].alive) tr[t].qsum += knot_cost(Tn[t], L.ft[t].qp);
    DDP_MARK("F_R");
      // ---- R: rows: s+, y+, c+, fraction-to-boundary tests
      // The reference steps the slacks and duals with gains it has formed in the backward pass (DDP:565-575, 610-614,
      // 680-703):  s+ = s + alpha ks + Ks dx,  y+ = y + alpha ky + Ky dx.  With u+ - u = du = alpha ku + Ku dx and
      // A_r = [cx | cu]_r those are, identically,
      //   feasible    (ks = -(r + s cu ku) / c, Ks = -(s / c)(cx + cu Ku), r = s c + mu):
      //       s+ = (1 - alpha) s - alpha mu / c - (s / c) A_r [dx; du]
      //   infeasible  (ks = (rhat + s cu ku) / y, Ks = (s / y)(cx + cu Ku), ky = -(c + y) - cu ku, Ky = -(cx + cu Ku)):
      //       y+ = y - alpha (c + y) - A_r [dx; du],   s+ = s + (alpha rhat + s A_r [dx; du]) / y
      // so a trial needs s, (y), the OLD c (re-evaluated from the old control values, as DDP:696 does) and one row
      // product - no per-row gain ever crosses HBM, and the backward sweep has no row work after its phase R1.
      PLA(int, bad, NT);
      LANES {
        pin_prefetch(LV(pre), true, infeas);  // before this phase's stores of the trial iterate
        DDP_PIN(Pnn);
#if !defined(DIRECT_EMULATE)
        DDP_PIN(cancel_v);  // (a helper's poll of the cancel flag: loaded a knot ahead like the prefetch)
#endif
#pragma unroll
        for (int t = 0; t < NT; t++) LV(bad)[t] = 0;
        if (kWide) {
          RowK<Real> rks_[RPL];
          Row3 oo[RPL];
#pragma unroll
          for (int i = 0; i < RPL; i++) {
            rks_[i] = row_unpack2(LV(pkc)[i]);
            oo[i] = row_ops(L.val, LV(pkc)[i]);
          }
          DDP_LOADS_ISSUED();
          // what belongs to the old iterate is shared by the trials of the round
          Real co[RPL], w1[RPL], w2[RPL], bs[RPL], bc[RPL];
#pragma unroll
          for (int i = 0; i < RPL; i++) {
            const Real s = LV(rs)[i];
            co[i] = row_dot(rks_[i], oo[i]) + rks_[i].o - (Real)B.k.shift;
            bs[i] = omt * s;
            if (infeas) {
              const Real y = LV(ry)[i];
              w1[i] = frcp(y);
              w2[i] = s * (co[i] + y) - (s * y - mu);  // rhat (DDP:536-537)
              bc[i] = omt * y;
            } else {
              w1[i] = frcp(co[i]);
              w2[i] = mu * w1[i];
              bc[i] = omt * co[i];
            }
          }
#pragma unroll
          for (int t = 0; t < NT; t++) {
            if (tr[t].alive) {
              GSt* sn = SpU(sp.S[1 + t], k);
              Row3 og[RPL], ov[RPL];  // one trial's operands at a time: both trials' would not fit the register file
#pragma unroll
              for (int i = 0; i < RPL; i++) {
                og[i] = row_ops(L.ft[t].G, LV(pkc)[i]);
                ov[i] = row_ops(L.ft[t].valn, LV(pkc)[i]);
              }
              DDP_LOADS_ISSUED();
#pragma unroll
              for (int i = 0; i < RPL; i++) {
                // branch-free rows: empty slots alias row 0, their stores / reductions are masked
                const RowK<Real>& rk = rks_[i];
                const int r = rk.r;
                const bool in = r >= 0;
                const Real s = LV(rs)[i];
                const Real az = row_dot(rk, og[i]);
                const Real cn = row_dot(rk, ov[i]) + rk.o - (Real)B.k.shift;
                Real snew;
                if (infeas) {  // DDP:680-687
                  GSt* yn = SpU(sp.Y[1 + t], k);
                  const Real y = LV(ry)[i];
                  const Real ynew = (Real)(St)(fma(-alpha[t], co[i] + y, y) - az);
                  snew = (Real)(St)fma(w1[i], fma(alpha[t], w2[i], s * az), s);
                  // bitwise, not short-circuit: the latter compiles to nested exec-masked branches per row
                  LV(bad)[t] |= (int)(in & ((ynew < bc[i]) | (snew < bs[i])));
                  LV(plog)[t].mul(in ? ynew : (Real)1);
                  LV(serr)[t] += in ? fabs(cn + ynew) : (Real)0;
                  if (in) {
                    yn[r] = (St)ynew;
                    sn[r] = (St)snew;
                  }
                } else {  // DDP:694-703
                  snew = (Real)(St)fma(s, fma(-w1[i], az, oma[t]), -(amu[t] * w1[i]));
                  LV(bad)[t] |= (int)(in & ((cn > bc[i]) | (snew < bs[i])));
                  LV(plog)[t].mul(in ? -cn : (Real)1);
                  if (in) sn[r] = (St)snew;
                }
                LV(nviol)[t] += (int)(in & (cn >= (Real)2.0e-4));
              }
            }
          }
        } else {  // double storage / many row slots: row by row, both trials of a row together (the registers hold no more)
          for (int i = 0; i < RPL; i++) {
            if (!slot_on(i, P)) continue;
            // branch-free rows: empty slots alias row 0, their stores / reductions are masked
            const RowK<Real> rk = row_unpack(LV(pkc)[i]);
            const int r = rk.r;
            const bool in = r >= 0;
            const Real s = LV(rs)[i], y = LV(ry)[i];
            const Real co = row_c(L.val, rk);
            const Real bs = omt * s;
            Real w1, w2, bc;
            if (infeas) {
              w1 = frcp(y);
              w2 = s * (co + y) - (s * y - mu);  // rhat (DDP:536-537)
              bc = omt * y;
            } else {
              w1 = frcp(co);
              w2 = mu * w1;
              bc = omt * co;
            }
#pragma unroll
            for (int t = 0; t < NT; t++) {
              if (tr[t].alive) {
                typename Lds::FwdT& F = L.ft[t];
                GSt* sn = SpU(sp.S[1 + t], k);
                const Real az = row_lin(F.G, rk);
                const Real cn = row_c(F.valn, rk);
                Real snew;
                if (infeas) {  // DDP:680-687
                  GSt* yn = SpU(sp.Y[1 + t], k);
                  const Real ynew = (Real)(St)(fma(-alpha[t], co + y, y) - az);
                  snew = (Real)(St)fma(w1, fma(alpha[t], w2, s * az), s);
                  // bitwise, not short-circuit: the latter compiles to nested exec-masked branches per row
                  LV(bad)[t] |= (int)(in & ((ynew < bc) | (snew < bs)));
                  LV(plog)[t].mul(in ? ynew : (Real)1);
                  LV(serr)[t] += in ? fabs(cn + ynew) : (Real)0;
                  if (in) {
                    yn[r] = (St)ynew;
                    sn[r] = (St)snew;
                  }
                } else {  // DDP:694-703
                  snew = (Real)(St)fma(s, fma(-w1, az, oma[t]), -(amu[t] * w1));
                  LV(bad)[t] |= (int)(in & ((cn > bc) | (snew < bs)));
                  LV(plog)[t].mul(in ? -cn : (Real)1);
                  if (in) sn[r] = (St)snew;
                }
                LV(nviol)[t] += (int)(in & (cn >= (Real)2.0e-4));
              }
            }
          }
        }
#pragma unroll
        for (int t = 0; t < NT; t++) {
          if (tr[t].alive) {
            LV(plog)[t].norm();
            if (lane < 19) stx(XpU(1 + t, k), lane, L.ft[t].zn[lane]);
            if (lane < 9) L.ft[t].xn[lane] = L.ft[t].xnx[lane];
          }
        }
      }
      int any_alive = 0;
#pragma unroll
      for (int t = 0; t < NT; t++) {  // a trial that violates the fraction-to-boundary rule at this knot is over (DDP:689-703)
        PLV(int, bt);
        LANES { LV(bt) = LV(bad)[t]; }
        if (tr[t].alive && WAVE_ANY(bt)) tr[t].alive = 0;
        any_alive |= tr[t].alive;
      }
      WSYNC();
      if (!any_alive) break;
    }
    DDP_MARK("F_END");
    count_visits(1, n_visits);
    // totals of the trials that survived every knot (DDP:716-732)
#pragma unroll
    for (int t = 0; t < NT; t++) {
      if (tr[t].alive) {
        PLV(Real, slog);
        PLV(Real, se1);
        PLV(int, nv1);
        LANES {
          if (lane < 9) {
            stx(XpU(1 + t, N), lane, L.ft[t].xn[lane]);
            L.z[lane] = L.ft[t].xn[lane] - B.xd[(size_t)b * 9 + lane];
          }
        }
        WSYNC();
        const double pterm = terminal_sq();
        WSYNC();
        LANES {
          LV(slog) = LV(plog)[t].value();
          LV(se1) = LV(serr)[t];
          LV(nv1) = LV(nviol)[t];
        }
        tr[t].cost = tr[t].qsum + 0.5 * B.k.w_term * pterm;  // DDP:716-717
        tr[t].sumlog = WAVE_SUM_D(slog);
        tr[t].errsum = WAVE_SUM_D(se1);
        tr[t].viol = WAVE_SUM_I(nv1);
      }
    }
  }

  // the filter's verdict on one completed trial (DDP:718-757); acceptance goes strictly in step order
  DDP_DEV void consider(const TrialRes& r, int cur, int infeas, double mu_d, int nfilter, double* filt, Accept& A) {
    if (A.accepted || !r.alive) return;
    const double logcost_t = r.cost - mu_d * r.sumlog;  // DDP:718-732
    const double err_t = infeas ? fmax(B.k.tol, r.errsum) : 0.0;
    if (filter_accept(filt, nfilter, logcost_t, err_t, A.nkeep)) {
      A.accepted = 1;
      A.step = r.step; A.neg = r.neg; A.stepsize = r.stepsize; A.buf = trial_buf(cur, r.step);
      A.cost = r.cost; A.costq = r.qsum; A.logcost = logcost_t; A.err = err_t; A.sumlog = r.sumlog; A.errsum = r.errsum;
      A.viol = r.viol;
    }
  }

  // Step sizes 2^0 .. 2^-10 are tried in order (DDP:666-670) and the first that passes is taken.  The trials of one
  // iteration are independent of each other (same gains, same nominal iterate, the filter only changes on
  // acceptance), which is used twice, both times without touching the arithmetic of a trial or the order in which
  // the trials are judged - results are bitwise those of the sequential search:
  //  * from the second attempt on TWO of them share a sweep (rounds {0}, {1,2}, {3,4}, ... {9,10}): everything that
  //    belongs to the old iterate (prefetch, gains, row descriptors, c and s/c, the old powers of T) is loaded once, and
  //    the two dependent chains interleave in a wave that is otherwise latency-bound;
  //  * with B.help, once step 0 has failed the rounds are handed out through HelpSlot::next_round: waves that wait
  //    for this trajectory's chunk to finish (k_iterate_dyn) run later rounds at the same time as the owner runs round
  //    1, each into its own trial buffers, and report TrialRes records.  `helper` != 0 is that other side: no filter,
  //    no TrajState, parameters from the slot.
  DDP_DEV_NOINLINE void fwd_pass(int helper = 0) {
    DDP_LAUNDER_S(b);
    DDP_LAUNDER_S(N);
    HelpSlot* hs = help_slot();
    int cur = 0, infeas = 0, tag = 0, last_round = 10;
    double mu_d = 0.0;
    if (helper) {
      if (!help_enter(hs, cur, mu_d, tag, last_round)) return;
    } else {
      cur = DDP_UNIFORM_I(st.cur);
      infeas = DDP_UNIFORM_I(st.infeas);
      mu_d = st.mu;
      tag = DDP_UNIFORM_I(st.fwd_passes) + 1;
    }
    const double tau_d = fmax(0.99, 1.0 - mu_d);
    const Real omt = DDP_UNIFORM_R((Real)(1.0 - tau_d));
    const Real mu_r = DDP_UNIFORM_R((Real)mu_d);
    const int nfilter = helper ? 0 : DDP_UNIFORM_I(st.nfilter);
    double* filt = B.filt + (size_t)b * B.fcap * 2;
    // rounds: {0}, then the pairs (2r-1, 2r), r = 1..5, or the single steps r = 1..10 (few trajectories on many waves:
    // every step then finds a wave of its own)
    int tail = 0;  // few trajectories left on many waves (see Batch::live)
#if !defined(DIRECT_EMULATE)
    if (!helper && hs != nullptr && B.live != nullptr) tail = a_load(B.live) <= B.tail_thresh ? 1 : 0;
#endif
    const int pair = helper ? (last_round == 5 ? 1 : 0) : ((B.k.pair_trials && !infeas && !tail) ? 1 : 0);
    const int share = (!helper && !infeas && hs != nullptr) ? 1 : 0;
    if (!helper) last_round = pair ? 5 : 10;
    Accept A;
    A.accepted = 0; A.nkeep = 0; A.step = 0; A.neg = 0; A.buf = cur; A.viol = 0;
    A.stepsize = 0.0; A.cost = 0.0; A.costq = 0.0; A.logcost = 0.0; A.err = 0.0; A.sumlog = 0.0; A.errsum = 0.0;
    int r_eval = 0;  // owner: the first round whose results have not been judged yet
    int opened = 0;
    if (share && !pair && (B.k.pair_trials == 0 || tail) && B.help_early) {
      // few trajectories on many waves: the search is open from step 0 on - the waves that would evaluate the later
      // steps have nothing else to do, and when step 0 is rejected the answer of steps 1 .. 10 is already there
      share_open(hs, cur, mu_d, tag, last_round, 0);
      opened = 1;
    }
#pragma unroll 1
    while (true) {
      int mine = (helper || opened) ? share_claim(hs) : r_eval;
      if (mine > last_round) mine = -1;
      DDP_DBG_MARK(B, (helper ? 200 : 100) + (mine < 0 ? 99 : mine));
      if (helper && (mine < 0 || share_cancelled(hs))) break;
      TrialRes res[2];
      res[0].alive = 0;
      res[1].alive = 0;
      const int nt = (pair && mine > 0) ? 2 : 1, step0 = (pair && mine > 0) ? 2 * mine - 1 : mine;
      if (mine >= 0) {
        if (nt == 2) run_round<2>(step0, cur, infeas, omt, mu_r, res, helper, hs);
        else run_round<1>(step0, cur, infeas, omt, mu_r, res, helper, hs);
      }
      if (helper) {
        post_results(hs, mine, step0, nt, res, tag);
        continue;
      }
      const int upto = mine >= 0 ? mine : last_round;
      int broken = 0;
#pragma unroll 1
      for (; r_eval <= upto && !A.accepted; r_eval++) {
        if (r_eval == mine) {
          consider(res[0], cur, infeas, mu_d, nfilter, filt, A);
          consider(res[1], cur, infeas, mu_d, nfilter, filt, A);
        } else {  // a helper's round
          TrialRes o[2];
          if (!fetch_results(hs, r_eval, pair ? 2 * r_eval - 1 : r_eval, pair ? 2 : 1, tag, o)) {
            broken = 1;
            break;
          }
          consider(o[0], cur, infeas, mu_d, nfilter, filt, A);
          consider(o[1], cur, infeas, mu_d, nfilter, filt, A);
        }
      }
      if (A.accepted || broken || mine < 0 || r_eval > last_round) break;
      if (share && !opened) {
        share_open(hs, cur, mu_d, tag, last_round, r_eval);
        opened = 1;
      }
    }
    if (helper) {
      help_leave(hs);
      return;
    }
    if (opened) share_close(hs);
    note_completed_sweep(cur);
    commit_search(A);
  }

  // GainBase bookkeeping (see the struct): a sweep that failed leaves knots kfail .. 0 with the gains they had (DDP:548-550
  // returns before 568-572 / 630-631) ...
  int kfail_last = 0;
  DDP_DEV void note_failed_sweep() {
    LANES {
      if (lane == 0) {
        GainBase* g = &B.gbase[b];
        if (kfail_last + 1 < g->kreach) g->kreach = kfail_last + 1;
      }
    }
  }
  // ... and a forward pass that runs at all follows a completed sweep (iterate_once parks the trajectory otherwise): every
  // knot's gains belong to the iterate the trials started from and to the barrier parameter of now - recorded at the end of
  // fwd_pass, once per iteration and outside the sweeps
  DDP_DEV void note_completed_sweep(int buf) {
    LANES {
      if (lane == 0) {
        GainBase* g = &B.gbase[b];
        g->mu = st.mu;
        g->buf = buf;
        g->kreach = N;
      }
    }
  }

  // the end of forwardpass() (DDP:760-776)
  DDP_DEV void commit_search(const Accept& A) {
    if (!A.accepted) {  // DDP:760-762
      st.fp_failed = 1;
      st.stepsize = 0.0;
    } else {  // DDP:763-776
      count_visits(3, 1);
      st.nfilter = A.nkeep + 1;
      st.cost = A.cost;
      st.costq = A.costq;
      st.logcost = A.logcost;
      st.err = A.err;
      st.sumlog = A.sumlog;
      st.errsum = A.errsum;
      st.viol = A.viol;
      st.neg_time = A.neg;
      st.stepsize = A.stepsize;
      st.step = A.step;
      st.fp_failed = 0;
      st.cur = A.buf;
    }
  }

  // ---- forwardpass() with the gains AS THE REFERENCE'S MEMBERS HOLD THEM (DDP:647-778 after DDP:297-310 gave up) --------
  // run_round regenerates the slack / dual gains of a knot from the iterate the trial starts from - the iterate every
  // gain was formed from as long as the last backward sweep completed.  After an aborted retry sequence that is only
  // true of the knots some sweep of the sequence reached (k >= kreach); knots [0, kreach) still carry bp.ku .. bp.Ky of
  // the last COMPLETED sweep (DDP:568-572, 611-614, 630-631 are never reached for them), formed from the iterate in
  // buffer gb.buf with barrier parameter gb.mu - or the zeros of DDP:154-159 when no sweep has ever completed.  With
  // (s_g, y_g, c_g, A_g = [cx | cu], mu_g) of THAT iterate and du = alpha ku + Ku dx the stored gains give
  //   feasible    s+ = s - alpha s_g - alpha mu_g / c_g - (s_g / c_g) A_g [dx; du]
  //   infeasible  y+ = y - alpha (c_g + y_g) - A_g [dx; du],   s+ = s + (alpha rhat_g + s_g A_g [dx; du]) / y_g
  // while the fraction-to-boundary tests (DDP:684-688, 698-702) compare with the CURRENT s, y, c.  The buffer the gains
  // were formed from is intact: the trials of a line search never write the buffer they start from, and nothing has
  // written a buffer since.  One trial at a time, no prefetch, no helpers: this runs at most once per solve (k_stuck,
  // and the stepwise direct_ddp_forward_pass after a failed direct_ddp_backward_pass).
  DDP_DEV void stale_fwd_pass() {
    const int cur = DDP_UNIFORM_I(st.cur), infeas = DDP_UNIFORM_I(st.infeas);
    const GainBase gb = B.gbase[b];
    const int gbuf = DDP_UNIFORM_I(gb.buf), kst = DDP_UNIFORM_I(gb.kreach);
    const double mu_d = st.mu;
    const double tau_d = fmax(0.99, 1.0 - mu_d);
    const Real omt = (Real)(1.0 - tau_d), mu_r = (Real)mu_d, gmu_r = (Real)gb.mu;
    int wb = 0;  // the trial buffer: neither the iterate nor the gains' iterate (nbuf >= 3)
    while (wb == cur || wb == gbuf) wb++;
    const int nfilter = DDP_UNIFORM_I(st.nfilter);
    double* filt = B.filt + (size_t)b * B.fcap * 2;
    Accept A;
    A.accepted = 0; A.nkeep = 0; A.step = 0; A.neg = 0; A.buf = cur; A.viol = 0;
    A.stepsize = 0.0; A.cost = 0.0; A.costq = 0.0; A.logcost = 0.0; A.err = 0.0; A.sumlog = 0.0; A.errsum = 0.0;
    typename Lds::FwdT& F = L.ft[0];
    typename Lds::FwdT& Gb = L.ft[1];  // zn, tpn: the knot record of the gains' iterate and the powers of its T
#pragma unroll 1
    for (int step = 0; step < 11 && !A.accepted; step++) {  // DDP:666-670
      double stepsize = 1.0;
      for (int q = 0; q < step; q++) stepsize *= 0.5;
      const Real alpha = (Real)stepsize;
      PLV(LogProd<Real>, plog);
      PLV(Real, serr);
      PLV(int, nviol);
      LANES {
        LV(plog).init(); LV(serr) = 0; LV(nviol) = 0;
        if (lane < 9) F.xn[lane] = ldx(Xp(cur, 0), lane);
      }
      WSYNC();
      double qsum = 0.0;
      int neg = 0, alive = 1, n_visits = 0;
#pragma unroll 1
      for (int k = 0; k < N && alive; k++) {
        const int P = np_(k);
        const bool nogain = k < kst && gbuf < 0;
        const int pb = (k < kst && !nogain) ? gbuf : cur;  // the iterate this knot's gains were formed from
        const Real gmu = k < kst ? gmu_r : mu_r;
        n_visits++;
        PLA(Real, sq, RPL);
        PLA(Real, yq, RPL);
        PLA(Real, sg, RPL);
        PLA(Real, yg, RPL);
        LANES {
          if (lane < 19) {
            L.z[lane] = ldx(Xp(cur, k), lane);
            Gb.zn[lane] = ldx(Xp(pb, k), lane);
          }
          for (int e = lane; e < 4 * P; e += 64) L.pl[e] = (Real)planes_(k)[e];
          for (int e = lane; e < 100; e += 64) L.KUr[e] = nogain ? (Real)0 : (Real)KUp(k)[e];
          for (int i = 0; i < RPL; i++) {
            if (!slot_on(i, P)) continue;
            const int r = (row_pack(i, lane, P) & kRMask) - 1;
            const int rc = r >= 0 ? r : 0;
            LV(sq)[i] = (Real)Sp_(B.S[cur], k)[rc];
            LV(sg)[i] = (Real)Sp_(B.S[pb], k)[rc];
            LV(yq)[i] = infeas ? (Real)Sp_(B.Y[cur], k)[rc] : (Real)1;
            LV(yg)[i] = infeas ? (Real)Sp_(B.Y[pb], k)[rc] : (Real)1;
          }
        }
        WSYNC();
        LANES {  // dx (DDP:681 / 695)
          if (lane < 9) {
            F.dz[lane] = F.xn[lane] - L.z[lane];
            F.zn[lane] = F.xn[lane];
          }
        }
        WSYNC();
        LANES {  // du = alpha ku + Ku dx, u+ = u + du (DDP:689 / 696)
          if (lane < 10) {
            Real acc = 0;
            for (int c = 0; c < 9; c++) acc = fma(L.KUr[10 + lane * 9 + c], F.dz[c], acc);
            const Real du = fma(alpha, L.KUr[lane], acc);
            F.dz[9 + lane] = du;
            F.zn[9 + lane] = pair_round(L.z[9 + lane] + du);
          }
        }
        WSYNC();
        const Real Tq = L.z[18], Tg = Gb.zn[18], Tn = F.zn[18];
        if (Tn < 0) neg = 1;
        LANES {
          if (lane < 8) {
            L.tp[lane] = powi(Tq, lane);
            Gb.tpn[lane] = powi(Tg, lane);
            F.tpn[lane] = powi(Tn, lane);
          }
        }
        WSYNC();
        LANES {
          if (lane < 45) {
            const int cr = lane / 3, d = lane % 3;
            L.val[lane] = ctrl_val(L.z, L.tp, cr, d);       // c of the current iterate (DDP:663 cold)
            L.G[lane] = ctrl_val(Gb.zn, Gb.tpn, cr, d);     // c of the gains' iterate
            F.valn[lane] = ctrl_val(F.zn, F.tpn, cr, d);    // c+ (DDP:696)
            // A_g [dx; du]: the x and u-bar columns act on [dx; du-bar], the T column (d val / dT at the gains' iterate) on dT
            F.G[lane] = ctrl_val(F.dz, Gb.tpn, cr, d) + ctrl_dval(Gb.zn, Gb.tpn, cr, d) * F.dz[18];
          } else if (lane < 54) {
            F.xnx[lane - 45] = next_x(F.zn, F.tpn, lane - 45);
          } else if (lane < 63) {
            F.qp[lane - 54] = jerk_part(F.zn, F.tpn, lane - 54);
          } else {
            L.val[45] = Tq;
            L.G[45] = Tg;
            F.valn[45] = Tn;
            F.G[45] = F.dz[18];
          }
        }
        WSYNC();
        qsum += knot_cost(Tn, F.qp);
        PLV(int, bad);
        St* sn = Sp_(B.S[wb], k);
        St* yn = Sp_(B.Y[wb], k);
        LANES {
          LV(bad) = 0;
          for (int i = 0; i < RPL; i++) {
            if (!slot_on(i, P)) continue;
            const RowK<Real> rk = row_slot(i, lane, P);
            const int r = rk.r;
            if (r < 0) continue;
            const Real cq = row_c(L.val, rk), cg = row_c(L.G, rk), cn = row_c(F.valn, rk), az = row_lin(F.G, rk);
            const Real s = LV(sq)[i], s_g = LV(sg)[i];
            Real snew = s;
            if (infeas) {  // DDP:680-687
              const Real y = LV(yq)[i], y_g = LV(yg)[i];
              Real ynew = y;
              if (!nogain) {
                ynew = (Real)(St)(fma(-alpha, cg + y_g, y) - az);
                const Real rhat = s_g * (cg + y_g) - (s_g * y_g - gmu);  // DDP:536-537
                snew = (Real)(St)fma(frcp(y_g), fma(alpha, rhat, s_g * az), s);
              }
              LV(bad) |= (int)((ynew < omt * y) | (snew < omt * s));
              LV(plog).mul(ynew);
              LV(serr) += fabs(cn + ynew);
              yn[r] = (St)ynew;
              sn[r] = (St)snew;
            } else {  // DDP:694-703
              if (!nogain) {
                const Real w1 = frcp(cg);
                snew = (Real)(St)(fma(-s_g, fma(w1, az, alpha), s) - alpha * gmu * w1);
              }
              LV(bad) |= (int)((cn > omt * cq) | (snew < omt * s));
              LV(plog).mul(-cn);
              sn[r] = (St)snew;
            }
            LV(nviol) += (int)(cn >= (Real)2.0e-4);
          }
          LV(plog).norm();
          if (lane < 19) stx(Xp(wb, k), lane, F.zn[lane]);
          if (lane < 9) F.xn[lane] = F.xnx[lane];
        }
        if (WAVE_ANY(bad)) alive = 0;  // DDP:684-688, 698-702
        WSYNC();
      }
      count_visits(1, n_visits);
      if (!alive) continue;
      PLV(Real, slog);
      LANES {
        if (lane < 9) {
          stx(Xp(wb, N), lane, F.xn[lane]);
          L.z[lane] = F.xn[lane] - (Real)B.xd[(size_t)b * 9 + lane];
        }
      }
      WSYNC();
      const double pterm = terminal_sq();
      WSYNC();
      LANES { LV(slog) = LV(plog).value(); }
      TrialRes res;
      res.alive = 1; res.step = step; res.neg = neg; res.stepsize = stepsize; res.qsum = qsum;
      res.cost = qsum + 0.5 * B.k.w_term * pterm;  // DDP:716-717
      res.sumlog = WAVE_SUM_D(slog);
      res.errsum = WAVE_SUM_D(serr);
      res.viol = WAVE_SUM_I(nviol);
      consider(res, cur, infeas, mu_d, nfilter, filt, A);
      A.buf = wb;
    }
    commit_search(A);
  }

  // k_stuck: the last trip of the outer loop of a trajectory whose backward pass got stuck (DDP:311-396; see iterate_once)
  DDP_DEV void stuck_tail() {
    st.rtn = 0;
    st.done = 0;
    stale_fwd_pass();
    exit_rules();  // ends the solve: -3, 2, the line-init exit, or -4 (DDP:392-396)
  }

  // ---- one trip of the outer loop (DDP:295-412).  Sets st.done when the loop breaks. ------------
  // helper != 0: no trip at all - this wave joins what trajectory b's owner has open: 1 its line search (fwd_pass),
  // 2 its backward sweep (bwd_front_run)
  DDP_DEV void iterate_once(int helper = 0) {
    // The helpers' half of a shared backward sweep: a waiting wave (helper == 2) joins the sweep that trajectory b's owner
    // has open; in the tests' forced split the owner itself plays the helper before each of its sweeps.
    BwdShare* bsf = bshare_slot();
    int f_tag = 0, f_cur = 0, f_infeas = 0;
    double f_mu = 0.0;
    if (helper == 2 && (bsf == nullptr || !bs_enter(bsf, f_tag, f_cur, f_infeas, f_mu))) return;
#if defined(DDP_TIMELINE) && !defined(DIRECT_EMULATE)
    unsigned long long* tl_ = (!helper && B.tl != nullptr && st.fwd_passes < kTimelineDepth) ? B.tl + ((size_t)b * kTimelineDepth + st.fwd_passes) * 4 : nullptr;
    if (tl_ != nullptr && threadIdx.x == 0) { tl_[0] = __builtin_amdgcn_s_memrealtime(); tl_[3] = 0; }
    tl_split_ = 0;
#endif
    if (helper != 1) {
#pragma unroll 1
      while (true) {  // DDP:297-310
        if (!helper) {
          bwd_share_open();
          if (sh_tag != 0 && B.bforce) {
            f_tag = sh_tag; f_cur = DDP_UNIFORM_I(st.cur); f_infeas = DDP_UNIFORM_I(st.infeas); f_mu = st.mu;
          }
        }
        if constexpr (kSplit) {
          if (f_tag != 0) front_cold<Real, St, RPL>((const Batch<St>*)B.self, (LdsPtr<Lds>)&L, b, N, f_tag, f_cur, f_infeas, f_mu);
        }
        f_tag = 0;
        if (helper == 2) {
          bs_leave(bsf);
          return;
        }
        DDP_DBG_MARK(B, 11);
        if (bwd_sweep()) break;
        DDP_DBG_MARK(B, 12);
        note_failed_sweep();
        if (st.reg == 24 && st.bp_failed) st.bp_no_upd++;
        else st.bp_no_upd = 0;
        if (st.bp_no_upd > 20) {
          // The backward pass is stuck (DDP:297-310).  The reference still runs forwardpass() - with the gains its members
          // hold: those of this iterate for the knots some sweep of the retry sequence reached, those of the LAST COMPLETED
          // sweep (another iterate, another mu) for the others.  That pass does not fit the rounds of fwd_pass (they
          // regenerate the slack / dual gains from the current iterate alone) and it always ends the solve (DDP:392-396 at
          // the latest): the trajectory leaves the hot kernel here and k_stuck (stuck_tail) finishes it.
          // (A return from INSIDE the loop: the same test in front of fwd_pass costs the hot kernel 4 %, wave-uniform or
          // not - same-box A/B, round 6.)
          st.rtn = kRtnStuckPending;
          st.done = 1;
          return;
        }
      }
    }
#if defined(DDP_TIMELINE) && !defined(DIRECT_EMULATE)
    if (tl_ != nullptr && threadIdx.x == 0) { tl_[1] = __builtin_amdgcn_s_memrealtime(); tl_[3] = (unsigned long long)tl_split_; }
#endif
    DDP_DBG_MARK(B, 20);
    fwd_pass(helper);
    DDP_DBG_MARK(B, 30);
    if (helper) return;
#if defined(DDP_TIMELINE) && !defined(DIRECT_EMULATE)
    if (tl_ != nullptr && threadIdx.x == 0) tl_[2] = __builtin_amdgcn_s_memrealtime();
#endif
    DDP_MARK("X_A");
    exit_rules();
    DDP_DBG_MARK(B, 31);
  }

  // what follows forwardpass() in one trip of the outer loop (DDP:312-410): bookkeeping and the exit rules
  DDP_DEV void exit_rules() {
    st.fwd_passes++;
    if (st.neg_time) {  // DDP:317-326
      st.rtn = -3;
      st.done = 1;
      return;
    }
    const double prev_cost = st.prev_cost;
    st.prev_cost = st.cost;
    if (!B.k.fixed_iters && fmax(st.opterr, st.mu) <= B.k.tol) {  // DDP:335-338
      st.done = 1;
      return;
    }
    if (st.opterr <= 0.2 * st.mu) {  // DDP:340-344
      st.mu = fmax(B.k.tol / 10.0, fmin(0.2 * st.mu, pow(st.mu, 1.2)));
      reset_filter();
      st.reg = 0;
      st.bp_failed = 0;
    }
    if (st.viol == 0 && !B.k.fixed_iters) {  // DDP:346-390
      if (B.k.zero_init) {
        st.infeas_ref = 0;
        st.rtn = 2;
        st.done = 1;
        return;
      }
      const double d = st.cost - prev_cost;
      if (!B.k.line_init) {
        if (d * d < prev_cost * 1.0e-2 && st.opterr < 5.0e1) {
          st.rtn = 1;
          st.done = 1;
          return;
        }
      } else if (d * d < prev_cost * 0.01) {
        st.line_failed = 0;
        st.done = 1;
        return;
      }
    }
    if (st.bp_no_upd > 20) {  // DDP:392-396
      st.rtn = -4;
      st.done = 1;
      return;
    }
    if (B.k.line_init) {  // DDP:398-409
      if (st.stepsize < 1.0e-6) st.no_upd++;
      else st.no_upd = 0;
      if (st.no_upd > 100) {
        st.done = 1;
        return;
      }
    }
  }

  DDP_DEV void iterate(int n_iters, int helper = 0) {
#pragma unroll 1
    for (int it = 0; it < n_iters; it++) {
      if (!helper) {
        if (DDP_UNIFORM_I(st.done)) break;
        if (st.iter >= B.k.iter_max) {
          st.done = 1;
          break;
        }
      }
#if !defined(DIRECT_EMULATE)
      // a later trip of the same ticket: the iterate this wave has just written must be in memory before a helper of
      // the next sweep reads it (the first trip's iterate was released by the previous ticket's owner)
      if (!helper && it > 0 && bshare_slot() != nullptr) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
      iterate_once(helper);
      if (helper) break;
      if (!st.done) {
        st.iter++;
        if (st.iter >= B.k.iter_max) st.done = 1;
      }
    }
  }
};

// The launch's Batch as the callee sees it: the device copy itself.  The callee's arguments arrive in vector registers;
// the pointer is made wave-uniform, so that everything loaded through it is uniform for the compiler as well (a
// kernel argument's fields are; the sweep pins some of them in scalar registers).
template <typename St>
DDP_DEV const Batch<St>& batch_of(const Batch<St>* Bg) {
#if defined(DIRECT_EMULATE)
  return *Bg;
#else
  const unsigned long long a = (unsigned long long)Bg;
  const unsigned long long u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                               (unsigned)__builtin_amdgcn_readfirstlane((int)a);
  typedef __attribute__((address_space(1))) const Batch<St> GB;
  return *(const Batch<St>*)(GB*)u;
#endif
}
template <typename Real, typename St, int RPL>
DDP_COLD void front_cold(const Batch<St>* Bg, LdsPtr<WaveLds<Real, St, RPL>> lds, int b, int N, int tag, int cur, int infeas, double mu) {
  const Batch<St>& Bl = batch_of(Bg);
  Wave<Real, St, RPL, true> W(Bl, *(WaveLds<Real, St, RPL>*)lds, DDP_UNIFORM_I(b));
  W.N = DDP_UNIFORM_I(N);
  W.bwd_front_run(&Bl.bshare[W.b], DDP_UNIFORM_I(tag), DDP_UNIFORM_I(cur), DDP_UNIFORM_I(infeas), mu);
}
template <typename Real, typename St, int RPL>
DDP_COLD void back_cold(const Batch<St>* Bg, LdsPtr<WaveLds<Real, St, RPL>> lds, int b, int N, void* io) {
  const Batch<St>& Bl = batch_of(Bg);
  Wave<Real, St, RPL, true> W(Bl, *(WaveLds<Real, St, RPL>*)lds, DDP_UNIFORM_I(b));
  W.N = DDP_UNIFORM_I(N);
  W.bwd_back_run(*(typename Wave<Real, St, RPL, true>::BackIO*)io);
}

// ---- results (DDP:414-437 and the getters of DDPH:299-340) --------------------------------------
template <typename Real>
struct OutPtrs {
  int32_t *rtn, *iter_used, *fwd_passes;
  uint8_t *infeas_out, *line_failed_out;
  Real *cost, *costq, *jerk_cost, *terminal_norm2, *opterr, *mu, *bez, *poly, *T;
};

template <typename Real, typename St, int RPL, bool SH>
DDP_DEV void finish_wave(Wave<Real, St, RPL, SH>& W, const OutPtrs<St>& O) {
  const Batch<St>& B = W.B;
  const int b = W.b, N = W.N, buf = W.st.cur;
  PLV(Real, jc);
  LANES {
    LV(jc) = 0;
    for (int k = lane; k < N; k += 64) {
      const St* zs = W.Xp(buf, k);
      Real zz[19];
      for (int a = 0; a < 19; a++) zz[a] = W.ldx(zs, a);
      Real T = zz[18], acc = 0;  // finalroll, DDP:1624-1634
      for (int a = 0; a < 3; a++)
        for (int a2 = 0; a2 < 3; a2++) {
          Real r = W.L.Rc[a * 3 + a2] * powi(T, a + a2 + 1);
          acc += r * (zz[9 + 3 * a] * zz[9 + 3 * a2] + zz[10 + 3 * a] * zz[10 + 3 * a2] + zz[11 + 3 * a] * zz[11 + 3 * a2]);
        }
      LV(jc) += acc;
      Real poly[18];  // sysparam2polyFunc, DDP:814-823
      for (int a = 0; a < 9; a++) poly[a] = ((a / 3 == 2) ? (Real)0.5 : (Real)1) * zz[a];
      for (int a = 0; a < 9; a++) poly[9 + a] = zz[9 + a];
      size_t row = ((size_t)b * B.nmax + k) * 18;
      if (O.poly)
        for (int a = 0; a < 18; a++) O.poly[row + a] = (St)poly[a];
      if (O.T) O.T[(size_t)b * B.nmax + k] = (St)T;
      if (O.bez) {  // poly2bezFunc + layout swap, DDP:799-812, 430-436
        Real invT = (Real)1 / T;
        for (int j = 0; j < 6; j++)
          for (int d = 0; d < 3; d++) {
            Real acc2 = 0;
            for (int i = 0; i < 6; i++) acc2 += (invT * poly[3 * i + d]) * powi(T, i) * (Real)kMono2Bez[i][j];
            O.bez[row + d * 6 + j] = (St)acc2;
          }
      }
    }
  }
  LANES {  // rows past the trajectory's last segment read zero (ragged batches)
    for (int k = N + lane; k < B.nmax; k += 64) {
      const size_t row = ((size_t)b * B.nmax + k) * 18;
      if (O.T) O.T[(size_t)b * B.nmax + k] = (St)0;
      for (int a = 0; a < 18; a++) {
        if (O.poly) O.poly[row + a] = (St)0;
        if (O.bez) O.bez[row + a] = (St)0;
      }
    }
  }
  double jsum = WAVE_SUM_D(jc);
  PLV(Real, tn);
  LANES {
    LV(tn) = 0;
    if (lane < 9) {
      Real d = W.ldx(W.Xp(buf, N), lane) - (Real)B.xd[(size_t)b * 9 + lane];
      LV(tn) = d * d;
    }
  }
  double tsum = WAVE_SUM_D(tn);
  LANES {
    if (lane == 0) {
      const TrajState& s = W.st;
      if (O.rtn) O.rtn[b] = s.rtn;
      if (O.iter_used) O.iter_used[b] = s.iter;
      if (O.fwd_passes) O.fwd_passes[b] = s.fwd_passes;
      if (O.infeas_out) O.infeas_out[b] = (uint8_t)s.infeas_ref;
      if (O.line_failed_out) O.line_failed_out[b] = (uint8_t)s.line_failed;
      if (O.cost) O.cost[b] = (St)s.cost;
      if (O.costq) O.costq[b] = (St)s.costq;
      if (O.jerk_cost) O.jerk_cost[b] = (St)jsum;
      if (O.terminal_norm2) O.terminal_norm2[b] = (St)tsum;
      if (O.opterr) O.opterr[b] = (St)s.opterr;
      if (O.mu) O.mu[b] = (St)s.mu;
    }
  }
}

// ---- stepwise interface helpers: dense read-out / injection of solver fields --------------------
// Field ids and layouts: include/direct_ddp.h (direct_field_t); nc_max = 6*pmax + 55.
template <typename Real, typename St, int RPL, bool SH>
DDP_DEV void get_field_wave(Wave<Real, St, RPL, SH>& W, int field, St* dst) {
  const Batch<St>& B = W.B;
  const int b = W.b, N = W.N, buf = W.st.cur, ncm = 6 * B.pmax + 55;
  if (field == 9) {
    const TrajState& s = W.st;
    St* o = dst + (size_t)b * 16;
    LANES {
      if (lane == 0) {
        o[0] = (St)s.cost; o[1] = (St)s.costq; o[2] = (St)s.logcost; o[3] = (St)s.err;
        o[4] = (St)s.mu; o[5] = (St)s.reg; o[6] = (St)s.opterr; o[7] = (St)s.stepsize;
        o[8] = (St)s.step; o[9] = (St)s.fp_failed; o[10] = (St)s.bp_failed; o[11] = (St)s.rtn;
        o[12] = (St)s.iter; o[13] = (St)s.done; o[14] = (St)s.nfilter; o[15] = (St)s.infeas;
      }
    }
    return;
  }
  if (field == 0) {
    LANES {
      for (int e = lane; e < (N + 1) * 9; e += 64) dst[(size_t)b * (B.nmax + 1) * 9 + e] = (St)W.ldx(W.Xp(buf, e / 9), e % 9);
    }
    return;
  }
  for (int k = 0; k < N; k++) {
    const int P = W.np_(k), nc = 6 * P + 55;
    size_t rowbase = ((size_t)b * B.nmax + k) * ncm;
    if (field == 1) {
      LANES { if (lane < 10) dst[((size_t)b * B.nmax + k) * 10 + lane] = (St)W.ldx(W.Xp(buf, k), 9 + lane); }
    } else if (field == 5) {
      LANES { if (lane < 10) dst[((size_t)b * B.nmax + k) * 10 + lane] = W.KUp(k)[lane]; }
    } else if (field == 6) {
      LANES { for (int e = lane; e < 90; e += 64) dst[((size_t)b * B.nmax + k) * 90 + e] = W.KUp(k)[10 + e]; }
    } else if (field == 4) {
      LANES {
        if (lane < 19) W.L.z[lane] = W.ldx(W.Xp(buf, k), lane);
        for (int e = lane; e < 4 * P; e += 64) W.L.pl[e] = W.planes_(k)[e];
      }
      WSYNC();
      const Real T = W.L.z[18];
      LANES { if (lane < 8) W.L.tp[lane] = powi(T, lane); }
      WSYNC();
      LANES {
        if (lane < 45) W.L.val[lane] = W.ctrl_val(W.L.z, W.L.tp, lane / 3, lane % 3);
        else if (lane == 45) W.L.val[45] = T;
      }
      WSYNC();
      LANES {
        for (int i = 0; i < RPL; i++) {
          if (!W.slot_on(i, P)) continue;
          const RowK<Real> rk = W.row_slot(i, lane, P);
          if (rk.r >= 0) dst[rowbase + rk.r] = (St)W.row_c(W.L.val, rk);
        }
      }
      WSYNC();
    } else {
      const St* src = (field == 2) ? W.Sp_(B.S[buf], k) : (field == 3) ? W.Sp_(B.Y[buf], k)
                        : (field == 7) ? W.Sp_(B.KS, k) : W.Sp_(B.KY, k);
      LANES { for (int r = lane; r < nc; r += 64) dst[rowbase + r] = src[r]; }
    }
  }
}

template <typename Real, typename St, int RPL, bool SH>
DDP_DEV void set_field_wave(Wave<Real, St, RPL, SH>& W, int field, const St* src) {
  const Batch<St>& B = W.B;
  const int b = W.b, N = W.N, buf = W.st.cur, ncm = 6 * B.pmax + 55;
  if (field == 0) {
    LANES {
      for (int e = lane; e < (N + 1) * 9; e += 64) W.stx(W.Xp(buf, e / 9), e % 9, (Real)src[(size_t)b * (B.nmax + 1) * 9 + e]);
    }
    return;
  }
  for (int k = 0; k < N; k++) {
    const int nc = 6 * W.np_(k) + 55;
    size_t rowbase = ((size_t)b * B.nmax + k) * ncm;
    if (field == 1) {
      LANES { if (lane < 10) W.stx(W.Xp(buf, k), 9 + lane, (Real)src[((size_t)b * B.nmax + k) * 10 + lane]); }
    } else if (field == 2 || field == 3) {
      St* d = (field == 2) ? W.Sp_(B.S[buf], k) : W.Sp_(B.Y[buf], k);
      LANES { for (int r = lane; r < nc; r += 64) d[r] = src[rowbase + r]; }
    }
  }
}

}  // namespace direct
