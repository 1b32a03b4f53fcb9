// TEST-ONLY harness: compiles direct_amd/csrc/ddp_wave.h with DIRECT_EMULATE, i.e. every LANES
// block becomes a 64-iteration loop, so the restructured (Kronecker / column-per-lane) algorithm
// of the HIP kernels can be checked against the oracle on a machine without a GPU.  This is NOT a
// CPU fallback: nothing under direct_amd/ builds, loads or calls it; the product library
// (libdirect_ddp.so) contains only gfx950 code and fails with DIRECT_ERR_NO_DEVICE without a GPU.
#define DIRECT_EMULATE 1
#include "../../direct_amd/csrc/ddp_wave.h"
#include "../../include/direct_ddp.h"

#include <cstdlib>
#include <cstring>
#include <vector>

using namespace direct;
#if defined(DDP_EMU_TRACE)
extern "C" void ddp_emu_lds_range(void*, size_t);
#endif

template <typename Real>   // Real = storage type here; the compute type is chosen per call
struct Emu {
  Batch<Real> B;
  int compute64 = 0;
  std::vector<Real> x0, xd, T0, planes, init_bez, init_poly, seeds, X0, X1, X2, S0, S1, S2, Y0, Y1, Y2, KU, KS, KY;
  std::vector<int32_t> n_seg, n_planes;
  std::vector<uint8_t> infeas_in;
  std::vector<double> filt;
  std::vector<TrajState> st;
  std::vector<GainBase> gbase;
  // forced split of the backward sweep (DIRECT_EMU_BSPLIT=1): front halves through hand-over records, as helper waves would
  std::vector<BwdShare> bshare;
  std::vector<int> bflag;
  std::vector<double> brec;
  int rpl;
};

template <typename Cmp, typename Real, int RPL, typename F>
static void for_each_wave(Emu<Real>& E, F f) {
  static WaveLds<Cmp, Real, RPL> lds;
#if defined(DDP_EMU_TRACE)  // tools/lds_trace
  ddp_emu_lds_range(&lds, sizeof lds);
#endif
  for (int b = 0; b < E.B.B; b++) {
    Wave<Cmp, Real, RPL, true> W(E.B, lds, b);  // the sharing instantiation: DIRECT_EMU_BSPLIT=1 forces its hand-over path
    f(W);
  }
}

template <typename Cmp, typename Real, typename F>
static void dispatch_c(Emu<Real>& E, F f) {
  if (E.rpl <= 2) for_each_wave<Cmp, Real, 2>(E, f);
  else if (E.rpl == 3) for_each_wave<Cmp, Real, 3>(E, f);
  else if (E.rpl == 4) for_each_wave<Cmp, Real, 4>(E, f);
  else if (E.rpl == 5) for_each_wave<Cmp, Real, 5>(E, f);
  else if (E.rpl == 6) for_each_wave<Cmp, Real, 6>(E, f);
  else if (E.rpl == 7) for_each_wave<Cmp, Real, 7>(E, f);
  else if (E.rpl == 8) for_each_wave<Cmp, Real, 8>(E, f);
  else if (E.rpl <= 10) for_each_wave<Cmp, Real, 10>(E, f);
  else if (E.rpl <= 12) for_each_wave<Cmp, Real, 12>(E, f);
  else for_each_wave<Cmp, Real, 14>(E, f);
}
template <typename Real, typename F>
static void dispatch(Emu<Real>& E, F f) {
  if (E.compute64) dispatch_c<double, Real>(E, f);
  else dispatch_c<Real, Real>(E, f);
}

template <typename Real>
static Emu<Real>* emu_begin_t(const direct_ddp_params_t* p, const direct_ddp_batch_in_t* in, int compute64) {
  Emu<Real>* E = new Emu<Real>();
  E->compute64 = compute64;
  int B = in->batch, nm = in->n_seg_max, pm = in->p_max;
  int ncm = 6 * pm + 55;
  E->rpl = (ncm + 63) / 64;
  Batch<Real>& Bt = E->B;
  memset(&Bt, 0, sizeof(Bt));
  Bt.B = B; Bt.nmax = nm; Bt.pmax = pm; Bt.ncs = ncm; Bt.fcap = p->iter_max + 4;
  auto cp = [](std::vector<Real>& v, const void* src, size_t n) {
    v.assign((const Real*)src, (const Real*)src + n);
  };
  E->n_seg.assign(in->n_seg, in->n_seg + B);
  E->n_planes.assign(in->n_planes, in->n_planes + (size_t)B * nm);
  cp(E->x0, in->x0, (size_t)B * 9);
  cp(E->xd, in->xd, (size_t)B * 9);
  cp(E->T0, in->T0, (size_t)B * nm);
  cp(E->planes, in->planes, (size_t)B * nm * pm * 4);
  if (in->init_bez) cp(E->init_bez, in->init_bez, (size_t)B * nm * 18);
  if (in->init_poly) cp(E->init_poly, in->init_poly, (size_t)B * nm * 18);
  if (in->seeds) cp(E->seeds, in->seeds, (size_t)B * nm * 3);
  E->infeas_in.assign(B, (uint8_t)p->infeas);
  if (in->infeas_in) E->infeas_in.assign(in->infeas_in, in->infeas_in + B);
  size_t nx = (size_t)B * (nm + 1) * x_stride<Real>(), ns = (size_t)B * nm * ncm;
  E->X0.assign(nx, 0); E->X1.assign(nx, 0); E->X2.assign(nx, 0);
  E->S0.assign(ns, 0); E->S1.assign(ns, 0); E->S2.assign(ns, 0);
  E->Y0.assign(ns, 0); E->Y1.assign(ns, 0); E->Y2.assign(ns, 0);
  E->KU.assign((size_t)B * nm * 100, 0); E->KS.assign(ns, 0); E->KY.assign(ns, 0);
  E->filt.assign((size_t)B * Bt.fcap * 2, 0.0);
  E->st.assign(B, TrajState());
  E->gbase.assign(B, GainBase());
  Bt.n_seg = E->n_seg.data(); Bt.x0 = E->x0.data(); Bt.xd = E->xd.data(); Bt.T0 = E->T0.data();
  Bt.n_planes = E->n_planes.data(); Bt.planes = E->planes.data();
  Bt.init_bez = in->init_bez ? E->init_bez.data() : nullptr;
  Bt.init_poly = in->init_poly ? E->init_poly.data() : nullptr;
  Bt.seeds = in->seeds ? E->seeds.data() : nullptr;
  Bt.infeas_in = E->infeas_in.data();
  Bt.X[0] = E->X0.data(); Bt.X[1] = E->X1.data(); Bt.X[2] = E->X2.data();
  Bt.S[0] = E->S0.data(); Bt.S[1] = E->S1.data(); Bt.S[2] = E->S2.data();
  Bt.Y[0] = E->Y0.data(); Bt.Y[1] = E->Y1.data(); Bt.Y[2] = E->Y2.data(); Bt.KU = E->KU.data(); Bt.KS = E->KS.data();
  Bt.KY = E->KY.data(); Bt.filt = E->filt.data(); Bt.st = E->st.data();
  Bt.gbase = E->gbase.data();
  Bt.nbuf = 3; Bt.help = nullptr; Bt.sched_err = nullptr; Bt.visits = nullptr;
  if (getenv("DIRECT_EMU_BSPLIT") && atoi(getenv("DIRECT_EMU_BSPLIT")) != 0) {
    E->bshare.assign(B, BwdShare());
    memset(E->bshare.data(), 0, sizeof(BwdShare) * (size_t)B);
    E->bflag.assign((size_t)B * nm, 0);
    E->brec.assign((size_t)B * nm * kRecDoubles, 0.0);
    Bt.bshare = E->bshare.data(); Bt.bflag = E->bflag.data(); Bt.brec = E->brec.data(); Bt.bforce = 1;
    Bt.self = &E->B;
  }
  SolveConst& k = Bt.k;
  k.max_vel = p->max_vel; k.max_acc = p->max_acc; k.w_snap = p->w_snap; k.w_term = p->w_terminal;
  k.w_time = p->w_time; k.reg_base = p->zero_init ? 1.6 : 4.0; k.shift = p->minvo ? 0.0 : 2.0e-4;
  k.tol = 1.0e-7; k.iter_max = p->iter_max; k.time_power = p->time_power; k.zero_init = p->zero_init;
  k.line_init = p->line_init; k.minvo = p->minvo; k.fixed_iters = p->fixed_iters; k.exact_dt = p->exact_dt;
  k.pair_trials = getenv("DIRECT_EMU_PAIR") ? atoi(getenv("DIRECT_EMU_PAIR")) : 1;
  dispatch(*E, [&](auto& W) {
    W.init_tables();
    memset(&W.st, 0, sizeof(W.st));
    W.begin();
    W.store_state();
  });
  return E;
}

template <typename Real, typename F>
static void with_state(Emu<Real>& E, F f) {
  dispatch(E, [&](auto& W) {
    W.load_state();
    W.init_tables();
    f(W);
    W.store_state();
  });
}

struct EmuHandle {
  int dtype;
  void* p;
};

#if defined(DDP_EMU_TRACE)  // tools/lds_trace: byte offsets of the LDS members of the two-slot float-storage wave
#include <cstddef>
#include <cstdio>
extern "C" void ddp_emu_layout(const char* path) {
  typedef WaveLds<double, float, 2> Lt;
  FILE* f = fopen(path, "w");
  if (!f) return;
#define MEM(m) fprintf(f, "%s %zu\n", #m, offsetof(Lt, m));
  MEM(st) MEM(WbE) MEM(WdE) MEM(Rc) MEM(lt) MEM(lt16) MEM(ones) MEM(tp) MEM(z) MEM(pl) MEM(val) MEM(G) MEM(We) MEM(dval) MEM(fT) MEM(Ru)
  MEM(Rpu) MEM(Rppu) MEM(V) MEM(Sd) MEM(dl) MEM(Vx) MEM(Hxx) MEM(HR) MEM(KU) MEM(Hzx) MEM(drow) MEM(grow) MEM(Sp) MEM(hh)
  MEM(last) MEM(VZ) MEM(UY) MEM(KUr) MEM(ft)
#undef MEM
  fprintf(f, "end %zu\n", sizeof(Lt));
  fclose(f);
}
#endif
extern "C" {
void* emu_begin(int dtype, const direct_ddp_params_t* p, const direct_ddp_batch_in_t* in) {
  EmuHandle* h = new EmuHandle();
  h->dtype = dtype == DIRECT_F64 ? DIRECT_F64 : DIRECT_F32;   // dtype 2: fp32 storage, fp64 arithmetic
  h->p = dtype == DIRECT_F64 ? (void*)emu_begin_t<double>(p, in, 1) : (void*)emu_begin_t<float>(p, in, dtype == 2);
  return h;
}
#define EMU_CALL(body)                                            \
  EmuHandle* h = (EmuHandle*)hv;                                  \
  if (h->dtype == DIRECT_F64) { auto& E = *(Emu<double>*)h->p; typedef double Real; (void)sizeof(Real); body; } \
  else { auto& E = *(Emu<float>*)h->p; typedef float Real; (void)sizeof(Real); body; }

void emu_backward(void* hv) { EMU_CALL(with_state(E, [](auto& W) { if (!W.st.done) W.backward_pass_stepwise(); })) }
// (as k_pass / k_stuck of direct_ddp.hip do)
void emu_forward(void* hv) {
  EMU_CALL(with_state(E, [](auto& W) {
    if (W.st.done) return;
    if (W.st.bp_failed) W.stale_fwd_pass();
    else W.fwd_pass();
  }))
}
void emu_forward_stored(void* hv) { EMU_CALL(with_state(E, [](auto& W) { if (!W.st.done) W.stale_fwd_pass(); })) }
void emu_iterate(void* hv, int n) {
  EMU_CALL(with_state(E, [n](auto& W) {
    W.iterate(n);
    if (W.st.rtn == kRtnStuckPending) W.stuck_tail();
  }))
}
void emu_get_field(void* hv, int field, void* dst) {
  EMU_CALL(with_state(E, [&](auto& W) { get_field_wave(W, field, (Real*)dst); }))
}
void emu_set_field(void* hv, int field, const void* src) {
  EMU_CALL(with_state(E, [&](auto& W) { set_field_wave(W, field, (const Real*)src); }))
}
void emu_finish(void* hv, direct_ddp_batch_out_t* out) {
  EMU_CALL({
    OutPtrs<Real> O;
    O.rtn = out->rtn; O.iter_used = out->iter_used; O.fwd_passes = out->fwd_passes;
    O.infeas_out = out->infeas_out; O.line_failed_out = out->line_failed_out;
    O.cost = (Real*)out->cost; O.costq = (Real*)out->costq; O.jerk_cost = (Real*)out->jerk_cost;
    O.terminal_norm2 = (Real*)out->terminal_norm2; O.opterr = (Real*)out->opterr; O.mu = (Real*)out->mu;
    O.bez = (Real*)out->bez; O.poly = (Real*)out->poly; O.T = (Real*)out->T;
    with_state(E, [&](auto& W) { finish_wave(W, O); });
  })
}
// knots whose front half went through a hand-over record in the last sweeps (forced split; 0 without DIRECT_EMU_BSPLIT)
int emu_split_knots(void* hv) {
  int n = 0;
  EMU_CALL({ for (int v : E.bflag) n += v != 0; })
  return n;
}
void emu_end(void* hv) {
  EmuHandle* h = (EmuHandle*)hv;
  if (h->dtype == DIRECT_F64) delete (Emu<double>*)h->p;
  else delete (Emu<float>*)h->p;
  delete h;
}
}
