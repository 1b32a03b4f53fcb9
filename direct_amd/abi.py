"""ctypes mirror of include/direct_ddp.h (the C-ABI of the batched IPDDP optimiser).

Field order and types must match the header exactly; tests/test_abi.py checks the struct sizes
against the compiled library.  Reference interface this replaces:
global_planner/include/global_planner/ddp_optimizer.h:267-289 (polyCurveGeneration) and
:299-340 (getters).
"""
import ctypes as C

import numpy as np

NX = 9
NU = 10
P_LIMIT = 128

DIRECT_OK = 0
DIRECT_ERR_INVALID = 1
DIRECT_ERR_UNSUPPORTED = 2
DIRECT_ERR_DEVICE = 3
DIRECT_ERR_NO_DEVICE = 4

F32 = 0
F64 = 1
MEM_HOST = 0
MEM_DEVICE = 1

RTN_DONE = 0
RTN_FEAS_OPT = 1
RTN_FEAS_FOUND = 2
RTN_NEG_TIME = -3
RTN_BP_STUCK = -4
RTN_INVALID = -100
RTN_SCHED_ERROR = -101

FIELD_X, FIELD_U, FIELD_S, FIELD_Y, FIELD_C = 0, 1, 2, 3, 4
FIELD_KU, FIELD_KUU, FIELD_KS, FIELD_KY, FIELD_SCALARS = 5, 6, 7, 8, 9
SCALAR_NAMES = ("cost", "costq", "logcost", "err", "mu", "reg", "opterr", "stepsize", "step",
                "fp_failed", "bp_failed", "rtn", "iter", "done", "filter_n", "infeas")


class Params(C.Structure):
    _fields_ = [
        ("max_vel", C.c_double), ("max_acc", C.c_double), ("w_snap", C.c_double),
        ("w_terminal", C.c_double), ("w_time", C.c_double),
        ("iter_max", C.c_int32), ("time_power", C.c_int32), ("zero_init", C.c_int32),
        ("line_init", C.c_int32), ("minvo", C.c_int32), ("infeas", C.c_int32),
        ("fixed_iters", C.c_int32), ("exact_dt", C.c_int32),
    ]


class BatchIn(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("n_seg_max", C.c_int32), ("p_max", C.c_int32), ("mem", C.c_int32),
        ("n_seg", C.c_void_p), ("x0", C.c_void_p), ("xd", C.c_void_p), ("T0", C.c_void_p),
        ("n_planes", C.c_void_p), ("planes", C.c_void_p), ("seeds", C.c_void_p),
        ("init_bez", C.c_void_p), ("infeas_in", C.c_void_p), ("init_poly", C.c_void_p),
    ]


class BatchOut(C.Structure):
    _fields_ = [
        ("mem", C.c_int32),
        ("rtn", C.c_void_p), ("iter_used", C.c_void_p), ("fwd_passes", C.c_void_p),
        ("infeas_out", C.c_void_p), ("line_failed_out", C.c_void_p),
        ("cost", C.c_void_p), ("costq", C.c_void_p), ("jerk_cost", C.c_void_p),
        ("terminal_norm2", C.c_void_p), ("opterr", C.c_void_p), ("mu", C.c_void_p),
        ("bez", C.c_void_p), ("poly", C.c_void_p), ("T", C.c_void_p),
    ]


class SampleIn(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("n_seg_max", C.c_int32), ("capacity", C.c_int32), ("derivs", C.c_int32),
        ("mem", C.c_int32),
        ("n_seg", C.c_void_p), ("bez", C.c_void_p), ("T", C.c_void_p), ("dt", C.c_double),
        ("p_max", C.c_int32), ("n_planes", C.c_void_p), ("planes", C.c_void_p),
    ]


class SampleOut(C.Structure):
    _fields_ = [
        ("count", C.c_void_p), ("seg_first", C.c_void_p), ("pos", C.c_void_p), ("vel", C.c_void_p),
        ("acc", C.c_void_p), ("length", C.c_void_p), ("vmax", C.c_void_p), ("amax", C.c_void_p),
        ("cmax", C.c_void_p),
    ]


class LaunchInfo(C.Structure):  # direct_ddp_launch_info_t
    _fields_ = [
        ("dynamic", C.c_int32), ("shared_search", C.c_int32), ("pair_trials", C.c_int32), ("single_steps", C.c_int32),
        ("n_buffers", C.c_int32), ("resident_waves", C.c_int32), ("batch", C.c_int32), ("shared_sweep", C.c_int32),
        ("bwd_knot_visits", C.c_uint64), ("fwd_knot_visits", C.c_uint64),
    ]


class Config(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("device", C.c_int32), ("max_batch", C.c_int32),
        ("n_seg_max", C.c_int32), ("p_max", C.c_int32), ("reserved", C.c_int32),
    ]


# Launch-file values that reach polyCurveGeneration
# (global_planner/launch/global_planner.launch:11-13, 61-70; teach_repeat_planner.cpp:895-897, 919-921)
FLAG_STATIC_SCHEDULE, FLAG_YIELD = 1, 2   # direct_ddp_config_t.reserved (include/direct_ddp.h)


def phase0_params(**kw):
    """Phase 0: zero init, infeasible start, w = 1/1/1, iter_max_zero = 50."""
    d = dict(max_vel=2.0, max_acc=2.0, w_snap=1.0, w_terminal=1.0, w_time=1.0, iter_max=50,
             time_power=2, zero_init=1, line_init=0, minvo=0, infeas=1, fixed_iters=0, exact_dt=0)
    d.update(kw)
    return Params(**d)


def phase1_params(**kw):
    """Phase 1: warm start from phase 0, w_snap 1, w_terminal 100, w_time 20, iter_max 100."""
    d = dict(max_vel=2.0, max_acc=2.0, w_snap=1.0, w_terminal=100.0, w_time=20.0, iter_max=100,
             time_power=2, zero_init=0, line_init=0, minvo=0, infeas=0, fixed_iters=0, exact_dt=0)
    d.update(kw)
    return Params(**d)


def real_dtype(dtype):
    return np.float64 if dtype == F64 else np.float32


def _ptr(a):
    return None if a is None else a.ctypes.data


class HostBatch:
    """Host-side (numpy) batch of corridors in the flat layout of include/direct_ddp.h."""

    def __init__(self, n_seg, x0, xd, T0, n_planes, planes, seeds=None, init_bez=None,
                 infeas_in=None, dtype=np.float64, init_poly=None):
        self.dtype = np.dtype(dtype)
        self.n_seg = np.ascontiguousarray(n_seg, dtype=np.int32)
        self.batch = int(self.n_seg.shape[0])
        self.T0 = np.ascontiguousarray(T0, dtype=self.dtype)
        self.n_seg_max = int(self.T0.shape[1])
        self.x0 = np.ascontiguousarray(x0, dtype=self.dtype).reshape(self.batch, 9)
        self.xd = np.ascontiguousarray(xd, dtype=self.dtype).reshape(self.batch, 9)
        self.n_planes = np.ascontiguousarray(n_planes, dtype=np.int32).reshape(self.batch, self.n_seg_max)
        self.planes = np.ascontiguousarray(planes, dtype=self.dtype)
        self.p_max = int(self.planes.shape[2])
        assert self.planes.shape == (self.batch, self.n_seg_max, self.p_max, 4)
        self.seeds = None if seeds is None else np.ascontiguousarray(seeds, dtype=self.dtype)
        self.init_bez = None if init_bez is None else np.ascontiguousarray(init_bez, dtype=self.dtype)
        self.infeas_in = None if infeas_in is None else np.ascontiguousarray(infeas_in, dtype=np.uint8)
        self.init_poly = None if init_poly is None else np.ascontiguousarray(init_poly, dtype=self.dtype)
        self.omit_T0 = False   # True: the C struct carries T0 = NULL and the library runs initTimeAllocation on the device

    def without_T0(self):
        """the same corridors with the durations left to the library (T0 = NULL: device-side initTimeAllocation from x0 / xd / seeds)"""
        b = HostBatch(self.n_seg, self.x0, self.xd, self.T0, self.n_planes, self.planes, self.seeds, self.init_bez,
                      self.infeas_in, dtype=self.dtype, init_poly=self.init_poly)
        b.omit_T0 = True
        return b

    @property
    def nc_max(self):
        return 6 * self.p_max + 55

    def astype(self, dtype):
        return HostBatch(self.n_seg, self.x0, self.xd, self.T0, self.n_planes, self.planes,
                         self.seeds, self.init_bez, self.infeas_in, dtype=dtype, init_poly=self.init_poly)

    def select(self, idx):
        idx = np.atleast_1d(np.asarray(idx))
        f = lambda a: None if a is None else a[idx]
        return HostBatch(self.n_seg[idx], self.x0[idx], self.xd[idx], self.T0[idx],
                         self.n_planes[idx], self.planes[idx], f(self.seeds), f(self.init_bez),
                         f(self.infeas_in), dtype=self.dtype, init_poly=f(self.init_poly))

    def with_init(self, init_bez, T0=None, infeas_in=None, init_poly=None):
        return HostBatch(self.n_seg, self.x0, self.xd, self.T0 if T0 is None else T0, self.n_planes,
                         self.planes, self.seeds, init_bez, infeas_in, dtype=self.dtype, init_poly=init_poly)

    def phase1_inputs(self, res0, monomial=True):
        """The second polyCurveGeneration call of fastTrajPlanning from a phase-0 result (teach_repeat_planner.cpp:911-921),
        as direct_ddp_plan_batch chains it on the device: UpdateTime where phase 0 returned 2, and the warm start
        getBezCoeff() -> initbezCoeff.  monomial: the same warm start as getPolyCoeff rows (what float storage needs,
        include/direct_ddp.h): where phase 0 did NOT return 2 the reference converts its control points back with the
        caller's durations T_1 instead of phase 0's T_0 (ddp_optimizer.cpp:167-193 after 799-812), i.e. coefficient c_i
        becomes c_i (T_0 / T_1)^(i-1) - applied here in double (k_chain of direct_ddp.hip does the same)."""
        found = (np.asarray(res0.rtn) == 2)[:, None]
        T1 = np.where(found, res0.T, self.T0)
        infeas = np.asarray(res0.infeas_out).astype(np.uint8)
        if not monomial:
            return self.with_init(res0.bez, T0=T1, infeas_in=infeas)
        T0d, T1d = np.asarray(res0.T, np.float64), np.asarray(T1, np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            q = np.where(found | (T1d == 0.0), 1.0, T0d / T1d)
        poly = np.asarray(res0.poly, np.float64).reshape(self.batch, self.n_seg_max, 6, 3).copy()
        f = 1.0 / q
        for c in range(6):
            poly[:, :, c, :] = np.where(found[:, :, None], poly[:, :, c, :], poly[:, :, c, :] * f[:, :, None])
            f = f * q
        return self.with_init(None, T0=T1, infeas_in=infeas, init_poly=poly.reshape(self.batch, self.n_seg_max, 18))

    def c_struct(self):
        s = BatchIn()
        s.batch, s.n_seg_max, s.p_max, s.mem = self.batch, self.n_seg_max, self.p_max, MEM_HOST
        s.n_seg = _ptr(self.n_seg)
        s.x0, s.xd, s.T0 = _ptr(self.x0), _ptr(self.xd), (None if self.omit_T0 else _ptr(self.T0))
        s.n_planes, s.planes = _ptr(self.n_planes), _ptr(self.planes)
        s.seeds, s.init_bez, s.infeas_in = _ptr(self.seeds), _ptr(self.init_bez), _ptr(self.infeas_in)
        s.init_poly = _ptr(self.init_poly)
        return s


class HostResult:
    """Host-side (numpy) result arrays, one entry per problem (ddp_optimizer.h:299-340)."""

    def __init__(self, batch, n_seg_max, dtype=np.float64):
        dt = np.dtype(dtype)
        self.rtn = np.zeros(batch, np.int32)
        self.iter_used = np.zeros(batch, np.int32)
        self.fwd_passes = np.zeros(batch, np.int32)
        self.infeas_out = np.zeros(batch, np.uint8)
        self.line_failed_out = np.zeros(batch, np.uint8)
        self.cost = np.zeros(batch, dt)
        self.costq = np.zeros(batch, dt)
        self.jerk_cost = np.zeros(batch, dt)
        self.terminal_norm2 = np.zeros(batch, dt)
        self.opterr = np.zeros(batch, dt)
        self.mu = np.zeros(batch, dt)
        self.bez = np.zeros((batch, n_seg_max, 18), dt)
        self.poly = np.zeros((batch, n_seg_max, 18), dt)
        self.T = np.zeros((batch, n_seg_max), dt)

    def c_struct(self):
        s = BatchOut()
        s.mem = MEM_HOST
        for name, _ in BatchOut._fields_[1:]:
            setattr(s, name, _ptr(getattr(self, name)))
        return s
