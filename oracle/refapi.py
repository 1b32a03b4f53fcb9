"""ctypes front-end of oracle/libdirect_ref.so -- TEST INFRASTRUCTURE (parity unpinned, see
oracle/direct_ref.c).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; nothing under direct_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from direct_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libdirect_ref.so")
    src = os.path.join(_HERE, "direct_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libdirect_ref.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libdirect_ref.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.direct_ref_begin.restype = C.c_void_p
        L.direct_ref_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        for n in ("direct_ref_backwardpass", "direct_ref_forwardpass", "direct_ref_computeall", "direct_ref_end"):
            getattr(L, n).argtypes = [C.c_void_p]
            getattr(L, n).restype = None
        L.direct_ref_iterate.argtypes = [C.c_void_p, C.c_int]
        L.direct_ref_ncmax.argtypes = [C.c_void_p]
        L.direct_ref_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.direct_ref_set.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.direct_ref_solve_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.direct_ref_plan_batch.argtypes = [C.c_void_p] * 5 + [C.c_int]
        L.direct_ref_eval_q.restype = C.c_double
        L.direct_ref_eval_q.argtypes = [C.c_void_p, C.c_void_p]
        L.direct_ref_eval_c.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.direct_ref_eval_nextx.argtypes = [C.c_void_p] * 3
        L.direct_ref_bez2poly.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
        L.direct_ref_poly2bez.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
        L.direct_ref_time_allocation.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        _LIB = L
    return _LIB


TRACE_COLS = ("cost", "costq", "logcost", "err", "mu", "reg", "step", "opterr", "stepsize", "fp_failed", "n_sweeps", "bp_failed")


def _f64(batch):
    return batch if batch.dtype == np.float64 else batch.astype(np.float64)


def solve_batch(params, batch, n_threads=0, trace_cap=0):
    """polyCurveGeneration for every problem of `batch` (fp64).  Returns (HostResult, trace)."""
    batch = _f64(batch)
    res = abi.HostResult(batch.batch, batch.n_seg_max, np.float64)
    cin, cout = batch.c_struct(), res.c_struct()
    trace = np.zeros((batch.batch, trace_cap, len(TRACE_COLS))) if trace_cap else None
    st = lib().direct_ref_solve_batch(C.addressof(params), C.addressof(cin), C.addressof(cout), n_threads,
                                      None if trace is None else trace.ctypes.data, trace_cap)
    if st != 0:
        raise RuntimeError("direct_ref_solve_batch status %d" % st)
    return res, trace


def plan_batch(params0, params1, batch, n_threads=0):
    """fastTrajPlanning protocol (phase 0 -> UpdateTime -> phase 1).  Returns (res0, res1)."""
    batch = _f64(batch)
    r0 = abi.HostResult(batch.batch, batch.n_seg_max)
    r1 = abi.HostResult(batch.batch, batch.n_seg_max)
    cin, c0, c1 = batch.c_struct(), r0.c_struct(), r1.c_struct()
    st = lib().direct_ref_plan_batch(C.addressof(params0), C.addressof(params1), C.addressof(cin),
                                     C.addressof(c0), C.addressof(c1), n_threads)
    if st != 0:
        raise RuntimeError("direct_ref_plan_batch status %d" % st)
    return r0, r1


class Stepper:
    """Stepwise oracle for ONE problem (index b of a batch): per-pass parity tests."""

    def __init__(self, params, batch, b=0):
        self.batch = _f64(batch)
        self.params = params
        self._cin = self.batch.c_struct()
        self.N = int(self.batch.n_seg[b])
        self.h = lib().direct_ref_begin(C.addressof(params), C.addressof(self._cin), b)
        self.ncmax = lib().direct_ref_ncmax(self.h)

    def close(self):
        if self.h:
            lib().direct_ref_end(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def backward(self):
        lib().direct_ref_backwardpass(self.h)

    def forward(self):
        lib().direct_ref_forwardpass(self.h)

    def computeall(self):
        lib().direct_ref_computeall(self.h)

    def iterate(self, n=1):
        return lib().direct_ref_iterate(self.h, n)

    _shapes = {
        abi.FIELD_X: lambda s: (s.N + 1, 9), abi.FIELD_U: lambda s: (s.N, 10),
        abi.FIELD_S: lambda s: (s.N, s.ncmax), abi.FIELD_Y: lambda s: (s.N, s.ncmax),
        abi.FIELD_C: lambda s: (s.N, s.ncmax), abi.FIELD_KU: lambda s: (s.N, 10),
        abi.FIELD_KUU: lambda s: (s.N, 10, 9), abi.FIELD_KS: lambda s: (s.N, s.ncmax),
        abi.FIELD_KY: lambda s: (s.N, s.ncmax), abi.FIELD_SCALARS: lambda s: (16,),
        100: lambda s: (s.N, s.ncmax, 9), 101: lambda s: (s.N, s.ncmax, 10),
        102: lambda s: (s.N, 9, 9), 103: lambda s: (s.N, 9, 10), 104: lambda s: (s.N, 10),
        105: lambda s: (s.N, 10, 10), 106: lambda s: (s.N, s.ncmax, 9), 107: lambda s: (s.N, s.ncmax, 9),
        109: lambda s: (s.N,),
    }

    def get(self, field):
        out = np.zeros(self._shapes[field](self))
        st = lib().direct_ref_get(self.h, field, out.ctypes.data)
        assert st == 0
        return out

    def set(self, field, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        assert a.shape == self._shapes[field](self), (a.shape, self._shapes[field](self))
        st = lib().direct_ref_set(self.h, field, a.ctypes.data)
        assert st == 0

    def scalars(self):
        return dict(zip(abi.SCALAR_NAMES, self.get(abi.FIELD_SCALARS)))

    def filter(self):
        n = int(self.scalars()["filter_n"])
        out = np.zeros((n, 2))
        lib().direct_ref_get(self.h, 108, out.ctypes.data)
        return out


def eval_c(params, x, u, planes):
    x = np.ascontiguousarray(x, np.float64)
    u = np.ascontiguousarray(u, np.float64)
    planes = np.ascontiguousarray(planes, np.float64).reshape(-1, 4)
    P = planes.shape[0]
    c = np.zeros(6 * P + 55)
    lib().direct_ref_eval_c(C.addressof(params), x.ctypes.data, u.ctypes.data, P, planes.ctypes.data, c.ctypes.data)
    return c


def eval_nextx(x, u):
    x = np.ascontiguousarray(x, np.float64)
    u = np.ascontiguousarray(u, np.float64)
    xn = np.zeros(9)
    lib().direct_ref_eval_nextx(x.ctypes.data, u.ctypes.data, xn.ctypes.data)
    return xn


def eval_q(params, u):
    u = np.ascontiguousarray(u, np.float64)
    return lib().direct_ref_eval_q(C.addressof(params), u.ctypes.data)


def bez2poly(bez_il, T):
    b = np.ascontiguousarray(bez_il, np.float64)
    out = np.zeros(18)
    lib().direct_ref_bez2poly(b.ctypes.data, float(T), out.ctypes.data)
    return out


def poly2bez(poly, T):
    p = np.ascontiguousarray(poly, np.float64)
    out = np.zeros(18)
    lib().direct_ref_poly2bez(p.ctypes.data, float(T), out.ctypes.data)
    return out


def time_allocation(n_seg, start, goal, seeds, max_vel=2.0, max_acc=2.0):
    n_seg = np.ascontiguousarray(n_seg, np.int32)
    B = n_seg.shape[0]
    seeds = np.ascontiguousarray(seeds, np.float64)
    nmax = seeds.shape[1]
    start = np.ascontiguousarray(start, np.float64)
    goal = np.ascontiguousarray(goal, np.float64)
    T = np.zeros((B, nmax))
    lib().direct_ref_time_allocation(B, nmax, n_seg.ctypes.data, start.ctypes.data, goal.ctypes.data,
                                     seeds.ctypes.data, max_vel, max_acc, T.ctypes.data)
    return T


def sample_batch(n_seg, bez, T, dt, capacity, derivs=2, n_planes=None, planes=None):
    """The caller's sampling loop (teach_repeat_planner.cpp:1551-1566) for every trajectory of a batch, fp64.
    With planes: also cmax[b] = max over the stored samples of max_p (n_p . x + d_p) against the polytope of the
    sample's own segment (the containment audit; plain numpy over the oracle's samples)."""
    n_seg = np.ascontiguousarray(n_seg, np.int32)
    bez = np.ascontiguousarray(bez, np.float64)
    T = np.ascontiguousarray(T, np.float64)
    B, nm = T.shape
    o = dict(count=np.zeros(B, np.int32), seg_first=np.zeros((B, nm), np.int32), pos=np.zeros((B, capacity, 3)),
             vel=np.zeros((B, capacity, 3)), acc=np.zeros((B, capacity, 3)), length=np.zeros(B), vmax=np.zeros(B),
             amax=np.zeros(B))
    f = lib().direct_ref_sample
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int] + [C.c_void_p] * 7
    for b in range(B):
        o["count"][b] = f(int(n_seg[b]), bez[b].ctypes.data, T[b].ctypes.data, float(dt), capacity, derivs,
                          o["seg_first"][b].ctypes.data, o["pos"][b].ctypes.data, o["vel"][b].ctypes.data,
                          o["acc"][b].ctypes.data, o["length"][b:].ctypes.data, o["vmax"][b:].ctypes.data,
                          o["amax"][b:].ctypes.data)
    if planes is not None:
        planes = np.asarray(planes, np.float64)
        o["cmax"] = np.zeros(B)
        for b in range(B):
            if o["count"][b] < 0:
                continue
            worst = -1.0e300
            for i in range(int(n_seg[b])):
                lo = int(o["seg_first"][b, i])
                hi = int(o["seg_first"][b, i + 1]) if i + 1 < int(n_seg[b]) else int(o["count"][b])
                pts = o["pos"][b, lo:min(hi, capacity)]
                pl = planes[b, i, :int(n_planes[b, i])]
                if len(pts):
                    worst = max(worst, float((pts @ pl[:, :3].T + pl[:, 3]).max()))
            o["cmax"][b] = worst
    return o


def num_threads():
    return lib().direct_ref_num_threads()
