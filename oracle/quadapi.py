"""ctypes front-end of oracle/libquad_ref.so -- TEST INFRASTRUCTURE (checker of the BASELINE-label model, which has
no reference counterpart).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libquad_ref.so")
    src = os.path.join(_HERE, "quad_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libquad_ref.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.quad_ref_begin.restype = C.c_void_p
        L.quad_ref_begin.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.quad_ref_end.argtypes = [C.c_void_p]
        L.quad_ref_iterate.argtypes = [C.c_void_p, C.c_int]
        L.quad_ref_get.argtypes = [C.c_void_p] * 6
        L.quad_ref_jacobians.argtypes = [C.c_void_p] * 5
        L.quad_ref_step.argtypes = [C.c_void_p] * 4
        L.quad_ref_solve_batch.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_int]
        _LIB = L
    return _LIB


SCALARS = ("cost", "reg", "step", "fp_failed", "bp_failed", "iter", "done", "fwd_passes")


class Stepper:
    def __init__(self, params, N, x0, xg):
        self.N = N
        self.x0, self.xg = np.ascontiguousarray(x0, np.float64), np.ascontiguousarray(xg, np.float64)
        self.params = params
        self.h = lib().quad_ref_begin(C.addressof(params), N, self.x0.ctypes.data, self.xg.ctypes.data)

    def iterate(self, n=1):
        return lib().quad_ref_iterate(self.h, n)

    def get(self):
        N = self.N
        x, u, K, kf, sc = np.zeros((N + 1, 12)), np.zeros((N, 4)), np.zeros((N, 4, 12)), np.zeros((N, 4)), np.zeros(8)
        lib().quad_ref_get(self.h, x.ctypes.data, u.ctypes.data, K.ctypes.data, kf.ctypes.data, sc.ctypes.data)
        return dict(x=x, u=u, K=K, kf=kf, **dict(zip(SCALARS, sc)))

    def close(self):
        if self.h:
            lib().quad_ref_end(self.h)
            self.h = None

    def __del__(self):
        self.close()


def jacobians(params, x, u):
    x, u = np.ascontiguousarray(x, np.float64), np.ascontiguousarray(u, np.float64)
    A, B = np.zeros((12, 12)), np.zeros((12, 4))
    lib().quad_ref_jacobians(C.addressof(params), x.ctypes.data, u.ctypes.data, A.ctypes.data, B.ctypes.data)
    return A, B


def step(params, x, u):
    x, u = np.ascontiguousarray(x, np.float64), np.ascontiguousarray(u, np.float64)
    xn = np.zeros(12)
    lib().quad_ref_step(C.addressof(params), x.ctypes.data, u.ctypes.data, xn.ctypes.data)
    return xn


def solve_batch(params, N, x0, xg, n_threads=0):
    x0, xg = np.ascontiguousarray(x0, np.float64), np.ascontiguousarray(xg, np.float64)
    B = x0.shape[0]
    cost, iters = np.zeros(B), np.zeros(B, np.int32)
    x, u = np.zeros((B, N + 1, 12)), np.zeros((B, N, 4))
    lib().quad_ref_solve_batch(C.addressof(params), B, N, x0.ctypes.data, xg.ctypes.data, cost.ctypes.data, iters.ctypes.data,
                               x.ctypes.data, u.ctypes.data, n_threads)
    return dict(cost=cost, iters=iters, x=x, u=u)
