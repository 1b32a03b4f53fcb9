"""Python binding of the BASELINE-label model of libdirect_ddp.so (include/direct_quad.h): batched iLQR for a
12-state / 4-control quadrotor.  NO REFERENCE COUNTERPART (see the header).  No CPU fallback."""
import ctypes as C

import numpy as np

from . import abi, solver

EXPORTS = ("direct_quad_default_params", "direct_quad_create", "direct_quad_destroy", "direct_quad_last_error",
           "direct_quad_solve_batch", "direct_quad_begin", "direct_quad_iterate", "direct_quad_get", "direct_quad_last_kernel_ms",
           "direct_quad_set_stream")
SCALARS = ("cost", "reg", "step", "fp_failed", "bp_failed", "iter", "done", "fwd_passes")
ALGORITHMIC_WORDS_PER_KNOT_ITER = 3 * 12 + 5 * 4 + 2 * 4 * 12   # 152


class Params(C.Structure):
    _fields_ = [("mass", C.c_double), ("gravity", C.c_double), ("inertia", C.c_double * 3), ("dt", C.c_double),
                ("q_pos", C.c_double), ("q_vel", C.c_double), ("q_ang", C.c_double), ("q_rate", C.c_double),
                ("r_thrust", C.c_double), ("r_torque", C.c_double),
                ("qf_pos", C.c_double), ("qf_vel", C.c_double), ("qf_ang", C.c_double), ("qf_rate", C.c_double),
                ("reg_base", C.c_double), ("tol", C.c_double), ("iter_max", C.c_int32), ("fixed_iters", C.c_int32)]


_BOUND = False


def _lib():
    global _BOUND
    L = solver.lib()
    if not _BOUND:
        L.direct_quad_last_error.restype = C.c_char_p
        L.direct_quad_default_params.argtypes = [C.c_void_p]
        L.direct_quad_default_params.restype = None
        L.direct_quad_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.direct_quad_destroy.argtypes = [C.c_void_p]
        L.direct_quad_solve_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 6
        L.direct_quad_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.direct_quad_iterate.argtypes = [C.c_void_p, C.c_int32]
        L.direct_quad_get.argtypes = [C.c_void_p] * 6
        L.direct_quad_last_kernel_ms.argtypes = [C.c_void_p, C.c_void_p]
        L.direct_quad_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        _BOUND = True
    return L


def _check(st):
    if st != abi.DIRECT_OK:
        raise solver.DirectError(st, _lib().direct_quad_last_error().decode())


def default_params(**kw):
    p = Params()
    _lib().direct_quad_default_params(C.addressof(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def label_problems(batch, seed=1000, first=0, reach=5.0):
    """Start / goal states of the label model from config 2's start / goal pairs (direct_amd/problems.py): the same
    start positions; the goal is `reach` metres along the same direction (config 2's goals are 150-350 m away, out of
    reach of a 5 s horizon), at rest and level."""
    from . import problems
    b = problems.make_batch("free", batch, 4, seed=seed, first=first)
    d = b.xd[:, :3] - b.x0[:, :3]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    x0, xg = np.zeros((batch, 12)), np.zeros((batch, 12))
    x0[:, :3] = b.x0[:, :3]
    xg[:, :3] = b.x0[:, :3] + reach * d
    return x0, xg


class QuadSolver:
    def __init__(self, max_batch, n_knots=100, dtype=np.float32, device=0):
        self.np_dtype = np.dtype(dtype)
        self.max_batch, self.N = int(max_batch), int(n_knots)
        h = C.c_void_p()
        _check(_lib().direct_quad_create(abi.F64 if self.np_dtype == np.float64 else abi.F32, device, self.max_batch, self.N,
                                         C.addressof(h)))
        self.h = h
        self.B = 0

    def close(self):
        if getattr(self, "h", None):
            _lib().direct_quad_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solve(self, params, x0, xg):
        x0, xg = np.ascontiguousarray(x0, self.np_dtype), np.ascontiguousarray(xg, self.np_dtype)
        B, N = x0.shape[0], self.N
        cost, iters = np.zeros(B, self.np_dtype), np.zeros(B, np.int32)
        x, u = np.zeros((B, N + 1, 12), self.np_dtype), np.zeros((B, N, 4), self.np_dtype)
        _check(_lib().direct_quad_solve_batch(self.h, C.addressof(params), B, abi.MEM_HOST, x0.ctypes.data, xg.ctypes.data,
                                              cost.ctypes.data, iters.ctypes.data, x.ctypes.data, u.ctypes.data))
        self.B = B
        return dict(cost=cost, iters=iters, x=x, u=u)

    def solve_device(self, params, batch, x0_ptr, xg_ptr, cost_ptr=None, iters_ptr=None):
        _check(_lib().direct_quad_solve_batch(self.h, C.addressof(params), batch, abi.MEM_DEVICE, C.c_void_p(x0_ptr),
                                              C.c_void_p(xg_ptr), C.c_void_p(cost_ptr), C.c_void_p(iters_ptr), None, None))
        self.B = batch

    def begin(self, params, x0, xg):
        x0, xg = np.ascontiguousarray(x0, self.np_dtype), np.ascontiguousarray(xg, self.np_dtype)
        self.B = x0.shape[0]
        _check(_lib().direct_quad_begin(self.h, C.addressof(params), self.B, abi.MEM_HOST, x0.ctypes.data, xg.ctypes.data))

    def set_stream(self, hip_stream):
        _check(_lib().direct_quad_set_stream(self.h, C.c_void_p(hip_stream)))

    def iterate(self, n=1):
        _check(_lib().direct_quad_iterate(self.h, int(n)))

    def get(self):
        B, N = self.B, self.N
        x, u = np.zeros((B, N + 1, 12), self.np_dtype), np.zeros((B, N, 4), self.np_dtype)
        K, kf, sc = np.zeros((B, N, 4, 12), self.np_dtype), np.zeros((B, N, 4), self.np_dtype), np.zeros((B, 8))
        _check(_lib().direct_quad_get(self.h, x.ctypes.data, u.ctypes.data, K.ctypes.data, kf.ctypes.data, sc.ctypes.data))
        return dict(x=x, u=u, K=K, kf=kf, **{n: sc[:, i] for i, n in enumerate(SCALARS)})

    def last_kernel_ms(self):
        ms = C.c_double()
        _check(_lib().direct_quad_last_kernel_ms(self.h, C.addressof(ms)))
        return ms.value
