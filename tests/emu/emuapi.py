"""ctypes front-end of the TEST-ONLY lane-loop emulator (tests/emu/emu.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

from direct_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libddp_emu.so")
    srcs = [os.path.join(_HERE, "emu.cpp"), os.path.join(_HERE, "../../direct_amd/csrc/ddp_wave.h"),
            os.path.join(_HERE, "../../direct_amd/csrc/ddp_tables.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(s) for s in srcs):
        tmp = "%s.tmp.%d" % (so, os.getpid())   # built aside and renamed: parallel test workers never load a half-written library
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-w",
                               "-o", tmp, srcs[0]])
        os.replace(tmp, so)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.emu_begin.restype = C.c_void_p
        L.emu_begin.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        for n in ("emu_backward", "emu_forward", "emu_forward_stored", "emu_end"):
            getattr(L, n).argtypes = [C.c_void_p]
            getattr(L, n).restype = None
        L.emu_iterate.argtypes = [C.c_void_p, C.c_int]
        L.emu_get_field.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.emu_set_field.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.emu_finish.argtypes = [C.c_void_p, C.c_void_p]
        L.emu_split_knots.argtypes = [C.c_void_p]
        L.emu_split_knots.restype = C.c_int
        _LIB = L
    return _LIB


class EmuSolver:
    def __init__(self, params, batch, dtype=np.float64, compute64=False):
        self.np_dtype = np.dtype(dtype)
        self.batch = batch if batch.dtype == self.np_dtype else batch.astype(self.np_dtype)
        self._cin = self.batch.c_struct()
        self.params = params
        code = abi.F64 if self.np_dtype == np.float64 else (2 if compute64 else abi.F32)
        self.h = lib().emu_begin(code, C.addressof(params), C.addressof(self._cin))

    def close(self):
        if self.h:
            lib().emu_end(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def backward(self):
        lib().emu_backward(self.h)

    def forward(self):
        lib().emu_forward(self.h)

    def forward_stored(self):
        lib().emu_forward_stored(self.h)

    def iterate(self, n):
        lib().emu_iterate(self.h, n)

    def field_shape(self, field):
        b = self.batch
        B, nm, ncm = b.batch, b.n_seg_max, b.nc_max
        return {abi.FIELD_X: (B, nm + 1, 9), abi.FIELD_U: (B, nm, 10), abi.FIELD_S: (B, nm, ncm),
                abi.FIELD_Y: (B, nm, ncm), abi.FIELD_C: (B, nm, ncm), abi.FIELD_KU: (B, nm, 10),
                abi.FIELD_KUU: (B, nm, 10, 9), abi.FIELD_KS: (B, nm, ncm), abi.FIELD_KY: (B, nm, ncm),
                abi.FIELD_SCALARS: (B, 16)}[field]

    def get(self, field):
        out = np.zeros(self.field_shape(field), self.np_dtype)
        lib().emu_get_field(self.h, field, out.ctypes.data)
        return out

    def set(self, field, arr):
        a = np.ascontiguousarray(arr, self.np_dtype)
        assert a.shape == self.field_shape(field)
        lib().emu_set_field(self.h, field, a.ctypes.data)

    def scalars(self):
        s = self.get(abi.FIELD_SCALARS)
        return {n: s[:, i] for i, n in enumerate(abi.SCALAR_NAMES)}

    def split_knots(self):
        """knots whose front half went through a hand-over record (DIRECT_EMU_BSPLIT=1 at construction; else 0)"""
        return int(lib().emu_split_knots(self.h))

    def finish(self):
        res = abi.HostResult(self.batch.batch, self.batch.n_seg_max, self.np_dtype)
        cout = res.c_struct()
        lib().emu_finish(self.h, C.addressof(cout))
        return res


def solve_batch(params, batch, dtype=np.float64, compute64=False):
    s = EmuSolver(params, batch, dtype, compute64)
    s.iterate(params.iter_max)
    res = s.finish()
    s.close()
    return res
