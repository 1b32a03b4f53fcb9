"""ctypes front-end of the TEST-ONLY host build of the hull phases (tests/emu/hull_emu.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libhull_emu.so")
        srcs = [os.path.join(_HERE, "hull_emu.cpp"), os.path.join(_HERE, "../../direct_amd/csrc/hull_core.h")]
        if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(s) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-w", "-o", so, srcs[0]])
        L = C.CDLL(so)
        L.hull_emu.argtypes = [C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_int] + [C.c_void_p] * 6
        _LIB = L
    return _LIB


def hull_planes(cluster, res, lower, plane_cap=512, vert_cap=2048):
    """same result dict as oracle.hullapi.hull_planes, plus n_cand (line-extreme points the pair phase works on)"""
    idx = np.ascontiguousarray(cluster, np.int32).reshape(-1, 3)
    lower = np.ascontiguousarray(lower, np.float64)
    pi, pd = np.zeros((plane_cap, 4), np.int64), np.zeros((plane_cap, 4), np.float64)
    vq, vd, ctr = np.zeros((vert_cap, 3), np.int32), np.zeros((vert_cap, 3), np.float64), np.zeros(3, np.float64)
    npl, nv, deg, nc = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = lib().hull_emu(len(idx), idx.ctypes.data, float(res), lower.ctypes.data, plane_cap, pi.ctypes.data, pd.ctypes.data,
                        C.addressof(npl), vert_cap, vq.ctypes.data, vd.ctypes.data, C.addressof(nv), ctr.ctypes.data,
                        C.addressof(deg), C.addressof(nc))
    P, V = min(npl.value, plane_cap), min(nv.value, vert_cap)
    return dict(rc=rc, degenerate=deg.value, n_planes=npl.value, n_vertices=nv.value, plane_int=pi[:P].copy(),
                planes=pd[:P].copy(), vert_q=vq[:V].copy(), vertices=vd[:V].copy(), center=ctr, n_cand=nc.value)
