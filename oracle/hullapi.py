"""ctypes front-end of oracle/libhull_ref.so (restatement of the hull -> H-rep step, hull_ref.c) and of
oracle/_ref/libquickhull_ref.so (the reference's own third_party/quickhull, built from its sources) -- TEST
INFRASTRUCTURE.  Only tests/ and __graft_entry__.smoke() may import this module; nothing under direct_amd/ does."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(_HERE, "_ref", "libquickhull_ref.so")
_LIB = None
_REF = None


def build(force=False):
    so, src = os.path.join(_HERE, "libhull_ref.so"), os.path.join(_HERE, "hull_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libhull_ref.so"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/global_planner/third_party/quickhull/QuickHull.cpp") and (force or not os.path.exists(REF_SO)):
        subprocess.check_call(["make", "-C", _HERE, "_ref"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.hull_ref.argtypes = [C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def ref_lib():
    """The reference's quickhull (None when oracle/_ref has not been built)."""
    global _REF
    if _REF is None and os.path.exists(REF_SO):
        R = C.CDLL(REF_SO)
        R.ref_quickhull.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _REF = R
    return _REF


def hull_planes(cluster, res, lower, plane_cap=512, vert_cap=2048):
    """-> dict(rc, degenerate, plane_int [P][4], planes [P][4], vert_q [V][3], vertices [V][3], center [3])"""
    idx = np.ascontiguousarray(cluster, np.int32).reshape(-1, 3)
    lower = np.ascontiguousarray(lower, np.float64)
    pi, pd = np.zeros((plane_cap, 4), np.int64), np.zeros((plane_cap, 4), np.float64)
    vq, vd = np.zeros((vert_cap, 3), np.int32), np.zeros((vert_cap, 3), np.float64)
    npl, nv, deg = C.c_int(), C.c_int(), C.c_int()
    ctr = np.zeros(3, np.float64)
    rc = lib().hull_ref(len(idx), idx.ctypes.data, float(res), lower.ctypes.data, plane_cap, pi.ctypes.data, pd.ctypes.data,
                        C.addressof(npl), vert_cap, vq.ctypes.data, vd.ctypes.data, C.addressof(nv), ctr.ctypes.data,
                        C.addressof(deg))
    P, V = min(npl.value, plane_cap), min(nv.value, vert_cap)
    return dict(rc=rc, degenerate=deg.value, n_planes=npl.value, n_vertices=nv.value, plane_int=pi[:P].copy(),
                planes=pd[:P].copy(), vert_q=vq[:V].copy(), vertices=vd[:V].copy(), center=ctr)


def lattice_points(cluster, degenerate):
    """the point set getConvexPoly hands to quickhull, on the half-voxel lattice (q = 2 index + 1 [+/- 1])"""
    idx = np.asarray(cluster, np.int64).reshape(-1, 3)
    if not degenerate:
        return 2 * idx + 1
    corners = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.int64)
    return ((2 * idx + 1)[:, None, :] + corners[None]).reshape(-1, 3)


def reference_quickhull(points):
    """the reference's getConvexHull(points, true, false): (vertex buffer [V][3], triangles [T][3] of vertex indices)"""
    R = ref_lib()
    if R is None:
        raise RuntimeError("oracle/_ref/libquickhull_ref.so is missing")
    pts = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    vcap, icap = len(pts), 12 * len(pts) + 64
    vb, ib = np.zeros((vcap, 3), np.float64), np.zeros(icap, np.int32)
    nv, ni = C.c_int(), C.c_int()
    rc = R.ref_quickhull(len(pts), pts.ctypes.data, vcap, vb.ctypes.data, C.addressof(nv), icap, ib.ctypes.data, C.addressof(ni))
    assert rc == 0, (rc, nv.value, ni.value)
    return vb[:nv.value].copy(), ib[:ni.value].reshape(-1, 3).copy()


def primitive_plane(a, b, c):
    """primitive integer plane (n, K) through three lattice points, sign undetermined; None when collinear"""
    a, b, c = (np.asarray(v, np.int64) for v in (a, b, c))
    n = np.cross(b - a, c - a)
    g = np.gcd.reduce(np.abs(n))
    if g == 0:
        return None
    n = n // g
    return np.array([n[0], n[1], n[2], -int(n @ a)], np.int64)


def check_mesh(vbq, tri, planes_int, vert_q):
    """The pin, given quickhull's output (vertex buffer on the lattice, triangles): its triangles lie in the facet
    planes, every facet plane carries a triangle, and its vertex buffer is the corner set plus, possibly, points that
    lie on the boundary without being corners (its initial tetrahedron is built from axis extremes, which need not be
    corners).  Raises AssertionError with the reason."""
    vbq = np.asarray(vbq, np.int64)
    keys = {tuple(int(x) for x in p) for p in planes_int}
    hit = set()
    for t in tri:
        pl = primitive_plane(vbq[t[0]], vbq[t[1]], vbq[t[2]])
        if pl is None:
            continue  # a sliver of three collinear boundary points
        k = tuple(int(x) for x in pl)
        if k not in keys:
            k = tuple(-x for x in k)
        assert k in keys, ("quickhull triangle in no facet plane", pl)
        hit.add(k)
    assert hit == keys, ("facet planes without a quickhull triangle", keys - hit)
    vs = {tuple(int(x) for x in v) for v in vert_q}
    qs = {tuple(int(x) for x in v) for v in vbq}
    assert vs <= qs, ("corners missing from quickhull's vertex buffer", vs - qs)
    P = np.asarray(planes_int, np.int64)
    for v in qs - vs:
        assert (P[:, :3] @ np.array(v) + P[:, 3] == 0).any(), ("quickhull vertex not on the boundary", v)


def check_against_quickhull(points_q, planes_int, vert_q):
    """check_mesh against the reference's quickhull run here on `points_q` -> (vertices, triangles) it returned"""
    vb, tri = reference_quickhull(np.asarray(points_q, np.float64))
    vbq = np.rint(vb).astype(np.int64)
    assert np.abs(vb - vbq).max() == 0
    check_mesh(vbq, tri, planes_int, vert_q)
    return len(vb), len(tri)
