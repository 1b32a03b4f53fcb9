"""ctypes front-end of oracle/libcluster_ref.so (restatement) and oracle/_ref/libcluster_engine_ref.so (the
reference's own serialConvexTest, built from its source) -- TEST INFRASTRUCTURE.  Only tests/ and
__graft_entry__.smoke() may import this module; nothing under direct_amd/ does."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(_HERE, "_ref", "libcluster_engine_ref.so")
_LIB = None
_REF = None
CONVEX_FN = C.CFUNCTYPE(C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)


def build(force=False):
    so = os.path.join(_HERE, "libcluster_ref.so")
    src = os.path.join(_HERE, "cluster_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libcluster_ref.so"], stdout=subprocess.DEVNULL)
    # the reference build: only where /root/reference exists (this container); the GPU box uses the prebuilt file
    if os.path.exists("/root/reference/polyhedron_generator/src/cluster_engine_cpu.cpp") and (force or not os.path.exists(REF_SO)):
        subprocess.check_call(["make", "-C", _HERE, "_ref"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.cl_serial_convex_test.argtypes = [C.c_int] * 6 + [C.c_void_p] * 3
        L.cl_polygon_generation.argtypes = [C.c_void_p] + [C.c_int] * 9 + [C.c_void_p] * 4
        L.cl_cluster_round.argtypes = [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p] * 5
        L.cl_set_convex_test.argtypes = [C.c_void_p]
        L.cl_candidates.argtypes = [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p, C.c_int, C.c_void_p]
        _LIB = L
    return _LIB


def ref_lib():
    """The reference's own serialConvexTest (None when oracle/_ref has not been built)."""
    global _REF
    if _REF is None and os.path.exists(REF_SO):
        R = C.CDLL(REF_SO)
        R.ref_serial_convex_test.argtypes = [C.c_int] * 6 + [C.c_void_p] * 3
        _REF = R
    return _REF


def use_reference_convex_test(on=True):
    """Route the restated clustering loop through the reference's serialConvexTest (pinned innermost function)."""
    r = ref_lib()
    if on and r is None:
        raise RuntimeError("oracle/_ref/libcluster_engine_ref.so is missing")
    lib().cl_set_convex_test(C.cast(r.ref_serial_convex_test, C.c_void_p) if on else None)


def _i32(a):
    return np.ascontiguousarray(a, np.int32)


def serial_convex_test(fn_lib, name, cand, cluster, inside, grid, dims):
    """serialConvexTest for every candidate voxel `cand[n][3]` against `cluster[m][3]` -> uint8[n]."""
    cand, cluster = _i32(cand), _i32(cluster)
    inside, grid = np.ascontiguousarray(inside, np.uint8), np.ascontiguousarray(grid, np.uint8)
    f = getattr(fn_lib, name)
    out = np.zeros(len(cand), np.uint8)
    for i, (x, y, z) in enumerate(cand):
        out[i] = f(int(x), int(y), int(z), len(cluster), dims[1] * dims[2], dims[2], cluster.ctypes.data,
                   inside.ctypes.data, grid.ctypes.data)
    return out


def polygon_generation(grid, seed, itr_inflate_max=1000, itr_cluster_max=50, cap=200000):
    """polygonGeneration for one seed voxel -> (vertex_idx[24], cluster[n][3], cluster iterations, rc)."""
    grid = np.ascontiguousarray(grid, np.uint8)
    dims = grid.shape
    v = np.zeros(24, np.int32)
    cl = np.zeros((cap, 3), np.int32)
    n, it = C.c_int(), C.c_int()
    rc = lib().cl_polygon_generation(grid.ctypes.data, dims[0], dims[1], dims[2], int(seed[0]), int(seed[1]), int(seed[2]),
                                     int(itr_inflate_max), int(itr_cluster_max), cap, v.ctypes.data, cl.ctypes.data,
                                     C.addressof(n), C.addressof(it))
    return v, cl[:n.value].copy(), it.value, rc


def cube_state(grid, v):
    """use / inside flag grids right before the clustering loop (cluster_server_cpu.cpp:425-506): use_data = 1 on the
    inflated cube, inside_data = 1 on its strict interior.  (A one-voxel cube sets neither.)"""
    use, inside = np.zeros(grid.shape, np.uint8), np.zeros(grid.shape, np.uint8)
    x0, x1, y0, y1, z0, z1 = v[7], v[1], v[15], v[9], v[23], v[17]
    if (x1 - x0 + 1) * (y1 - y0 + 1) * (z1 - z0 + 1) > 1:
        use[x0:x1 + 1, y0:y1 + 1, z0:z1 + 1] = 1
        inside[x0 + 1:x1, y0 + 1:y1, z0 + 1:z1] = 1
    return use, inside


def candidates(grid, use, invalid, inside, active):
    """One round's candidate list (cluster_server_cpu.cpp:301-353); marks them in `use` like the reference."""
    grid = np.ascontiguousarray(grid, np.uint8)
    active = _i32(active)
    cand = np.zeros((26 * len(active) + 1, 3), np.int32)
    n = lib().cl_candidates(grid.ctypes.data, use.ctypes.data, invalid.ctypes.data, inside.ctypes.data, grid.shape[0],
                            grid.shape[1], grid.shape[2], active.ctypes.data, len(active), cand.ctypes.data)
    return cand[:n].copy()


def accept_sequential(can_clu, can_can):
    """The accept decisions of polytopeCluster_cpu's loop (cluster_server_cpu.cpp:360-384) from per-pair results:
    candidate i joins iff it sees the whole old cluster and every earlier candidate that has joined."""
    n = len(can_clu)
    acc = np.zeros(n, np.uint8)
    for i in range(n):
        ok = bool(can_clu[i])
        if ok:
            row = can_can[i * (i - 1) // 2:i * (i - 1) // 2 + i]
            ok = not np.any((row == 0) & (acc[:i] == 1))
        acc[i] = ok
    return acc
