"""Python binding of the corridor-cluster generator of libdirect_ddp.so (include/direct_cluster.h) for tests and
tools.  Mirrors cudaPolytopeGeneration's calling protocol (polyhedron_generator/include/polyhedron_generator/
cluster_server_cpu.h:54-74): paramSet -> ClusterGenerator(...), setObs/mapUpload -> set_map, polygonGeneration ->
polygon_generation (for a batch of seed voxels).  No CPU fallback."""
import ctypes as C

import numpy as np

from . import abi, solver


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("max_x", C.c_int32), ("max_y", C.c_int32), ("max_z", C.c_int32),
                ("max_batch", C.c_int32), ("cluster_capacity", C.c_int32), ("candidate_capacity", C.c_int32),
                ("reserved", C.c_int32)]


EXPORTS = ("direct_cluster_create", "direct_cluster_destroy", "direct_cluster_last_error", "direct_cluster_set_map",
           "direct_cluster_polygon_generation_batch", "direct_cluster_convex_test", "direct_cluster_last_ms",
           "direct_cluster_set_stream", "direct_cluster_hull_planes_batch")
CLUSTER_OK, CLUSTER_OVERFLOW, CLUSTER_BAD_SEED = 0, 1, 2
HULL_OK, HULL_OVERFLOW, HULL_BAD_VOXEL, HULL_FLAT = 0, 1, 2, 3
_BOUND = False


def _lib():
    global _BOUND
    L = solver.lib()
    if not _BOUND:
        L.direct_cluster_last_error.restype = C.c_char_p
        L.direct_cluster_create.argtypes = [C.c_void_p, C.c_void_p]
        L.direct_cluster_destroy.argtypes = [C.c_void_p]
        L.direct_cluster_set_map.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.direct_cluster_polygon_generation_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                                               C.c_int32] + [C.c_void_p] * 5
        L.direct_cluster_convex_test.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]
        L.direct_cluster_last_ms.argtypes = [C.c_void_p, C.c_void_p]
        L.direct_cluster_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.direct_cluster_hull_planes_batch.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_double,
                                                        C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 8
        _BOUND = True
    return L


def _check(st):
    if st != abi.DIRECT_OK:
        raise solver.DirectError(st, _lib().direct_cluster_last_error().decode())


class ClusterGenerator:
    def __init__(self, dims, max_batch=64, cluster_capacity=50000, candidate_capacity=10000, device=0):
        self.dims = tuple(int(d) for d in dims)
        self.max_batch, self.ccap, self.kcap = int(max_batch), int(cluster_capacity), int(candidate_capacity)
        cfg = Config(device, self.dims[0], self.dims[1], self.dims[2], self.max_batch, self.ccap, self.kcap, 0)
        h = C.c_void_p()
        _check(_lib().direct_cluster_create(C.addressof(cfg), C.addressof(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            _lib().direct_cluster_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_map(self, grid):
        g = np.ascontiguousarray(grid, np.uint8)
        assert g.shape == self.dims
        _check(_lib().direct_cluster_set_map(self.h, abi.MEM_HOST, g.ctypes.data))

    def polygon_generation(self, seeds, itr_inflate_max=1000, itr_cluster_max=50, fetch_clusters=True):
        """-> dict(vertex_idx [B][24], clusters: list of [n][3] arrays, cluster_num, iters, rtn).  fetch_clusters=False:
        the voxels stay on the device (for hull_planes(batch=...)), clusters is None."""
        seeds = np.ascontiguousarray(seeds, np.int32).reshape(-1, 3)
        B = seeds.shape[0]
        if not fetch_clusters:
            v = np.zeros((B, 24), np.int32)
            n, it, rtn = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32)
            _check(_lib().direct_cluster_polygon_generation_batch(self.h, B, seeds.ctypes.data, int(itr_inflate_max),
                                                                  int(itr_cluster_max), abi.MEM_HOST, v.ctypes.data, None,
                                                                  n.ctypes.data, it.ctypes.data, rtn.ctypes.data))
            return dict(vertex_idx=v, clusters=None, cluster_num=n, iters=it, rtn=rtn)
        v = np.zeros((B, 24), np.int32)
        cl = np.zeros((B, self.ccap, 3), np.int32)
        n, it, rtn = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32)
        _check(_lib().direct_cluster_polygon_generation_batch(self.h, B, seeds.ctypes.data, int(itr_inflate_max),
                                                              int(itr_cluster_max), abi.MEM_HOST, v.ctypes.data,
                                                              cl.ctypes.data, n.ctypes.data, it.ctypes.data, rtn.ctypes.data))
        return dict(vertex_idx=v, clusters=[cl[b, :n[b]].copy() for b in range(B)], cluster_num=n, iters=it, rtn=rtn)

    def convex_test(self, inside, cand, cluster):
        """-> (can_clu [n], can_can packed lower triangle [n(n-1)/2], accept [n]), uint8 each"""
        inside = np.ascontiguousarray(inside, np.uint8)
        cand = np.ascontiguousarray(cand, np.int32).reshape(-1, 3)
        cluster = np.ascontiguousarray(cluster, np.int32).reshape(-1, 3)
        n = cand.shape[0]
        clu, acc = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        cc = np.zeros(max(n * (n - 1) // 2, 1), np.uint8)
        _check(_lib().direct_cluster_convex_test(self.h, inside.ctypes.data, n, cand.ctypes.data, cluster.shape[0],
                                                 cluster.ctypes.data, clu.ctypes.data, cc.ctypes.data, acc.ctypes.data))
        return clu, cc[:n * (n - 1) // 2], acc

    def hull_planes(self, resolution, map_lower, clusters=None, plane_capacity=256, vertex_capacity=1024, batch=None):
        """getConvexPoly's hull + Polyhedron::hrep + polyHrep2Utils (poly_utils.cpp:301-389, 127-206) for a batch of
        clusters.  clusters=None: those of the last polygon_generation, still on the device (`batch` of them);
        otherwise a list of [n][3] voxel-index arrays.  -> dict(planes: list of [P][4], plane_int, vertices: list of
        [V][3], center [B][3], degenerate, n_planes, n_vertices, rtn)"""
        lower = np.ascontiguousarray(map_lower, np.float64)
        if clusters is not None:
            B = len(clusters)
            xyz = np.zeros((B, self.ccap, 3), np.int32)
            num = np.zeros(B, np.int32)
            for b, c in enumerate(clusters):
                c = np.ascontiguousarray(c, np.int32).reshape(-1, 3)
                assert len(c) <= self.ccap
                xyz[b, :len(c)] = c
                num[b] = len(c)
            px, pn = xyz.ctypes.data, num.ctypes.data
        else:
            B, px, pn = int(batch), None, None
        pl = np.zeros((B, plane_capacity, 4), np.float64)
        pi = np.zeros((B, plane_capacity, 4), np.int64)
        vt = np.zeros((B, vertex_capacity, 3), np.float64)
        ctr = np.zeros((B, 3), np.float64)
        npl, nv, deg, rtn = (np.zeros(B, np.int32) for _ in range(4))
        _check(_lib().direct_cluster_hull_planes_batch(self.h, B, abi.MEM_HOST, px, pn, float(resolution), lower.ctypes.data,
                                                       int(plane_capacity), int(vertex_capacity), abi.MEM_HOST, pl.ctypes.data,
                                                       pi.ctypes.data, npl.ctypes.data, vt.ctypes.data, nv.ctypes.data,
                                                       ctr.ctypes.data, deg.ctypes.data, rtn.ctypes.data))
        cut = lambda a, n, cap: [a[b, :min(int(n[b]), cap)].copy() for b in range(B)]
        return dict(planes=cut(pl, npl, plane_capacity), plane_int=cut(pi, npl, plane_capacity),
                    vertices=cut(vt, nv, vertex_capacity), center=ctr, degenerate=deg, n_planes=npl, n_vertices=nv, rtn=rtn)

    def set_stream(self, hip_stream):
        _check(_lib().direct_cluster_set_stream(self.h, C.c_void_p(hip_stream)))

    def last_ms(self):
        ms = C.c_float()
        _check(_lib().direct_cluster_last_ms(self.h, C.addressof(ms)))
        return ms.value
