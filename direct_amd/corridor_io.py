"""Corridor wire format and replay, Python side of the C-ABI.

msgs/corridor of the reference (msgs/msg/corridor.msg, polyhedron.msg, facet3.msg) in ROS 1
serialisation -- what its recorder publishes and its replay reads (writeCorridorMsg / readCorridorMsg,
global_planner/src/teach_repeat_planner.cpp:354-410) -- and the replay protocol of corridorRecCallBack /
fastTrajPlanning (TRP:308-352, 796-823): the problems "first n polytopes", n = n_first .. n_first+B-1,
as ONE batch for direct_ddp_plan_batch.  All byte handling is done by libdirect_ddp.so; a recording
file is a plain concatenation of [uint32 length][message] records.
"""
import ctypes as C
import struct

import numpy as np

from . import abi
from .solver import lib, _check


class Corridor:
    """One recorded corridor: path_id, n_planes[N], planes[N][p_max][4], seeds[N][3], centers[N][3]."""

    def __init__(self, path_id, n_planes, planes, seeds, centers):
        self.path_id = int(path_id)
        self.n_planes = np.ascontiguousarray(n_planes, np.int32)
        self.planes = np.ascontiguousarray(planes, np.float64)
        self.seeds = np.ascontiguousarray(seeds, np.float64)
        self.centers = np.ascontiguousarray(centers, np.float64)
        self.n_seg, self.p_max = int(self.planes.shape[0]), int(self.planes.shape[1])
        assert self.planes.shape == (self.n_seg, self.p_max, 4)
        assert self.seeds.shape == self.centers.shape == (self.n_seg, 3) and self.n_planes.shape == (self.n_seg,)


def _setup():
    L = lib()
    L.direct_corridor_wire_size.restype = C.c_size_t
    L.direct_corridor_wire_size.argtypes = [C.c_int32, C.c_void_p]
    L.direct_corridor_pack.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_size_t, C.c_void_p]
    L.direct_corridor_unpack.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_int32] + [C.c_void_p] * 7
    L.direct_corridor_replay_batch.argtypes = ([C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                                C.c_int32, C.c_int32, C.c_double, C.c_double] + [C.c_void_p] * 7)
    return L


def pack(cor):
    """Corridor -> bytes (one msgs/corridor message)."""
    L = _setup()
    n = L.direct_corridor_wire_size(cor.n_seg, cor.n_planes.ctypes.data)
    buf = (C.c_uint8 * n)()
    written = C.c_size_t()
    _check(L.direct_corridor_pack(cor.path_id, cor.n_seg, cor.n_planes.ctypes.data, cor.planes.ctypes.data, cor.p_max,
                                  cor.seeds.ctypes.data, cor.centers.ctypes.data, buf, n, C.addressof(written)))
    return bytes(buf[:written.value])


def unpack(data, n_seg_max=256, p_max=abi.P_LIMIT, offset=0):
    """bytes -> (Corridor, bytes consumed)."""
    L = _setup()
    raw = (C.c_uint8 * (len(data) - offset)).from_buffer_copy(data[offset:])
    path_id, n_seg, used = C.c_int32(), C.c_int32(), C.c_size_t()
    n_planes = np.zeros(n_seg_max, np.int32)
    planes = np.zeros((n_seg_max, p_max, 4))
    seeds, centers = np.zeros((n_seg_max, 3)), np.zeros((n_seg_max, 3))
    _check(L.direct_corridor_unpack(raw, len(raw), n_seg_max, p_max, C.addressof(path_id), C.addressof(n_seg),
                                    n_planes.ctypes.data, planes.ctypes.data, seeds.ctypes.data, centers.ctypes.data,
                                    C.addressof(used)))
    n = n_seg.value
    return Corridor(path_id.value, n_planes[:n], planes[:n], seeds[:n], centers[:n]), used.value


def write_recording(path, corridors):
    with open(path, "wb") as f:
        for c in corridors:
            msg = pack(c)
            f.write(struct.pack("<I", len(msg)))
            f.write(msg)


def read_recording(path, n_seg_max=256, p_max=abi.P_LIMIT):
    data = open(path, "rb").read()
    out, off = [], 0
    while off < len(data):
        (n,) = struct.unpack_from("<I", data, off)
        cor, used = unpack(data[off + 4:off + 4 + n], n_seg_max, p_max)
        if used != n:
            raise ValueError("record length %d but the message has %d bytes" % (n, used))
        out.append(cor)
        off += 4 + n
    return out


def replay_batch(cor, n_first=2, batch=None, max_vel=2.0, max_acc=2.0, dtype=np.float64):
    """The benchmark replay of corridorRecCallBack (TRP:316-320: poly counts 2..64) as one HostBatch."""
    L = _setup()
    if batch is None:
        batch = cor.n_seg - n_first + 1
    nm = n_first + batch - 1
    n_seg = np.zeros(batch, np.int32)
    x0, xd = np.zeros((batch, 9)), np.zeros((batch, 9))
    T0 = np.zeros((batch, nm))
    n_planes = np.zeros((batch, nm), np.int32)
    planes = np.zeros((batch, nm, cor.p_max, 4))
    seeds = np.zeros((batch, nm, 3))
    _check(L.direct_corridor_replay_batch(cor.n_seg, cor.n_planes.ctypes.data, cor.planes.ctypes.data, cor.p_max,
                                          cor.seeds.ctypes.data, cor.centers.ctypes.data, n_first, batch, max_vel, max_acc,
                                          n_seg.ctypes.data, x0.ctypes.data, xd.ctypes.data, T0.ctypes.data,
                                          n_planes.ctypes.data, planes.ctypes.data, seeds.ctypes.data))
    return abi.HostBatch(n_seg, x0, xd, T0, n_planes, planes, seeds=seeds, dtype=dtype)
