"""Python binding of libdirect_ddp.so (the C-ABI of include/direct_ddp.h) for tests and bench.py.

The product is the shared library; this module only loads it with ctypes and mirrors the calling
protocol of the reference: `DdpSolver.solve` = ddpTrajOptimizer::polyCurveGeneration + getters
(global_planner/include/global_planner/ddp_optimizer.h:267-340), `DdpSolver.plan` =
fastTrajPlanning's two-phase chaining (global_planner/src/teach_repeat_planner.cpp:886-921).
There is no CPU fallback: loading fails loudly when the library is missing and creating a solver
fails when no gfx950 device is visible.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DIRECT_DDP_LIB", os.path.join(_HERE, "lib", "libdirect_ddp.so"))
_LIB = None

EXPORTS = (
    "direct_ddp_abi_version", "direct_ddp_last_error", "direct_ddp_create", "direct_ddp_destroy",
    "direct_ddp_set_stream", "direct_ddp_solve_batch", "direct_ddp_plan_batch", "direct_time_allocation",
    "direct_ddp_begin", "direct_ddp_backward_pass", "direct_ddp_forward_pass", "direct_ddp_forward_pass_stored", "direct_ddp_iterate",
    "direct_ddp_finish", "direct_ddp_get_field", "direct_ddp_set_field", "direct_ddp_last_kernel_ms",
    "direct_ddp_best_cost", "direct_ddp_sched_error", "direct_ddp_sched_debug", "direct_traj_sample_batch", "direct_traj_sample_last_ms",
    "direct_rccl_unique_id", "direct_rccl_comm_create", "direct_rccl_comm_destroy", "direct_ddp_gather_best",
    "direct_corridor_wire_size", "direct_corridor_pack", "direct_corridor_unpack", "direct_corridor_replay_batch",
    "direct_ddp_last_launch_info", "direct_ddp_last_counters",
)


class DirectError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("direct_ddp status %d: %s" % (status, msg))
        self.status = status


def lib():
    """Load libdirect_ddp.so (built by __graft_entry__.build()).  Raises if it is missing."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.direct_ddp_abi_version.restype = C.c_int32
        L.direct_ddp_last_error.restype = C.c_char_p
        L.direct_ddp_create.argtypes = [C.c_void_p, C.c_void_p]
        L.direct_ddp_destroy.argtypes = [C.c_void_p]
        L.direct_ddp_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.direct_ddp_solve_batch.argtypes = [C.c_void_p] * 4
        L.direct_ddp_plan_batch.argtypes = [C.c_void_p] * 6
        L.direct_ddp_begin.argtypes = [C.c_void_p] * 3
        L.direct_ddp_backward_pass.argtypes = [C.c_void_p]
        L.direct_ddp_forward_pass.argtypes = [C.c_void_p]
        if hasattr(L, "direct_ddp_forward_pass_stored"):  # absent from libraries of earlier rounds (tools/ab_libs.sh A/B runs)
            L.direct_ddp_forward_pass_stored.argtypes = [C.c_void_p]
        L.direct_ddp_iterate.argtypes = [C.c_void_p, C.c_int32]
        L.direct_ddp_finish.argtypes = [C.c_void_p, C.c_void_p]
        L.direct_ddp_get_field.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.direct_ddp_set_field.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.direct_ddp_last_kernel_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.direct_ddp_best_cost.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                           C.c_void_p, C.c_void_p]
        L.direct_ddp_sched_error.argtypes = [C.c_void_p, C.c_void_p]
        if hasattr(L, "direct_ddp_sched_debug"):
            L.direct_ddp_sched_debug.argtypes = [C.c_void_p, C.c_void_p]
        if hasattr(L, "direct_ddp_last_launch_info"):  # absent from libraries of earlier rounds (tools/ab_libs.sh A/B runs)
            L.direct_ddp_last_launch_info.argtypes = [C.c_void_p, C.c_void_p]
        if hasattr(L, "direct_ddp_last_counters"):
            L.direct_ddp_last_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.direct_rccl_unique_id.argtypes = [C.c_void_p]
        L.direct_rccl_comm_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.direct_rccl_comm_destroy.argtypes = [C.c_void_p]
        L.direct_ddp_gather_best.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p]
        L.direct_traj_sample_batch.argtypes = [C.c_void_p] * 3
        L.direct_traj_sample_last_ms.argtypes = [C.c_void_p, C.c_void_p]
        L.direct_time_allocation.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        _LIB = L
    return _LIB


def _check(st):
    if st != abi.DIRECT_OK:
        raise DirectError(st, lib().direct_ddp_last_error().decode())


def time_allocation(n_seg, start, goal, seeds, max_vel=2.0, max_acc=2.0):
    """initTimeAllocation (teach_repeat_planner.cpp:583-639) through the C-ABI."""
    n_seg = np.ascontiguousarray(n_seg, np.int32)
    seeds = np.ascontiguousarray(seeds, np.float64)
    start = np.ascontiguousarray(start, np.float64)
    goal = np.ascontiguousarray(goal, np.float64)
    T = np.zeros(seeds.shape[:2])
    _check(lib().direct_time_allocation(n_seg.shape[0], seeds.shape[1], n_seg.ctypes.data, start.ctypes.data,
                                        goal.ctypes.data, seeds.ctypes.data, max_vel, max_acc, T.ctypes.data))
    return T


class DdpSolver:
    """One handle = one GPU; reusable across calls (the reference object is single-use, quirk Q9)."""

    def __init__(self, max_batch, n_seg_max, p_max, dtype=np.float32, device=0, flags=0):
        self.np_dtype = np.dtype(dtype)
        self.dtype = abi.F64 if self.np_dtype == np.float64 else abi.F32
        self.max_batch, self.n_seg_max, self.p_max = int(max_batch), int(n_seg_max), int(p_max)
        cfg = abi.Config(self.dtype, device, self.max_batch, self.n_seg_max, self.p_max, int(flags))  # DIRECT_FLAG_* of include/direct_ddp.h
        h = C.c_void_p()
        _check(lib().direct_ddp_create(C.addressof(cfg), C.addressof(h)))
        self.h = h
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            lib().direct_ddp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        _check(lib().direct_ddp_set_stream(self.h, C.c_void_p(stream_ptr)))

    def _host_batch(self, batch):
        return batch if batch.dtype == self.np_dtype else batch.astype(self.np_dtype)

    # -- host-memory interface ------------------------------------------------------------------
    def solve(self, params, batch):
        """polyCurveGeneration for every corridor of `batch` (abi.HostBatch) -> abi.HostResult."""
        hb = self._host_batch(batch)
        res = abi.HostResult(hb.batch, hb.n_seg_max, self.np_dtype)
        cin, cout = hb.c_struct(), res.c_struct()
        _check(lib().direct_ddp_solve_batch(self.h, C.addressof(params), C.addressof(cin), C.addressof(cout)))
        return res

    def plan(self, params0, params1, batch):
        """Phase 0 -> UpdateTime -> phase 1 (teach_repeat_planner.cpp:886-921) -> (res0, res1)."""
        hb = self._host_batch(batch)
        r0 = abi.HostResult(hb.batch, hb.n_seg_max, self.np_dtype)
        r1 = abi.HostResult(hb.batch, hb.n_seg_max, self.np_dtype)
        cin, c0, c1 = hb.c_struct(), r0.c_struct(), r1.c_struct()
        _check(lib().direct_ddp_plan_batch(self.h, C.addressof(params0), C.addressof(params1), C.addressof(cin),
                                           C.addressof(c0), C.addressof(c1)))
        return r0, r1

    # -- stepwise interface (per-pass parity) -----------------------------------------------------
    def begin(self, params, batch):
        hb = self._host_batch(batch)
        cin = hb.c_struct()
        self._keep = (hb, cin, params)
        self._shape_src = hb
        _check(lib().direct_ddp_begin(self.h, C.addressof(params), C.addressof(cin)))

    def backward(self):
        _check(lib().direct_ddp_backward_pass(self.h))

    def forward(self):
        _check(lib().direct_ddp_forward_pass(self.h))

    def forward_stored(self):
        """forwardpass() in the stored-gain form (what the solver runs after its backward pass got stuck, rtn = -4)"""
        _check(lib().direct_ddp_forward_pass_stored(self.h))

    def iterate(self, n):
        _check(lib().direct_ddp_iterate(self.h, int(n)))

    def finish(self):
        hb = self._shape_src
        res = abi.HostResult(hb.batch, hb.n_seg_max, self.np_dtype)
        cout = res.c_struct()
        _check(lib().direct_ddp_finish(self.h, C.addressof(cout)))
        return res

    def field_shape(self, field):
        b = self._shape_src
        B, nm, ncm = b.batch, b.n_seg_max, b.nc_max
        return {abi.FIELD_X: (B, nm + 1, 9), abi.FIELD_U: (B, nm, 10), abi.FIELD_S: (B, nm, ncm),
                abi.FIELD_Y: (B, nm, ncm), abi.FIELD_C: (B, nm, ncm), abi.FIELD_KU: (B, nm, 10),
                abi.FIELD_KUU: (B, nm, 10, 9), abi.FIELD_KS: (B, nm, ncm), abi.FIELD_KY: (B, nm, ncm),
                abi.FIELD_SCALARS: (B, 16)}[field]

    def get(self, field):
        out = np.zeros(self.field_shape(field), self.np_dtype)
        _check(lib().direct_ddp_get_field(self.h, field, out.ctypes.data))
        return out

    def set(self, field, arr):
        a = np.ascontiguousarray(arr, self.np_dtype)
        assert a.shape == self.field_shape(field)
        _check(lib().direct_ddp_set_field(self.h, field, a.ctypes.data))

    def scalars(self):
        s = self.get(abi.FIELD_SCALARS)
        return {n: s[:, i] for i, n in enumerate(abi.SCALAR_NAMES)}

    def last_kernel_ms(self):
        ms, n = C.c_double(), C.c_int32()
        _check(lib().direct_ddp_last_kernel_ms(self.h, C.addressof(ms), C.addressof(n)))
        return ms.value, n.value

    # -- device-memory interface (pointers are raw device addresses, e.g. torch .data_ptr()) ------
    def solve_device(self, params, cin, cout):
        """cin: abi.BatchIn, cout: abi.BatchOut, both with mem = MEM_DEVICE.  Asynchronous."""
        _check(lib().direct_ddp_solve_batch(self.h, C.addressof(params), C.addressof(cin), C.addressof(cout)))

    def plan_device(self, params0, params1, cin, cout0, cout1):
        _check(lib().direct_ddp_plan_batch(self.h, C.addressof(params0), C.addressof(params1), C.addressof(cin),
                                           None if cout0 is None else C.addressof(cout0), C.addressof(cout1)))

    def sample(self, n_seg, bez, T, dt, capacity, derivs=2, n_planes=None, planes=None):
        """Batched output sampling (direct_traj_sample_batch; teach_repeat_planner.cpp:1551-1566 over
        utils/bezier_base.h:77-127).  Host numpy arrays in, dict of numpy arrays out."""
        n_seg = np.ascontiguousarray(n_seg, np.int32)
        bez = np.ascontiguousarray(bez, self.np_dtype)
        T = np.ascontiguousarray(T, self.np_dtype)
        B, nm = T.shape
        assert bez.shape == (B, nm, 18)
        o = dict(count=np.zeros(B, np.int32), seg_first=np.zeros((B, nm), np.int32),
                 pos=np.zeros((B, capacity, 3), self.np_dtype), length=np.zeros(B, self.np_dtype))
        if derivs >= 1:
            o["vel"] = np.zeros((B, capacity, 3), self.np_dtype)
            o["vmax"] = np.zeros(B, self.np_dtype)
        if derivs >= 2:
            o["acc"] = np.zeros((B, capacity, 3), self.np_dtype)
            o["amax"] = np.zeros(B, self.np_dtype)
        cin, cout = abi.SampleIn(), abi.SampleOut()
        cin.batch, cin.n_seg_max, cin.capacity, cin.derivs, cin.mem = B, nm, capacity, derivs, abi.MEM_HOST
        cin.n_seg, cin.bez, cin.T, cin.dt = n_seg.ctypes.data, bez.ctypes.data, T.ctypes.data, float(dt)
        if planes is not None:  # containment audit against the corridor
            planes = np.ascontiguousarray(planes, self.np_dtype)
            n_planes = np.ascontiguousarray(n_planes, np.int32)
            cin.p_max, cin.n_planes, cin.planes = planes.shape[2], n_planes.ctypes.data, planes.ctypes.data
            o["cmax"] = np.zeros(B, self.np_dtype)
        for k, v in o.items():
            setattr(cout, k, v.ctypes.data)
        _check(lib().direct_traj_sample_batch(self.h, C.addressof(cin), C.addressof(cout)))
        return o

    def sample_device(self, cin, cout):
        """direct_traj_sample_batch with caller-built structs (device-resident arrays)."""
        _check(lib().direct_traj_sample_batch(self.h, C.addressof(cin), C.addressof(cout)))

    def sample_last_ms(self):
        ms = C.c_float()
        _check(lib().direct_traj_sample_last_ms(self.h, C.addressof(ms)))
        return ms.value

    # -- config-5 reduction through the C entry points (RCCL) ------------------------------------
    @staticmethod
    def rccl_unique_id():
        """128-byte ncclUniqueId (call on rank 0, ship to the other ranks)."""
        buf = C.create_string_buffer(128)
        _check(lib().direct_rccl_unique_id(buf))
        return buf.raw

    def rccl_comm_create(self, uid, n_ranks, rank):
        comm = C.c_void_p()
        _check(lib().direct_rccl_comm_create(self.h, C.c_char_p(uid), n_ranks, rank, C.addressof(comm)))
        return comm

    @staticmethod
    def rccl_comm_destroy(comm):
        _check(lib().direct_rccl_comm_destroy(comm))

    def gather_best(self, comm, n_ranks, rank, cost, rtn, bez, T, first_index, mem=abi.MEM_HOST, batch=None,
                    out_bez=None, out_T=None):
        """direct_ddp_gather_best.  Host mode: numpy arrays in, returns (index, cost, owner, bez, T) with numpy
        outputs; device mode: raw pointers in (and optional raw output pointers), returns (index, cost, owner)."""
        idx, val, owner = C.c_int64(), C.c_double(), C.c_int32()
        if mem == abi.MEM_HOST:
            cost = np.ascontiguousarray(cost, self.np_dtype)
            rtn = np.ascontiguousarray(rtn, np.int32)
            bez = np.ascontiguousarray(bez, self.np_dtype)
            T = np.ascontiguousarray(T, self.np_dtype)
            batch = cost.shape[0]
            ob = np.zeros((self.n_seg_max, 18), self.np_dtype)
            oT = np.zeros(self.n_seg_max, self.np_dtype)
            _check(lib().direct_ddp_gather_best(self.h, comm, n_ranks, rank, mem, cost.ctypes.data, rtn.ctypes.data,
                                                bez.ctypes.data, T.ctypes.data, batch, int(first_index), C.addressof(idx),
                                                C.addressof(val), C.addressof(owner), ob.ctypes.data, oT.ctypes.data))
            return idx.value, val.value, owner.value, ob, oT
        _check(lib().direct_ddp_gather_best(self.h, comm, n_ranks, rank, mem, C.c_void_p(cost), C.c_void_p(rtn),
                                            C.c_void_p(bez), C.c_void_p(T), int(batch), int(first_index), C.addressof(idx),
                                            C.addressof(val), C.addressof(owner), C.c_void_p(out_bez), C.c_void_p(out_T)))
        return idx.value, val.value, owner.value

    def launch_info(self):
        """direct_ddp_last_launch_info as a dict: how the last hot-kernel launch was scheduled + knot visits executed."""
        li = abi.LaunchInfo()
        _check(lib().direct_ddp_last_launch_info(self.h, C.addressof(li)))
        out = {n: int(getattr(li, n)) for n, _ in abi.LaunchInfo._fields_ if n != "reserved"}
        if hasattr(lib(), "direct_ddp_last_counters"):
            v = (C.c_uint64 * 4)()
            _check(lib().direct_ddp_last_counters(self.h, C.addressof(v)))
            out["helper_front_knots"] = int(v[2])
            out["accepted_line_searches"] = int(v[3])
        return out

    def sched_error(self):
        """Synchronise and read the sticky scheduler-error flag (include/direct_ddp.h)."""
        v = C.c_int32()
        _check(lib().direct_ddp_sched_error(self.h, C.addressof(v)))
        return v.value

    def sched_debug(self):
        """what the first wait that ran into the spin limit saw (include/direct_ddp.h, direct_ddp_sched_debug)"""
        v = (C.c_int32 * 64)()
        _check(lib().direct_ddp_sched_debug(self.h, v))
        return dict(zip(("ticket", "epoch", "trajectory", "done_epoch", "ticket_counter", "waves", "alive", "set", "wait_ms", "batch",
                         "tickets", "spins", "dbg_waiting_now", "dbg_running_now", "dbg_waiting_at_timeout", "dbg_running_at_timeout",
                         "dbg_early_ticket", "dbg_early_epoch", "dbg_early_done_epoch", "dbg_early_counter", "dbg_early_waiting",
                         "dbg_early_running", "dbg_early_ms", "dbg_early_set", "pad24", "pad25", "dbg_pub_in_now", "dbg_pub_out_now", "dbg_pub_in_at_timeout", "dbg_pub_out_at_timeout", "pad30", "pad31",
                         "m_none", "m_chunk", "m_help", "m_bwd", "m_bwd_failed", "m_before_fwd", "m_after_fwd", "m_after_exit", "m_done", "m_wait",
                         "m_r0", "m_r1", "m_r2", "m_r3", "m_r4", "m_r5", "m_r6", "m_r7", "m_r8", "m_r9", "m_r10", "m_rend", "m_helper", "m_other",
                         "w0", "b0", "w1", "b1", "w2", "b2", "pad62", "pad63"), list(v)))

    def best_cost(self, cost, rtn, mem=abi.MEM_HOST, batch=None):
        """(index, cost) of the cheapest trajectory with rtn >= 0.  cost/rtn: numpy arrays or raw pointers."""
        idx, val = C.c_int32(), C.c_double()
        if mem == abi.MEM_HOST:
            cost = np.ascontiguousarray(cost, self.np_dtype)
            rtn = np.ascontiguousarray(rtn, np.int32)
            batch = cost.shape[0]
            pc, pr = cost.ctypes.data, rtn.ctypes.data
        else:
            pc, pr = cost, rtn
        _check(lib().direct_ddp_best_cost(self.h, mem, C.c_void_p(pc), C.c_void_p(pr), int(batch),
                                          C.addressof(idx), C.addressof(val)))
        return idx.value, val.value
