"""Timeline of a natural-exit launch of config 2 (needs build_variants/iter_times.so): when every trajectory finished,
and how long each outer iteration of the slowest one took against the number of trajectories still running.
usage (gpurun): DIRECT_DDP_LIB=build_variants/iter_times.so python tools/iter_times.py"""
import ctypes as C, json, sys
import numpy as np
sys.path.insert(0, ".")
from direct_amd import abi, problems, solver
B, N = 4096, 100
sb = problems.make_batch("free", B, N, seed=1000).astype(np.float32)
s = solver.DdpSolver(B, N, int(sb.p_max), np.float32)
g0 = s.solve(abi.phase0_params(), sb)
b1 = sb.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, sb.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
g = s.solve(abi.phase1_params(), b1)
ms = s.last_kernel_ms()[0]
lib = solver.lib()
stride = C.c_int32()
buf = np.zeros(B * 1024 // 4, np.int32)
lib.direct_ddp_debug_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
lib.direct_ddp_debug_state(s.h, buf.ctypes.data, C.addressof(stride))
w = stride.value // 4
st = buf[:B * w].reshape(B, w)
t = st[:, 44:44 + 192].copy().view(np.int64)            # t_it[96] at byte 176
n = g.fwd_passes.astype(int)
t0 = int(t[:, 0].min())                                  # first stamp of anyone (end of its first iteration)
def stamps(b):                                           # the valid prefix: increasing, within a second of the start
    v = [int(t[b, 0])]
    for k in range(1, 96):
        if not (v[-1] < t[b, k] < t0 + 100_000_000):
            break
        v.append(int(t[b, k]))
    return np.array(v) - t0
S = [stamps(b) for b in range(B)]
fin = np.array([x[-1] for x in S])
lag = int(np.argmax(fin))
tl = S[lag] / 100.0                                      # us
dur = np.diff(np.r_[0, tl])
live = [(fin / 100.0 > x).sum() for x in tl]
n = np.array([len(x) for x in S])
out = {"kernel_ms": ms, "slowest": lag, "iterations": int(n[lag]), "finish_ms_quantiles_10_50_90_99_100": [float(v) / 1e5 for v in np.quantile(fin, [0.1, 0.5, 0.9, 0.99, 1.0])],
       "slowest_iteration_us": [int(v) for v in dur], "live_at_its_iteration_end": [int(v) for v in live]}
print(json.dumps(out))
