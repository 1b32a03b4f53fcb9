"""Per-trajectory chain composition of the bench launch (needs build_variants/chain.so from tools/chain_build.py)."""
import ctypes as C, sys, numpy as np
sys.path.insert(0, ".")
from direct_amd import abi, problems, solver
B, N = 4096, 100
b = problems.make_batch("free", B, N, seed=1000)
s = solver.DdpSolver(B, N, b.p_max, np.float32)
g0 = s.solve(abi.phase0_params(), b)
b1 = b.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, b.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
pf = abi.phase1_params(iter_max=20, fixed_iters=1)
g1 = s.solve(pf, b1)
ms, _ = s.last_kernel_ms()
lib = solver.lib()
buf = np.zeros(B * 64, np.int32); stride = C.c_int32()
lib.direct_ddp_debug_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
lib.direct_ddp_debug_state(s.h, buf.ctypes.data, C.addressof(stride))
w = stride.value // 4
st = buf[:B * w].reshape(B, w)
# fields: 10 doubles (20 ints) + 24 ints... locate the new fields: after int neg_time,nseg,nc0,npos -> offset
off = 20 + 20   # doubles(10) -> 20 ints; then 5 rows of 4 ints
cb = st[:, off:off+2].copy().view(np.int64)[:, 0]; cf = st[:, off+2:off+4].copy().view(np.int64)[:, 0]
nr, nf = st[:, off+4], st[:, off+5]
tot = cb + cf
print("kernel ms %.2f; sum of totals / 3072 slots = %.1f Mcycles (work bound); per-trajectory Mcycles:" % (ms, tot.sum() / 3072 / 1e6))
for name, a in (("bwd", cb), ("fwd", cf), ("total", tot)):
    print(name, "min/med/p90/max", np.percentile(a, [0, 50, 90, 100]) / 1e6)
print("rounds run by the owner med/p90/max", np.percentile(nr, [50, 90, 100]), "| fetched from helpers: total", int(nf.sum()), "max", nf.max())
i = np.argsort(tot)[-5:]
print("slowest 5: total", tot[i] / 1e6, "bwd", cb[i] / 1e6, "fwd", cf[i] / 1e6, "rounds", nr[i], "fetched", nf[i])
print("corr(total, rounds)", np.corrcoef(tot, nr)[0, 1])
