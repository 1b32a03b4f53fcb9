"""Per-phase cycles of workgroup 0 (needs a -DDDP_TIMING build passed via DIRECT_DDP_LIB)."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from direct_amd import abi, problems, solver
kind = sys.argv[1] if len(sys.argv) > 1 else "corridor"
B, N = (int(sys.argv[2]) if len(sys.argv) > 2 else 4096), 100
b = problems.make_batch(kind, B, N, seed=1000)
s = solver.DdpSolver(B, N, b.p_max, np.float32)
g0 = s.solve(abi.phase0_params(), b)
b1 = b.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, b.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
pf = abi.phase1_params(iter_max=20, fixed_iters=1)
lib = solver.lib()
lib.direct_ddp_debug_phase_cycles.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 64)()
lib.direct_ddp_debug_phase_cycles(buf, 1)
g1 = s.solve(pf, b1)
ms, _ = s.last_kernel_ms()
lib.direct_ddp_debug_phase_cycles(buf, 0)
names = "B_L B_T1 B_T2 B_R1 B_S B_S2 B_H B_C B_G B_R2 B_END F_L F_D F_T F_R F_END X_T X_G X_A".split()
tot = sum(buf[i] for i in range(19))
print("kernel %.1f ms; workgroup 0 total %.1f Mcycles" % (ms, tot / 1e6))
for i, n in enumerate(names):
    c, k = buf[i], buf[32 + i]
    print("%-6s %10.0f cycles total  %7d visits  %8.0f cycles/visit  %5.1f%%" % (n, c, k, c / max(k, 1), 100.0 * c / tot))
