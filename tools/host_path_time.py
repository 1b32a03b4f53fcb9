"""Wall time of the HOST-buffer boundary (what a reference-side caller sees, PCIe included) against the kernel time:
direct_ddp_solve_batch with numpy arrays, fixed-20 phase-1 solve of config 2.  usage: host_path_time.py [B]"""
import sys, time
import ctypes as C
import numpy as np
sys.path.insert(0, ".")
from direct_amd import abi, problems, solver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b = problems.make_batch("free", B, 100, seed=1000).astype(np.float32)
s = solver.DdpSolver(B, 100, b.p_max, np.float32)
g0 = s.solve(abi.phase0_params(), b)
b1 = b.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, b.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
pf = abi.phase1_params(iter_max=20, fixed_iters=1)
hb = s._host_batch(b1)
res = abi.HostResult(hb.batch, hb.n_seg_max, s.np_dtype)
cin, cout = hb.c_struct(), res.c_struct()
lib = solver.lib()
for _ in range(2):
    lib.direct_ddp_solve_batch(s.h, C.addressof(pf), C.addressof(cin), C.addressof(cout))
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    lib.direct_ddp_solve_batch(s.h, C.addressof(pf), C.addressof(cin), C.addressof(cout))
    ts.append((time.perf_counter() - t0) * 1e3)
ms = s.last_kernel_ms()[0]
inb = sum(np.asarray(getattr(hb, f)).nbytes for f in ("x0", "xd", "T0", "planes", "n_planes", "n_seg") if getattr(hb, f, None) is not None)
print("B = %d: host-buffer solve %.2f ms (min of 10, median %.2f), iterate kernel %.2f ms -> boundary overhead %.2f ms; %.3f M iter/s PCIe-inclusive"
      % (B, min(ts), float(np.median(ts)), ms, min(ts) - ms, B * 20 / min(ts) / 1e3))
