"""Real-corridor replay plans on the library named by DIRECT_DDP_LIB: 710 plans and `rep` copies of them (one class).
usage: python tools/rc_ab.py [rep]   prints wall / kernel times and a checksum of the results (A/B of library builds)"""
import hashlib, sys, time
import numpy as np
sys.path.insert(0, ".")
from direct_amd import abi, solver
from tests import real_corridor_lib
batch, meta = real_corridor_lib.real_corridor_batch(64)
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 8
p0, p1 = abi.phase0_params(), abi.phase1_params()
for r in (1, rep):
    big = batch.select(np.tile(np.arange(batch.batch), r))
    for dt in (np.float64, np.float32):
        s = solver.DdpSolver(big.batch, int(big.n_seg_max), int(big.p_max), dt)
        hb = big.astype(dt)
        s.plan(p0, p1, hb)
        best = None
        for _ in range(3):
            t = time.perf_counter(); g0, g1 = s.plan(p0, p1, hb); wall = time.perf_counter() - t
            k1 = s.last_kernel_ms()[0]
            best = (wall, k1) if best is None or wall < best[0] else best
        s.close()
        h = hashlib.sha1()
        for g in (g0, g1):
            for f in ("rtn", "iter_used", "cost", "T", "bez"):
                h.update(np.ascontiguousarray(getattr(g, f)).tobytes())
        its = int(g0.fwd_passes.sum() + g1.fwd_passes.sum())
        print("B=%d %s: plan wall %.1f ms, phase-1 kernel %.1f ms, %d iterations -> %.2f M iter/s, sha1 %s"
              % (big.batch, np.dtype(dt).name, best[0] * 1e3, best[1], its, its / best[0] / 1e6, h.hexdigest()[:12]))
