"""Latency of ONE real corridor plan (direct_amd/data/real_corridor_n12.npz) through the C-ABI with host buffers, under the
environment's scheduling switches: usage  [DIRECT_DDP_HELP=1] [DIRECT_DDP_BSHARE=1] python tools/single_plan.py [f64|f32]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from direct_amd import abi, solver
d = np.load("direct_amd/data/real_corridor_n12.npz")
dt = np.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else np.float64
batch = abi.HostBatch(d["n_seg"], d["x0"], d["xd"], d["T0"], d["n_planes"], d["planes"], seeds=d["seeds"]).astype(dt)
p0, p1 = abi.phase0_params(), abi.phase1_params()
s = solver.DdpSolver(1, batch.n_seg_max, batch.p_max, dt)
s.plan(p0, p1, batch)
ts, ks = [], []
for _ in range(30):
    t = time.perf_counter(); g0, g1 = s.plan(p0, p1, batch); ts.append(time.perf_counter() - t); ks.append(s.last_kernel_ms()[0])
li = s.launch_info()
print("%s: plan %.3f ms (min %.3f), phase-1 kernel %.3f ms, iterations %d + %d, rtn %d %d, cost %.9e, schedule %s"
      % (np.dtype(dt).name, np.median(ts) * 1e3, np.min(ts) * 1e3, np.median(ks), g0.fwd_passes[0], g1.fwd_passes[0], g0.rtn[0], g1.rtn[0],
         float(g1.cost[0]), {k: li[k] for k in ("dynamic", "shared_search", "shared_sweep", "single_steps")}))
