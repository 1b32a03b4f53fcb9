"""The whole chain on data the reference's own pipeline would produce: voxel map -> grid paths -> corridorGeneration
(direct_amd/host/poly_utils.hpp on the device path) -> replay protocol of corridorRecCallBack (first n polytopes,
n = 2 ..; TRP:316-320) -> two-phase DDP plan, one ragged batch.  Reports what the synthetic corridors of BASELINE
configs 3 - 5 cannot: return codes / feasibility of REAL polytopes, containment of the result,
and the times next to the CPU oracle on a sample.
usage (through gpurun): python tests/soak/real_corridor_bench.py [n_paths] > gpurun_out/r03_real_corridors.json"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from direct_amd import abi, corridor_io, problems, solver  # noqa: E402
from oracle import refapi  # noqa: E402

from tests import real_corridor_lib  # noqa: E402

n_paths = int(sys.argv[1]) if len(sys.argv) > 1 else 64
batch, out = real_corridor_lib.real_corridor_batch(n_paths)
p0, p1 = abi.phase0_params(), abi.phase1_params()
hist = lambda a: {str(int(v)): int(c) for v, c in zip(*np.unique(a, return_counts=True))}
for name, dt in (("f64", np.float64), ("f32", np.float32)):
    s = solver.DdpSolver(batch.batch, int(batch.n_seg_max), int(batch.p_max), dt)
    hb = batch.astype(dt)
    s.plan(p0, p1, hb)
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        g0, g1 = s.plan(p0, p1, hb)
        ts.append(time.perf_counter() - t)
    d = s.sample(batch.n_seg, g1.bez, g1.T, 0.05, 8192, derivs=0, n_planes=batch.n_planes, planes=batch.planes.astype(dt))
    s.close()
    okm = (g1.rtn >= 0) & (g1.infeas_out == 0)   # left the infeasible mode: every constraint row holds
    out["device_" + name] = {"plan_wall_ms_best": min(ts) * 1e3, "phase0_rtn": hist(g0.rtn), "phase1_rtn": hist(g1.rtn),
                             "phase1_still_infeasible": int((g1.infeas_out != 0).sum()),
                             "iterations_total": int(g0.iter_used.sum() + g1.iter_used.sum()),
                             "iterations_per_s": float((g0.iter_used.sum() + g1.iter_used.sum()) / min(ts)),
                             "feasible_results": int(okm.sum()), "worst_plane_value_of_feasible_samples": float(d["cmax"][okm].max()) if okm.any() else None}
    if name == "f64":
        keep = (g0, g1)
m = min(batch.batch, 64)
sb = batch.select(np.arange(m))
t = time.perf_counter()
r0, r1 = refapi.plan_batch(p0, p1, sb, n_threads=1)
cpu = time.perf_counter() - t
g0, g1 = keep
both = (r1.rtn >= 0) & (g1.rtn[:m] >= 0)
out["cpu_oracle_sample"] = {"problems": m, "one_thread_ms": cpu * 1e3, "iterations_per_s": float((r0.iter_used.sum() + r1.iter_used.sum()) / cpu),
                            "same_rtn_phase0": int((r0.rtn == g0.rtn[:m]).sum()), "same_rtn_phase1": int((r1.rtn == g1.rtn[:m]).sum()),
                            "same_iterations_phase1": int((r1.iter_used == g1.iter_used[:m]).sum()),
                            "cost_dev_max_f64": float(np.abs(g1.cost[:m][both] / r1.cost[both] - 1).max()) if both.any() else None}
print(json.dumps(out))
