"""Corridor-cluster generation: batched polygonGeneration on the device against the CPU oracle on the same seeds.
usage: python tests/soak/cluster_bench.py [n_seeds] [X Y Z]    (run through gpurun; add rocprofv3 --kernel-trace --stats for
per-kernel times).  Prints one JSON line."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from direct_amd import cluster, problems  # noqa: E402
from oracle import clusterapi as ca  # noqa: E402
from oracle import hullapi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dims = tuple(int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (200, 200, 40)
grid, seeds = problems.make_voxel_map(dims, seed=7, n_pillars=170, n_boxes=70, n_rings=12)
seeds = seeds[:n]
gen = cluster.ClusterGenerator(dims, max_batch=n, cluster_capacity=50000, candidate_capacity=10000)
gen.set_map(grid)
gen.polygon_generation(seeds[:2])   # warm up
ts = []
for rep in range(3):
    t = time.perf_counter()
    r = gen.polygon_generation(seeds, 1000, 50)
    ts.append(time.perf_counter() - t)
kms = gen.last_ms()
ca.use_reference_convex_test(ca.ref_lib() is not None)
m = min(n, 8)
t = time.perf_counter()
ref = [ca.polygon_generation(grid, s, 1000, 50) for s in seeds[:m]]
cpu = (time.perf_counter() - t) / m
ca.use_reference_convex_test(False)
same = all(np.array_equal(r["clusters"][b], ref[b][1]) for b in range(m))
# hull -> planes of the same clusters, still resident on the device (poly_utils.cpp:301-389, 127-206)
RES, LOWER = 0.2, np.array([-20.0, -20.0, 0.0])
gen.hull_planes(RES, LOWER, batch=n)
hts, hk = [], []
for rep in range(3):
    t = time.perf_counter()
    hp = gen.hull_planes(RES, LOWER, batch=n)
    hts.append(time.perf_counter() - t)
    hk.append(gen.last_ms())
cts = []   # the chain as direct::polyhedronGenerator::getConvexPolyBatch runs it: clusters stay on the device, planes come back
for rep in range(3):
    t = time.perf_counter()
    gen.polygon_generation(seeds, 1000, 50, fetch_clusters=False)
    hp2 = gen.hull_planes(RES, LOWER, batch=n, vertex_capacity=1)
    cts.append(time.perf_counter() - t)
t = time.perf_counter()
href = [hullapi.hull_planes(r["clusters"][b], RES, LOWER) for b in range(m)]
hcpu = (time.perf_counter() - t) / m
t = time.perf_counter()
pts = [hullapi.lattice_points(r["clusters"][b], 0).astype(np.float64) for b in range(m)]
t = time.perf_counter()
qh = [hullapi.reference_quickhull(p) for p in pts] if hullapi.ref_lib() is not None else None
qcpu = (time.perf_counter() - t) / m
hsame = all(np.array_equal(hp["planes"][b], href[b]["planes"]) and np.array_equal(hp["vertices"][b], href[b]["vertices"]) for b in range(m))
hull = {"device_wall_ms_best": min(hts) * 1e3, "device_event_ms": min(hk), "device_ms_per_seed": min(hts) * 1e3 / n,
        "planes_mean": float(hp["n_planes"].mean()), "planes_max": int(hp["n_planes"].max()), "corners_mean": float(hp["n_vertices"].mean()),
        "rtn_ok": int((hp["rtn"] == 0).sum()), "cpu_checker_ms_per_seed": hcpu * 1e3,
        "reference_quickhull_ms_per_seed (hull only, no H-rep)": qcpu * 1e3 if qh is not None else None,
        "bit_identical_to_cpu_on_first_%d" % m: bool(hsame)}
rays = 0   # rays cast = sum over rounds of candidates x (cluster + earlier candidates): not tracked on the device; report voxels instead
print(json.dumps({"seeds": n, "map": dims, "obstacle_frac": float(grid.mean()), "rtn_ok": int((r["rtn"] == 0).sum()),
                  "cluster_voxels_mean": float(r["cluster_num"].mean()), "rounds_mean": float(r["iters"].mean()),
                  "device_wall_ms_best": min(ts) * 1e3, "device_event_ms": kms, "device_ms_per_seed": min(ts) * 1e3 / n,
                  "cpu_oracle_ms_per_seed": cpu * 1e3, "cpu_kind": "reference serialConvexTest + restated loops, 1 thread",
                  "bit_identical_to_cpu_on_first_%d" % m: bool(same), "hull_planes": hull,
                  "seeds_to_planes_chain": {"wall_ms_best": min(cts) * 1e3, "ms_per_seed": min(cts) * 1e3 / n,
                                            "same_planes": bool(all(np.array_equal(a, b) for a, b in zip(hp2["planes"], hp["planes"])))}}))
gen.close()
