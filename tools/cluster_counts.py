"""Ray statistics of the cluster bench (needs build_variants/cluster_counts.so from tools/cluster_counts_build.py):
rays tested / walked / DDA steps per batch.  usage (gpurun): DIRECT_DDP_LIB=build_variants/cluster_counts.so python tools/cluster_counts.py"""
import ctypes as C, json, sys
import numpy as np
sys.path.insert(0, ".")
from direct_amd import cluster, problems, solver
dims = (200, 200, 40)
grid, seeds = problems.make_voxel_map(dims, seed=7, n_pillars=170, n_boxes=70, n_rings=12)
gen = cluster.ClusterGenerator(dims, max_batch=64, cluster_capacity=50000, candidate_capacity=10000)
gen.set_map(grid)
L = solver.lib()
cnt = (C.c_ulonglong * 4)()
L.direct_cluster_debug_counts(cnt, 1)
r = gen.polygon_generation(seeds[:64], 1000, 50)
L.direct_cluster_debug_counts(cnt, 1)
t, w, st, cc = (int(v) for v in cnt)
print(json.dumps({"cluster_rays_tested": t, "cluster_rays_walked": w, "dda_steps": st, "candidate_rays_tested": cc,
                  "walked_frac_of_cluster_rays": w / max(t, 1), "steps_per_walk": st / max(w, 1), "kernel_ms": gen.last_ms()}))
