"""Hull -> planes soak: clusters of many seeds on maps of different obstacle density (generated on the device), the
device's planes / corners / centre against the oracle (oracle/hull_ref.c) bit for bit, the oracle against the reference's
quickhull on every cluster (oracle/_ref).  Prints one JSON line.  usage (gpurun): python tests/soak/hull_soak.py [maps]"""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from direct_amd import cluster  # noqa: E402
from oracle import hullapi  # noqa: E402

n_maps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
RES, LOWER = 0.2, np.array([-10.0, -10.0, 0.0])
dims = (100, 100, 32)
rng = np.random.default_rng(77)
gen = cluster.ClusterGenerator(dims, max_batch=64, cluster_capacity=50000, candidate_capacity=10000)
tot = dict(clusters=0, degenerate=0, flat=0, overflow=0, bit_identical=0, quickhull_pinned=0, planes_max=0, voxels_max=0)
have_ref = hullapi.ref_lib() is not None
for mi in range(n_maps):
    dens = [0.002, 0.01, 0.03, 0.08, 0.2][mi % 5]
    grid = (rng.random(dims) < dens).astype(np.uint8)
    if mi % 4 == 3:  # slabs: flat clusters (free layers one voxel thick)
        grid[:, :, 1::2] = 1
    gen.set_map(grid)
    free = np.argwhere(grid == 0)
    seeds = free[rng.choice(len(free), 64, replace=False)].astype(np.int32)
    r = gen.polygon_generation(seeds)
    dev = gen.hull_planes(RES, LOWER, batch=64, plane_capacity=128, vertex_capacity=512)
    for b in range(64):
        if r["rtn"][b] != 0:
            continue
        ref = hullapi.hull_planes(r["clusters"][b], RES, LOWER, plane_cap=128, vert_cap=512)
        tot["clusters"] += 1
        tot["degenerate"] += int(ref["degenerate"])
        tot["flat"] += int(ref["rc"] == 3)
        tot["overflow"] += int(ref["rc"] == 1)
        tot["planes_max"] = max(tot["planes_max"], int(ref["n_planes"]))
        tot["voxels_max"] = max(tot["voxels_max"], len(r["clusters"][b]))
        same = (dev["rtn"][b] == ref["rc"] and dev["degenerate"][b] == ref["degenerate"] and dev["n_planes"][b] == ref["n_planes"]
                and dev["n_vertices"][b] == ref["n_vertices"])
        if same and ref["rc"] == 0:
            same = (np.array_equal(dev["plane_int"][b], ref["plane_int"]) and np.array_equal(dev["planes"][b], ref["planes"])
                    and np.array_equal(dev["vertices"][b], ref["vertices"]) and np.array_equal(dev["center"][b], ref["center"]))
        tot["bit_identical"] += int(bool(same))
        if have_ref and ref["rc"] == 0:
            try:
                hullapi.check_against_quickhull(hullapi.lattice_points(r["clusters"][b], ref["degenerate"]), ref["plane_int"], ref["vert_q"])
                tot["quickhull_pinned"] += 1
            except AssertionError as ex:
                tot.setdefault("quickhull_mismatch", []).append(str(ex)[:200])
tot["reference_quickhull_available"] = have_ref
print(json.dumps(tot))
