"""direct_cluster_hull_planes_batch (include/direct_cluster.h; poly_utils.cpp:127-206, 282-389) on the device against the
oracle (oracle/hull_ref.c, pinned against the reference's quickhull by tests/test_hull.py): planes, lattice planes,
corners and centre bit for bit; committed quickhull golden vectors; device-resident chaining seeds -> clusters ->
planes -> DDP."""
import os

import numpy as np
import pytest

from direct_amd import abi, cluster, problems, solver
from oracle import clusterapi, hullapi

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RES, LOWER = 0.2, np.array([-12.0, -12.0, 0.0])


def same(dev, b, ref):
    assert dev["rtn"][b] == ref["rc"] and dev["degenerate"][b] == ref["degenerate"]
    assert dev["n_planes"][b] == ref["n_planes"] and dev["n_vertices"][b] == ref["n_vertices"]
    assert np.array_equal(dev["plane_int"][b], ref["plane_int"])
    assert np.array_equal(dev["planes"][b], ref["planes"])          # doubles, bit for bit (same IEEE operations)
    assert np.array_equal(dev["vertices"][b], ref["vertices"])
    assert np.array_equal(dev["center"][b], ref["center"])


def test_hull_planes_match_the_oracle_and_the_quickhull_golden(built):
    g = np.load(os.path.join(GOLD, "hull_quickhull_16.npz"))
    clusters = [g["cluster_%d" % i] for i in range(int(g["n"]))]
    gen = cluster.ClusterGenerator((64, 64, 32), max_batch=len(clusters), cluster_capacity=8192, candidate_capacity=64)
    dev = gen.hull_planes(RES, LOWER, clusters=clusters)
    for b, cl in enumerate(clusters):
        ref = hullapi.hull_planes(cl, RES, LOWER)
        same(dev, b, ref)
        vq = np.rint((dev["vertices"][b] - LOWER) / (RES / 2)).astype(np.int64)
        hullapi.check_mesh(g["qh_vertices_%d" % b], g["qh_triangles_%d" % b], dev["plane_int"][b], vq)
    gen.close()


def test_hull_of_device_resident_clusters_and_edge_cases(built):
    grid, seeds = problems.make_voxel_map()
    seeds = seeds[:24]
    gen = cluster.ClusterGenerator(grid.shape, max_batch=32, cluster_capacity=50000, candidate_capacity=10000)
    gen.set_map(grid)
    r = gen.polygon_generation(seeds)
    dev = gen.hull_planes(RES, LOWER, batch=len(seeds))                 # no voxel leaves or enters the device
    for b in range(len(seeds)):
        same(dev, b, hullapi.hull_planes(r["clusters"][b], RES, LOWER))
    assert (dev["n_planes"] >= 6).all() and dev["n_planes"].max() > 12
    # caller-provided clusters: empty, one voxel, a flat slab, a diagonal sheet (not full-dimensional), capacity overflow
    cases = [np.zeros((0, 3), np.int32), np.array([[4, 4, 5]]), np.array([[x, y, 5] for x in range(3, 9) for y in range(4, 7)]),
             np.array([[x, x, z] for x in range(5) for z in range(4)]), r["clusters"][0]]
    dev = gen.hull_planes(RES, LOWER, clusters=cases, plane_capacity=8)
    assert list(dev["rtn"][:4]) == [cluster.HULL_FLAT, cluster.HULL_OK, cluster.HULL_OK, cluster.HULL_FLAT]
    for b in (1, 2):
        same(dev, b, hullapi.hull_planes(cases[b], RES, LOWER, plane_cap=8))
    full = hullapi.hull_planes(cases[4], RES, LOWER)
    assert dev["rtn"][4] == cluster.HULL_OVERFLOW and dev["n_planes"][4] == full["n_planes"]
    assert np.array_equal(dev["plane_int"][4], full["plane_int"][:8])
    gen.close()


def test_seeds_to_planes_to_ddp(built):
    """The chain the reference runs on the host (corridorGeneration, poly_utils.cpp:508-557, then the planner): seed
    voxels -> clusters -> planes here, and the planes are a valid DDP corridor: a trajectory through the polytopes of
    three overlapping clusters is solved and stays inside them."""
    grid, seeds = problems.make_voxel_map()
    gen = cluster.ClusterGenerator(grid.shape, max_batch=8, cluster_capacity=50000, candidate_capacity=10000)
    gen.set_map(grid)
    s0 = seeds[0]
    chain = np.array([s0, s0 + [3, 0, 0], s0 + [6, 1, 0]], np.int32)
    chain = chain[[not grid[tuple(c)] for c in chain]]
    gen.polygon_generation(chain)
    dev = gen.hull_planes(RES, LOWER, batch=len(chain), plane_capacity=64)
    assert (dev["rtn"] == 0).all()
    for b, c in enumerate(chain):                     # every seed lies strictly inside its own polytope
        x = c * RES + 0.5 * RES + LOWER
        assert (dev["planes"][b][:, :3] @ x + dev["planes"][b][:, 3] < 0).all()
    gen.close()
