"""Corridor-cluster kernels (include/direct_cluster.h) against the reference: bit-exact integer / byte parity.

k_convex + k_resolve are checked against the REFERENCE'S OWN serialConvexTest (oracle/_ref, built from its source)
on the committed golden vectors and on fresh scenes; the whole polygonGeneration against the restated loops with
the reference function plugged in.  PARITY PINNED."""
import os

import numpy as np
import pytest

from direct_amd import cluster, problems
from oracle import clusterapi as ca

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["a", "b"])
def test_convex_test_kernel_matches_reference_golden(built, name):
    g = np.load(os.path.join(GOLD, "cluster_convex_%s.npz" % name))
    gen = cluster.ClusterGenerator(g["grid"].shape, max_batch=1, cluster_capacity=4096, candidate_capacity=1024)
    gen.set_map(g["grid"])
    clu, cc, acc = gen.convex_test(g["inside"], g["cand"], g["cluster"])
    assert np.array_equal(clu, g["can_clu"])
    assert np.array_equal(cc, g["can_can"])        # the whole packed triangle, every ray
    assert np.array_equal(acc, g["accept"])
    gen.close()


def test_polygon_generation_matches_golden(built):
    g = np.load(os.path.join(GOLD, "cluster_polygon_48.npz"))
    gen = cluster.ClusterGenerator(g["grid"].shape, max_batch=16, cluster_capacity=8192, candidate_capacity=4096)
    gen.set_map(g["grid"])
    r = gen.polygon_generation(g["seeds"])
    assert (r["rtn"] == 0).all()
    assert np.array_equal(r["vertex_idx"], g["vertex_idx"])
    assert np.array_equal(r["cluster_num"], g["cluster_num"]) and np.array_equal(r["iters"], g["iters"])
    assert np.array_equal(np.concatenate(r["clusters"]), g["cluster_xyz"])   # same voxels in the same order
    # a second call on the same handle (flagClear) and a different batch composition give the same rows
    r2 = gen.polygon_generation(g["seeds"][::-1][:5])
    for q in range(5):
        assert np.array_equal(r2["clusters"][q], r["clusters"][len(g["seeds"]) - 1 - q])
    gen.close()


def test_large_candidate_capacity_takes_the_general_kernels(built):
    """Above 16384 candidates per round the row of the bit matrix no longer fits the registers / LDS of the fast
    accept loop and of k_convex's queued candidate rays: the general kernels (k_resolve_pipe, un-queued rows) run
    instead and must give the same clusters."""
    g = np.load(os.path.join(GOLD, "cluster_polygon_48.npz"))
    gen = cluster.ClusterGenerator(g["grid"].shape, max_batch=16, cluster_capacity=8192, candidate_capacity=20000)
    gen.set_map(g["grid"])
    r = gen.polygon_generation(g["seeds"])
    assert (r["rtn"] == 0).all()
    assert np.array_equal(r["cluster_num"], g["cluster_num"]) and np.array_equal(r["iters"], g["iters"])
    assert np.array_equal(np.concatenate(r["clusters"]), g["cluster_xyz"])
    gen.close()


def test_polygon_generation_on_a_larger_map_against_the_oracle(built):
    """120 x 120 x 24 map, 48 seeds in one batch, against the oracle with the reference's serialConvexTest."""
    grid, seeds = problems.make_voxel_map()
    seeds = seeds[:48]
    gen = cluster.ClusterGenerator(grid.shape, max_batch=48, cluster_capacity=50000, candidate_capacity=10000)
    gen.set_map(grid)
    r = gen.polygon_generation(seeds, itr_inflate_max=1000, itr_cluster_max=50)
    have_ref = ca.ref_lib() is not None
    ca.use_reference_convex_test(have_ref)
    try:
        for b in range(0, 48, 4):
            v, cl, it, rc = ca.polygon_generation(grid, seeds[b])
            assert rc == 0 and r["rtn"][b] == 0
            assert np.array_equal(r["vertex_idx"][b], v) and r["iters"][b] == it
            assert np.array_equal(r["clusters"][b], cl)
    finally:
        ca.use_reference_convex_test(False)
    # size-independent properties of every row: free voxels only, no duplicates, cube obstacle free
    for b in range(48):
        cl, v = r["clusters"][b], r["vertex_idx"][b]
        assert not grid[cl[:, 0], cl[:, 1], cl[:, 2]].any()
        assert len(np.unique(cl, axis=0)) == len(cl)
        assert not grid[v[7]:v[1] + 1, v[15]:v[9] + 1, v[23]:v[17] + 1].any()
    # cluster-off mode of paramSet (1000, 0): the cluster is the cube's surface
    r0 = gen.polygon_generation(seeds[:4], 1000, 0)
    for b in range(4):
        assert np.array_equal(r0["clusters"][b], ca.polygon_generation(grid, seeds[b], 1000, 0)[1]) and r0["iters"][b] == 0
    assert gen.last_ms() > 0
    gen.close()


def test_edge_cases(built):
    grid = np.zeros((12, 10, 6), np.uint8)
    grid[5, :, :] = 1                      # a wall splits the map
    grid[0, 0, 0] = 1
    gen = cluster.ClusterGenerator(grid.shape, max_batch=8, cluster_capacity=64, candidate_capacity=64)
    gen.set_map(grid)
    seeds = np.array([[2, 3, 3], [20, 0, 0], [-1, 2, 2], [8, 4, 2]], np.int32)
    r = gen.polygon_generation(seeds)
    assert r["rtn"][1] == cluster.CLUSTER_BAD_SEED and r["rtn"][2] == cluster.CLUSTER_BAD_SEED
    assert r["cluster_num"][1] == 0 and r["cluster_num"][2] == 0
    assert r["rtn"][0] == cluster.CLUSTER_OVERFLOW     # the 5 x 10 x 6 half-room has more than 64 surface voxels
    gen.close()
    gen = cluster.ClusterGenerator(grid.shape, max_batch=8, cluster_capacity=4096, candidate_capacity=1024)
    gen.set_map(grid)
    r = gen.polygon_generation(seeds[[0, 3]])
    for b, s in enumerate(seeds[[0, 3]]):
        v, cl, it, rc = ca.polygon_generation(grid, s)
        assert np.array_equal(r["clusters"][b], cl) and np.array_equal(r["vertex_idx"][b], v) and r["iters"][b] == it
    # degenerate cube (one voxel thick): no clustering (cluster_server_cpu.cpp:509-518)
    thin = np.ones((6, 6, 6), np.uint8)
    thin[2, 1:5, 1:5] = 0
    g2 = cluster.ClusterGenerator(thin.shape, max_batch=1, cluster_capacity=256, candidate_capacity=256)
    g2.set_map(thin)
    rt = g2.polygon_generation([[2, 2, 2]])
    v, cl, it, rc = ca.polygon_generation(thin, (2, 2, 2))
    assert np.array_equal(rt["clusters"][0], cl) and rt["iters"][0] == 0 and len(cl) == 16
    # a seed enclosed on all sides: the one-voxel cube
    one = np.ones((5, 5, 5), np.uint8)
    one[2, 2, 2] = 0
    g3 = cluster.ClusterGenerator(one.shape, max_batch=1, cluster_capacity=16, candidate_capacity=16)
    g3.set_map(one)
    ro = g3.polygon_generation([[2, 2, 2]])
    assert ro["cluster_num"][0] == 1 and np.array_equal(ro["clusters"][0], [[2, 2, 2]])
    g2.close(); g3.close(); gen.close()
