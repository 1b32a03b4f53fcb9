"""The corridor-cluster oracle (oracle/cluster_ref.c) against the reference itself.

serialConvexTest is the one function of the reference that builds here from its own source
(oracle/_ref/libcluster_engine_ref.so <- /root/reference/polyhedron_generator/src/cluster_engine_cpu.cpp): the
restatement is checked against it bit for bit on seeded voxel maps, and both against the committed golden vectors
that reference build produced (tests/golden/make_cluster_golden.py).  PARITY PINNED for this row."""
import os

import numpy as np
import pytest

from direct_amd import problems
from oracle import clusterapi as ca

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases():
    return [np.load(os.path.join(GOLD, "cluster_convex_%s.npz" % n)) for n in ("a", "b")]


@pytest.mark.parametrize("which", ["restatement", "reference"])
def test_convex_test_against_golden(which):
    L = ca.lib() if which == "restatement" else ca.ref_lib()
    if L is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine and no prebuilt library)")
    name = "cl_serial_convex_test" if which == "restatement" else "ref_serial_convex_test"
    for g in _cases():
        dims = g["grid"].shape
        got = ca.serial_convex_test(L, name, g["cand"], g["cluster"], g["inside"], g["grid"], dims)
        assert np.array_equal(got, g["can_clu"])
        n = len(g["cand"])
        rows = np.random.default_rng(1).choice(np.arange(1, n), 12, replace=False)
        for i in rows:   # a sample of the candidate-candidate triangle (the full one is checked on the GPU)
            for j in range(i):
                v = ca.serial_convex_test(L, name, g["cand"][i:i + 1], g["cand"][j:j + 1], g["inside"], g["grid"], dims)[0]
                assert v == g["can_can"][i * (i - 1) // 2 + j]
        assert np.array_equal(ca.accept_sequential(g["can_clu"], g["can_can"]), g["accept"])


def test_restatement_equals_reference_on_random_scenes():
    R = ca.ref_lib()
    if R is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(3)
    n_false = n_total = 0
    for trial in range(12):
        dims = (int(rng.integers(10, 40)), int(rng.integers(10, 40)), int(rng.integers(5, 16)))
        grid = (rng.random(dims) < rng.uniform(0.002, 0.03)).astype(np.uint8)
        inside = np.zeros(dims, np.uint8)
        lo = [int(rng.integers(1, d // 2)) for d in dims]
        hi = [int(rng.integers(d // 2, d - 1)) for d in dims]
        inside[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = 1
        cand = np.stack([rng.integers(0, d, 60) for d in dims], 1)
        clu = np.stack([rng.integers(0, d, 6) for d in dims], 1)
        a = ca.serial_convex_test(ca.lib(), "cl_serial_convex_test", cand, clu, inside, grid, dims)
        b = ca.serial_convex_test(R, "ref_serial_convex_test", cand, clu, inside, grid, dims)
        assert np.array_equal(a, b)
        n_false += int((b == 0).sum())
        n_total += len(b)
    assert 0.1 < n_false / n_total < 0.9   # the scenes exercise both outcomes


@pytest.mark.parametrize("which", ["restatement", "reference"])
def test_polygon_generation_against_golden(which):
    if which == "reference" and ca.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    g = np.load(os.path.join(GOLD, "cluster_polygon_48.npz"))
    ca.use_reference_convex_test(which == "reference")
    try:
        off = 0
        for b, seed in enumerate(g["seeds"]):
            v, cl, it, rc = ca.polygon_generation(g["grid"], seed)
            n = int(g["cluster_num"][b])
            assert rc == 0 and it == g["iters"][b] and np.array_equal(v, g["vertex_idx"][b])
            assert np.array_equal(cl, g["cluster_xyz"][off:off + n])
            off += n
    finally:
        ca.use_reference_convex_test(False)


def test_polygon_generation_properties():
    """Properties that do not depend on any restatement: the inflated cube is obstacle free and maximal, every
    cluster voxel is free, the surface of the cube opens the cluster, no voxel is listed twice."""
    grid, seeds = problems.make_voxel_map((64, 64, 20), seed=5, n_pillars=30, n_boxes=12, n_rings=3)
    for seed in seeds[:10]:
        v, cl, it, rc = ca.polygon_generation(grid, seed)
        x0, x1, y0, y1, z0, z1 = v[7], v[1], v[15], v[9], v[23], v[17]
        assert x0 <= seed[0] <= x1 and y0 <= seed[1] <= y1 and z0 <= seed[2] <= z1
        assert not grid[x0:x1 + 1, y0:y1 + 1, z0:z1 + 1].any()
        for axis, (lo, hi, dim) in enumerate(((x0, x1, 64), (y0, y1, 64), (z0, z1, 20))):
            for face, edge in ((lo - 1, lo == 0), (hi + 1, hi == dim - 1)):
                if not edge:   # a face that stopped inside the map is blocked by an obstacle right behind it
                    sl = [slice(x0, x1 + 1), slice(y0, y1 + 1), slice(z0, z1 + 1)]
                    sl[axis] = face
                    assert grid[tuple(sl)].any()
        assert not grid[cl[:, 0], cl[:, 1], cl[:, 2]].any()
        assert len(np.unique(cl, axis=0)) == len(cl)
        if (x1 - x0 + 1) * (y1 - y0 + 1) * (z1 - z0 + 1) > 1:
            on_face = (cl[:, 0] == x0) | (cl[:, 0] == x1) | (cl[:, 1] == y0) | (cl[:, 1] == y1) | (cl[:, 2] == z0) | (cl[:, 2] == z1)
            n_surf = int(np.argmin(on_face)) if not on_face.all() else len(cl)
            box = (x1 - x0 + 1) * (y1 - y0 + 1) * (z1 - z0 + 1)
            inner = max(x1 - x0 - 1, 0) * max(y1 - y0 - 1, 0) * max(z1 - z0 - 1, 0)
            assert n_surf >= box - inner
