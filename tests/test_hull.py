"""Hull -> planes (SURVEY.md 8f-4 tail; poly_utils.cpp:127-206, 282-389) without a device: the oracle
(oracle/hull_ref.c) against the REFERENCE'S OWN quickhull - committed golden vectors and, where oracle/_ref is built,
live on fresh clusters -, the geometric properties no implementation may break, and the device algorithm's predicates
and formulas (direct_amd/csrc/hull_core.h, built for the host by tests/emu/hull_emu.cpp) against the oracle, bit for bit.
PARITY: pinned for the facet planes and the corner set (quickhull); cdd's row order and per-plane vertex choice are
not reproducible and are defined in include/direct_cluster.h."""
import os

import numpy as np
import pytest

from oracle import clusterapi, hullapi
from tests.emu import hullemu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RES, LOWER = 0.2, np.array([-12.0, -12.0, 0.0])
KEYS = ("plane_int", "planes", "vert_q", "vertices", "center")


def golden():
    g = np.load(os.path.join(GOLD, "hull_quickhull_16.npz"))
    return [(g["cluster_%d" % i], int(g["deg_%d" % i]), g["qh_vertices_%d" % i], g["qh_triangles_%d" % i]) for i in range(int(g["n"]))]


def properties(cluster, r):
    """what every correct H-rep / V-rep of the cluster's hull satisfies (exact integer checks)"""
    pts = hullapi.lattice_points(cluster, r["degenerate"])
    P = r["plane_int"].astype(np.int64)
    resid = pts @ P[:, :3].T + P[:, 3]
    assert (resid <= 0).all()                                   # every point inside every half-space
    for k in range(len(P)):                                     # every plane is a facet: three non-collinear points on it
        on = pts[resid[:, k] == 0]
        assert len(on) >= 3 and np.linalg.matrix_rank((on - on[0]).astype(np.float64)) == 2
        assert np.gcd.reduce(np.abs(P[k, :3])) == 1
    assert (np.diff(np.lexsort(P.T[::-1])) == 1).all()          # rows ascending by (nx, ny, nz, K)
    V = r["vert_q"].astype(np.int64)
    rv = V @ P[:, :3].T + P[:, 3]
    for i in range(len(V)):                                     # every corner lies on three independent facets
        assert np.linalg.matrix_rank(P[rv[i] == 0][:, :3].astype(np.float64)) == 3
    assert len({tuple(v) for v in V}) == len(V)
    # doubles: unit normals, offsets consistent with the lattice planes, half-voxel inflation of axis-aligned faces
    D = r["planes"]
    assert np.allclose(np.linalg.norm(D[:, :3], axis=1), 1.0, rtol=0, atol=1e-15)
    world = pts * (RES / 2) + LOWER
    margin = world @ D[:, :3].T + D[:, 3]
    axis = (P[:, :3] != 0).sum(axis=1) == 1
    infl = np.where(axis & (r["degenerate"] == 0), RES / 2, 0.0)
    assert np.abs(margin.max(axis=0) + infl).max() < 1e-12     # the touching points sit exactly `infl` inside


def test_oracle_against_the_reference_quickhull_golden():
    for cl, deg, vbq, tri in golden():
        r = hullapi.hull_planes(cl, RES, LOWER)
        assert r["rc"] == 0 and r["degenerate"] == deg
        hullapi.check_mesh(vbq, tri, r["plane_int"], r["vert_q"])
        properties(cl, r)


@pytest.mark.skipif(hullapi.ref_lib() is None and not os.path.exists("/root/reference"), reason="oracle/_ref not built")
def test_oracle_against_the_reference_quickhull_live():
    hullapi.build()
    rng = np.random.default_rng(3)
    grid = (rng.random((90, 90, 28)) < 0.006).astype(np.uint8)
    done = 0
    while done < 10:
        seed = [int(rng.integers(15, 75)), int(rng.integers(15, 75)), int(rng.integers(5, 23))]
        if grid[tuple(seed)]:
            continue
        cl = clusterapi.polygon_generation(grid, seed)[1]
        r = hullapi.hull_planes(cl, RES, LOWER)
        assert r["rc"] == 0
        hullapi.check_against_quickhull(hullapi.lattice_points(cl, r["degenerate"]), r["plane_int"], r["vert_q"])
        done += 1


def test_device_algorithm_on_the_host_matches_the_oracle_bitwise():
    cases = [g[0] for g in golden()]
    rng = np.random.default_rng(8)
    for dens in (0.004, 0.02, 0.06):
        grid = (rng.random((80, 80, 24)) < dens).astype(np.uint8)
        for _ in range(6):
            seed = [int(rng.integers(12, 68)), int(rng.integers(12, 68)), int(rng.integers(4, 20))]
            if not grid[tuple(seed)]:
                cases.append(clusterapi.polygon_generation(grid, seed)[1])
    # a cluster that is not convex as a voxel set, and one with a long diagonal edge (collinear candidates)
    cases.append(np.array([[x, y, z] for x in range(6) for y in range(6) for z in range(3) if not (2 <= x <= 3 and y >= 2)]))
    cases.append(np.array([[x, y, z] for x in range(8) for y in range(8) for z in range(4) if x + y <= 7]))
    for cl in cases:
        r, g = hullapi.hull_planes(cl, RES, LOWER), hullemu.hull_planes(cl, RES, LOWER)
        assert r["rc"] == g["rc"] == 0 and r["degenerate"] == g["degenerate"]
        for k in KEYS:
            assert np.array_equal(r[k], g[k]), k
        assert g["n_cand"] <= max(64, len(cl) // 2)


def test_edge_cases():
    one = hullapi.hull_planes([[4, 4, 5]], RES, LOWER)       # a single voxel: its own cube, no inflation (degenerate)
    assert one["rc"] == 0 and one["degenerate"] == 1 and one["n_planes"] == 6 and one["n_vertices"] == 8
    c = np.array([4, 4, 5]) * RES + 0.5 * RES + LOWER
    # polyHrep2Utils' centre is the mean of ONE VERTEX PER PLANE (:129-144), not the centroid: inside the cube, no more
    assert (np.abs(one["center"] - c) <= RES / 2).all()
    assert np.allclose(sorted(one["planes"][:, 3] + one["planes"][:, :3] @ c), [-RES / 2] * 6, atol=1e-15)
    box = np.array([[x, y, z] for x in range(2, 6) for y in range(3, 5) for z in range(1, 4)])
    b = hullapi.hull_planes(box, RES, LOWER)                  # a solid box: six faces, each half a voxel outside the centres
    assert b["degenerate"] == 0 and b["n_planes"] == 6 and b["n_vertices"] == 8
    lo, hi = box.min(axis=0) * RES + LOWER, (box.max(axis=0) + 1) * RES + LOWER
    for row in b["planes"]:
        a = int(np.argmax(np.abs(row[:3])))
        assert np.isclose(-row[3] / row[a], hi[a] if row[a] > 0 else lo[a], atol=1e-12)
    properties(box, b)
    # not full-dimensional and not flat along an axis: the reference's cdd call fails; code 3 here
    diag = [[x, x, z] for x in range(5) for z in range(4)]
    assert hullapi.hull_planes(diag, RES, LOWER)["rc"] == 3 and hullemu.hull_planes(diag, RES, LOWER)["rc"] == 3
    assert hullapi.hull_planes(np.zeros((0, 3), np.int32), RES, LOWER)["rc"] == 3
    # capacity: reported, counts still exact
    g0 = golden()[0][0]
    full, cut = hullapi.hull_planes(g0, RES, LOWER), hullapi.hull_planes(g0, RES, LOWER, plane_cap=4)
    assert cut["rc"] == 1 and cut["n_planes"] == full["n_planes"]
    e = hullemu.hull_planes(g0, RES, LOWER, plane_cap=4)
    assert e["rc"] == 1 and e["n_planes"] == full["n_planes"] and np.array_equal(e["plane_int"], full["plane_int"][:4])
