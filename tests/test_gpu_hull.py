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


def grid_path(grid, start, goal):
    """a 4-connected shortest voxel path at the height of `start` (breadth first), as voxel-centre coordinates"""
    from collections import deque
    z = start[2]
    free = grid[:, :, z] == 0
    prev = -np.ones(free.shape + (2,), np.int32)
    seen = np.zeros(free.shape, bool)
    dq = deque([(int(start[0]), int(start[1]))])
    seen[start[0], start[1]] = True
    while dq:
        x, y = dq.popleft()
        if (x, y) == (int(goal[0]), int(goal[1])):
            break
        for dx, dy in ((1, 0), (-1, 0), (0, 1), (0, -1)):
            u, v = x + dx, y + dy
            if 0 <= u < free.shape[0] and 0 <= v < free.shape[1] and free[u, v] and not seen[u, v]:
                seen[u, v] = True
                prev[u, v] = (x, y)
                dq.append((u, v))
    assert seen[goal[0], goal[1]]
    path, cur = [], (int(goal[0]), int(goal[1]))
    while cur != (int(start[0]), int(start[1])):
        path.append(cur)
        cur = tuple(int(c) for c in prev[cur])
    path.append(cur)
    return np.array([[x, y, z] for x, y in path[::-1]], np.float64) * RES + 0.5 * RES + LOWER


def test_corridor_generation_walk_and_ddp(built, tmp_path):
    """corridorGeneration (poly_utils.cpp:508-557) and corridorInsertGeneration (:391-449, the live caller's walk,
    teach_repeat_planner.cpp:172 / 228) through the C++ host class (direct_amd/host/poly_utils.hpp): grid
    paths -> corridors, path by path and all paths in lock step, against the walk restated over the CPU checkers
    (planes bit for bit); then the corridor is what the DDP path consumes: the replay batch of one of them is solved."""
    import struct
    import subprocess
    from direct_amd import corridor_io
    from oracle import corridor_walk
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    grid, _ = problems.make_voxel_map()
    free = np.argwhere(grid[:, :, 8] == 0)
    rng = np.random.default_rng(4)
    paths = []
    while len(paths) < 5:
        a, b = free[rng.integers(len(free))], free[rng.integers(len(free))]
        if np.abs(a - b).sum() > 60:
            try:
                paths.append(grid_path(grid, [a[0], a[1], 8], [b[0], b[1], 8]))
            except AssertionError:
                pass
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<3id3di", *grid.shape, RES, *LOWER, len(paths)))
        for p in paths:
            f.write(struct.pack("<i", len(p)))
            f.write(np.ascontiguousarray(p, np.float64).tobytes())
        f.write(np.ascontiguousarray(grid, np.uint8).tobytes())
    exe = str(tmp_path / "test_corridor_gen")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(root, "tests/cpp/test_corridor_gen.cpp"), "-o", exe,
                           "-L" + os.path.join(root, "direct_amd/lib"), "-ldirect_ddp",
                           "-Wl,-rpath," + os.path.join(root, "direct_amd/lib") + ":/opt/rocm/lib"])
    out = subprocess.run([exe, fin, fout], capture_output=True, text=True)
    assert out.returncode == 0 and "PASS" in out.stdout, out.stdout + out.stderr
    raw = open(fout, "rb").read()
    off = 0

    def take(fmt):
        nonlocal off
        v = struct.unpack_from(fmt, raw, off)
        off += struct.calcsize(fmt)
        return v
    cache, modes = {}, []
    want = [corridor_walk.corridor_generation(grid, RES, LOWER, p, cache=cache) for p in paths]
    want_ins = []  # corridorInsertGeneration (poly_utils.cpp:391-449): first half into an empty corridor, second half into that
    for p in paths:
        c1, r1 = corridor_walk.corridor_insert_generation(grid, RES, LOWER, p[:len(p) // 2], [], cache=cache)
        c2, r2 = corridor_walk.corridor_insert_generation(grid, RES, LOWER, p[len(p) // 2:], c1, cache=cache)
        want_ins.append((c2, r1 == 1 and r2 == 1))
    for mode in range(4):
        got = []
        for p in range(len(paths)):
            ok, n = take("<2i")
            cor = []
            for _ in range(n):
                (k,) = take("<i")
                pl = np.array(take("<%dd" % (4 * k))).reshape(k, 4)
                cor.append(dict(planes=pl, center=np.array(take("<3d")), seed_coord=np.array(take("<3d"))))
            got.append((cor, bool(ok)))
        modes.append(got)
    assert off == len(raw)
    n_poly = 0
    # the insert walk never pops: on these paths it keeps polytopes the plain walk drops again
    assert sum(len(c) for c, _ in want_ins) >= sum(len(c) for c, _ in want)
    for mi, got in enumerate(modes):
        for (gc, gok), (wc, wok) in zip(got, want if mi < 2 else want_ins):
            assert gok == wok and len(gc) == len(wc) and len(gc) >= 2
            for a, b in zip(gc, wc):
                assert np.array_equal(a["planes"], b["planes"]) and np.array_equal(a["center"], b["center"])
                assert np.array_equal(a["seed_coord"], b["seed_coord"])
            n_poly += len(gc)
    # the corridor feeds the DDP path: wire format -> replay batch (first n polytopes, n = 2 ..) -> two-phase plan
    cor = max((c for c, _ in modes[1] if max(len(q["planes"]) for q in c) <= abi.P_LIMIT), key=len)   # P_LIMIT 54: all of them
    assert len(cor) >= 3
    pm = max(len(q["planes"]) for q in cor)
    planes = np.zeros((len(cor), pm, 4))
    for i, q in enumerate(cor):
        planes[i, :len(q["planes"])] = q["planes"]
    c = corridor_io.Corridor(7, [len(q["planes"]) for q in cor], planes, [q["seed_coord"] for q in cor], [q["center"] for q in cor])
    c2, _ = corridor_io.unpack(corridor_io.pack(c), 64, pm)
    assert np.array_equal(c2.planes, c.planes)
    batch = corridor_io.replay_batch(c2, n_first=2)
    s = solver.DdpSolver(batch.batch, int(batch.n_seg.max()), pm, np.float64)
    g0, g1 = s.plan(abi.phase0_params(), abi.phase1_params(), batch)
    d = s.sample(batch.n_seg, g1.bez, g1.T, 0.05, 4096, derivs=0, n_planes=batch.n_planes, planes=batch.planes)
    s.close()
    assert (g1.rtn >= 0).any()
    assert (d["cmax"][g1.rtn >= 0] < 1e-6).all()       # solved trajectories stay inside their polytopes


def test_hull_input_guards(built):
    """(a) a caller-provided voxel outside the map disqualifies ITS cluster (DIRECT_HULL_BAD_VOXEL) instead of
    addressing lattice lines outside the per-cluster arrays, the other clusters of the call are unaffected; (b) the
    resident-cluster form needs a generation of at least `batch` seeds, and caller-provided voxels - which are packed
    into the handle's cluster storage - retire the resident clusters; (c) a generation that overflowed the cluster
    capacity in a clustering round keeps the valid PREFIX of the cluster (every returned voxel is a map voxel of the
    oracle's cluster, in its order) and its hull is refused (DIRECT_HULL_OVERFLOW) rather than computed on slots that
    were never written."""
    from direct_amd.solver import DirectError
    grid, seeds = problems.make_voxel_map()
    gen = cluster.ClusterGenerator(grid.shape, max_batch=8, cluster_capacity=20000, candidate_capacity=4000)
    gen.set_map(grid)
    r = gen.polygon_generation(seeds[:3])
    assert (r["rtn"] == cluster.CLUSTER_OK).all()
    good = gen.hull_planes(RES, LOWER, batch=3)
    assert (good["rtn"] == cluster.HULL_OK).all()
    # (b) more clusters than the last generation produced
    with pytest.raises(DirectError) as e:
        gen.hull_planes(RES, LOWER, batch=4)
    assert e.value.status == abi.DIRECT_ERR_INVALID
    # (a) one voxel of cluster 1 outside the map, in every direction the packing could alias
    for badv in ([-1, 3, 3], [grid.shape[0], 2, 2], [3, 3, 1024 + 5], [2, grid.shape[1] + 7, 1]):
        cl = [c.copy() for c in r["clusters"]]
        cl[1][len(cl[1]) // 2] = badv
        h = gen.hull_planes(RES, LOWER, clusters=cl)
        assert h["rtn"][1] == cluster.HULL_BAD_VOXEL and h["n_planes"][1] == 0
        for b in (0, 2):
            assert h["rtn"][b] == cluster.HULL_OK and np.array_equal(h["planes"][b], good["planes"][b])
    # (b) the caller's voxels replaced the resident clusters: a resident call needs a new generation
    with pytest.raises(DirectError):
        gen.hull_planes(RES, LOWER, batch=3)
    gen.close()
    # (c) overflow in a clustering round: capacity just above the inflated cube's surface
    _, cl_full, _, _ = clusterapi.polygon_generation(grid, tuple(int(v) for v in seeds[0]))
    gen = cluster.ClusterGenerator(grid.shape, max_batch=2, cluster_capacity=len(cl_full) - 5, candidate_capacity=4000)
    gen.set_map(grid)
    ro = gen.polygon_generation(seeds[:1])
    assert ro["rtn"][0] == cluster.CLUSTER_OVERFLOW
    n = int(ro["cluster_num"][0])
    assert 0 < n <= len(cl_full) - 5
    assert np.array_equal(ro["clusters"][0], cl_full[:n])          # a valid prefix, never stale storage
    ho = gen.hull_planes(RES, LOWER, batch=1)
    assert ho["rtn"][0] == cluster.HULL_OVERFLOW and ho["n_planes"][0] == 0
    gen.close()
