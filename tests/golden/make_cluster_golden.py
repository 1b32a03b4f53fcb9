"""Generates tests/golden/cluster_*.npz: vectors for the corridor-cluster path produced with the REFERENCE'S OWN
serialConvexTest (oracle/_ref/libcluster_engine_ref.so, compiled from
/root/reference/polyhedron_generator/src/cluster_engine_cpu.cpp by `make -C oracle _ref`).

  cluster_convex_*.npz  one clustering round frozen right before its convex tests: map, inside flags, candidates,
                        cluster -> can_clu[i] = serialConvexTest(candidate i, whole cluster), can_can (packed lower
                        triangle) = serialConvexTest(candidate i, {candidate j}), accept = the sequential loop's
                        decisions.  Every value comes from the reference function.
  cluster_polygon_*.npz polygonGeneration for a few seeds: the loops around serialConvexTest are the restatement
                        of oracle/cluster_ref.c (cluster_server_cpu.cpp needs ROS headers and cannot be built),
                        run with the reference function plugged in.
Run from the repo root in the build container: python tests/golden/make_cluster_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from direct_amd import problems  # noqa: E402
from oracle import clusterapi as ca  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def convex_case(grid, seed, rounds):
    """State of the clustering loop of `seed` after `rounds` completed rounds, frozen before the next convex tests."""
    R = ca.ref_lib()
    dims = grid.shape
    v, surf, _, _ = ca.polygon_generation(grid, seed, itr_cluster_max=0)
    use, inside = ca.cube_state(grid, v)
    invalid = np.zeros(dims, np.uint8)
    cluster, active = surf.copy(), surf.copy()
    for r in range(rounds + 1):
        cand = ca.candidates(grid, use, invalid, inside, active)
        n = len(cand)
        can_clu = ca.serial_convex_test(R, "ref_serial_convex_test", cand, cluster, inside, grid, dims)
        can_can = np.zeros(n * (n - 1) // 2, np.uint8)
        for i in range(n):
            for j in range(i):
                can_can[i * (i - 1) // 2 + j] = ca.serial_convex_test(R, "ref_serial_convex_test", cand[i:i + 1], cand[j:j + 1],
                                                                       inside, grid, dims)[0]
        accept = ca.accept_sequential(can_clu, can_can)
        if r == rounds:
            return dict(grid=grid, inside=inside, cand=cand, cluster=cluster, can_clu=can_clu, can_can=can_can, accept=accept,
                        vertex_idx=v)
        for i in range(n):
            if not accept[i]:
                invalid[tuple(cand[i])] = 1
        cluster = np.concatenate([cluster, cand[accept == 1]])
        active = cand[accept == 1]


def main():
    assert ca.ref_lib() is not None, "build oracle/_ref first (make -C oracle _ref)"
    grid, seeds = problems.make_voxel_map((48, 48, 16), seed=11, n_pillars=22, n_boxes=10, n_rings=2)
    for name, seed, rounds in (("a", seeds[0], 0), ("b", seeds[3], 1)):
        c = convex_case(grid, seed, rounds)
        assert 0 < c["accept"].sum() < len(c["accept"]), (name, c["accept"].sum(), len(c["accept"]))
        np.savez_compressed(os.path.join(OUT, "cluster_convex_%s.npz" % name), seed=seed, **c)
        print(name, "candidates", len(c["cand"]), "cluster", len(c["cluster"]), "can_clu", int(c["can_clu"].sum()),
              "accept", int(c["accept"].sum()), "blocked pairs", int((c["can_can"] == 0).sum()))
    ca.use_reference_convex_test(True)
    sel = seeds[:12]
    vs, cls, its = [], [], []
    for s in sel:
        v, cl, it, rc = ca.polygon_generation(grid, s, itr_inflate_max=1000, itr_cluster_max=50)
        assert rc == 0
        vs.append(v); cls.append(cl); its.append(it)
    ca.use_reference_convex_test(False)
    num = np.array([len(c) for c in cls], np.int32)
    np.savez_compressed(os.path.join(OUT, "cluster_polygon_48.npz"), grid=grid, seeds=sel, vertex_idx=np.array(vs),
                        cluster_num=num, cluster_xyz=np.concatenate(cls), iters=np.array(its, np.int32))
    print("polygon: clusters", num.tolist(), "iters", its)


if __name__ == "__main__":
    main()
