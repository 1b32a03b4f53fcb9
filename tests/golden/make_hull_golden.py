"""Golden vectors for the hull -> planes step from the REFERENCE'S OWN convex hull (global_planner/third_party/quickhull,
built from its sources into oracle/_ref/libquickhull_ref.so): for the 12 clusters of cluster_polygon_48.npz and four
flat ones, the vertex buffer and the triangles getConvexHull(points, true, false) returns for the point set
getConvexPoly builds (poly_utils.cpp:301-389).  Run in the build container (needs /root/reference):
    python tests/golden/make_hull_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import hullapi  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def flat_clusters():
    return [np.array([[4, 4, 5]]), np.array([[x, y, 5] for x in range(3, 9) for y in range(4, 7)]),
            np.array([[3, y, z] for y in range(2, 5) for z in range(1, 7) if y + z != 3]),
            np.array([[x, 4, 5] for x in range(2, 7)])]


def main():
    hullapi.build()
    g = np.load(os.path.join(HERE, "cluster_polygon_48.npz"))
    off = np.concatenate([[0], np.cumsum(g["cluster_num"])])
    clusters = [g["cluster_xyz"][off[i]:off[i + 1]] for i in range(len(g["cluster_num"]))] + flat_clusters()
    out = dict(n=len(clusters))
    for i, cl in enumerate(clusters):
        deg = int(any(len(set(cl[:, a])) == 1 for a in range(3)))
        pts = hullapi.lattice_points(cl, deg)
        vb, tri = hullapi.reference_quickhull(pts.astype(np.float64))
        vbq = np.rint(vb).astype(np.int32)
        assert np.abs(vb - vbq).max() == 0
        out["cluster_%d" % i] = cl.astype(np.int32)
        out["deg_%d" % i] = deg
        out["qh_vertices_%d" % i] = vbq
        out["qh_triangles_%d" % i] = tri.astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "hull_quickhull_16.npz"), **out)
    print("wrote", len(clusters), "clusters")


if __name__ == "__main__":
    main()
