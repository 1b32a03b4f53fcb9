"""corridorGeneration (global_planner/src/utils/poly_utils.cpp:508-557) restated on top of the cluster and hull oracles
-- TEST INFRASTRUCTURE.  The walk along a grid path: snap to the voxel centre (:3-40), skip repeats, pop the last
polytope when the path is back inside the last but one (:526-530), ask for a new polytope when it leaves the latest
(isOutsidePolytope's margin 0.01, :42-52)."""
import numpy as np

from . import clusterapi, hullapi


def corridor_generation(grid, res, lower, path, itr_inflate_max=1000, itr_cluster_max=50, cache=None):
    """-> (list of dict(planes [P][4], center [3], seed_coord [3]), ok)"""
    lower = np.asarray(lower, np.float64)
    dims = np.array(grid.shape)
    cor, lst = [], None
    def outside(c, p):  # the reference's left-to-right sum, plane by plane
        q = p["planes"]
        return bool((q[:, 0] * c[0] + q[:, 1] * c[1] + q[:, 2] * c[2] + q[:, 3] > 0.01).any())
    for pt in path:
        idx = np.clip(((np.asarray(pt, np.float64) - lower) * (1.0 / res)).astype(np.int64), 0, dims - 1)
        cur = idx * res + 0.5 * res + lower
        if lst is not None and (cur == lst).all():
            continue
        if len(cor) > 1 and not outside(cur, cor[-2]):
            cor.pop()
        if not cor or outside(cur, cor[-1]):
            key = tuple(int(i) for i in idx)
            if cache is None or key not in cache:
                cl = clusterapi.polygon_generation(grid, key, itr_inflate_max, itr_cluster_max)[1]
                r = hullapi.hull_planes(cl, res, lower)
                if cache is not None:
                    cache[key] = r
            else:
                r = cache[key]
            if r["rc"] != 0:
                return cor, False
            cor.append(dict(planes=r["planes"], center=r["center"].copy(), seed_coord=cur.copy()))
        lst = cur
    return cor, True
