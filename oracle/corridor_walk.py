"""corridorGeneration (global_planner/src/utils/poly_utils.cpp:508-557) and corridorInsertGeneration (:391-449) restated on top of the cluster and hull oracles
-- TEST INFRASTRUCTURE.  The walk along a grid path: snap to the voxel centre (:3-40), skip repeats, pop the last
polytope when the path is back inside the last but one (:526-530), ask for a new polytope when it leaves the latest
(isOutsidePolytope's margin 0.01, :42-52)."""
import numpy as np

from . import clusterapi, hullapi


def corridor_insert_generation(grid, res, lower, path, corridor, itr_inflate_max=1000, itr_cluster_max=50, cache=None):
    """corridorInsertGeneration (poly_utils.cpp:391-449), the walk of the live caller (teach_repeat_planner.cpp:172,
    228): extends a COPY of `corridor` (list of dicts as corridor_generation returns them); a point that is outside the
    latest polytope gets a new one, there is no pop.  -> (corridor, 1) on success, (the corridor as given, 0) where the
    reference's cdd call fails."""
    cor, ok = corridor_generation(grid, res, lower, path, itr_inflate_max, itr_cluster_max, cache, start=list(corridor), pop=False)
    return (cor, 1) if ok else (list(corridor), 0)


def corridor_generation(grid, res, lower, path, itr_inflate_max=1000, itr_cluster_max=50, cache=None, start=None, pop=True):
    """-> (list of dict(planes [P][4], center [3], seed_coord [3]), ok)"""
    lower = np.asarray(lower, np.float64)
    dims = np.array(grid.shape)
    cor, lst = ([] if start is None else start), None
    def outside(c, p):  # the reference's left-to-right sum, plane by plane
        q = p["planes"]
        return bool((q[:, 0] * c[0] + q[:, 1] * c[1] + q[:, 2] * c[2] + q[:, 3] > 0.01).any())
    for pt in path:
        idx = np.clip(((np.asarray(pt, np.float64) - lower) * (1.0 / res)).astype(np.int64), 0, dims - 1)
        cur = idx * res + 0.5 * res + lower
        if lst is not None and (cur == lst).all():
            continue
        if pop and len(cor) > 1 and not outside(cur, cor[-2]):
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
