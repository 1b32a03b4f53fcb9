"""Writes direct_amd/data/real_corridor_n12.npz: ONE replay plan of the real-corridor chain (voxel map -> grid path ->
corridorGeneration on the device -> first n polytopes, tests/real_corridor_lib.py) with N = 12 segments - the single plan
bench.py's `single_plan_latency` block solves (the reference's caller plans one corridor per click,
teach_repeat_planner.cpp:895-921, N ~ 3 - 15).  Needs the GPU (the cluster path): run through gpurun from the repo root;
the file lands in gpurun_out/ and is copied into direct_amd/data/ by hand.
usage: python tools/make_real_corridor_fixture.py [n_seg] [out.npz]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests import real_corridor_lib

n_want = int(sys.argv[1]) if len(sys.argv) > 1 else 12
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/real_corridor_n12.npz"
batch, meta = real_corridor_lib.real_corridor_batch(64)
cand = np.flatnonzero(batch.n_seg == n_want)
assert len(cand), "no replay plan with %d segments" % n_want
# the widest of them: the row-slot class a lone plan of the pipeline typically lands in (median widest polytope 32 planes)
pm = np.array([batch.n_planes[i, :n_want].max() for i in cand])
i = int(cand[np.argsort(pm)[len(pm) // 2]])
N, P = n_want, int(batch.n_planes[i, :n_want].max())
np.savez_compressed(out, n_seg=batch.n_seg[i:i + 1], x0=batch.x0[i:i + 1], xd=batch.xd[i:i + 1], T0=batch.T0[i:i + 1, :N],
                    n_planes=batch.n_planes[i:i + 1, :N], planes=batch.planes[i:i + 1, :N, :P], seeds=batch.seeds[i:i + 1, :N],
                    source=np.array("tests/real_corridor_lib.real_corridor_batch(64): plan %d of %d, %d segments, widest polytope %d planes"
                                    % (i, batch.batch, N, P)))
print("wrote", out, "plan", i, "N", N, "P", P)
