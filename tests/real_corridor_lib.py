"""The DDP path fed what the reference's own pipeline would feed it - TEST INFRASTRUCTURE shared by
tests/test_gpu_real_corridors.py and tests/soak/real_corridor_bench.py: voxel map -> grid paths -> corridorGeneration
(direct_amd/host/poly_utils.hpp, device path) -> replay protocol of corridorRecCallBack (first n polytopes, n = 2 ..;
teach_repeat_planner.cpp:316-320) -> ONE ragged batch of two-phase plans."""
import os
import struct
import subprocess
import tempfile
from collections import deque

import numpy as np

from direct_amd import abi, corridor_io, problems

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES, LOWER = 0.2, np.array([-20.0, -20.0, 0.0])


def grid_path(grid, start, goal):
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
    if not seen[goal[0], goal[1]]:
        return None
    path, cur = [], (int(goal[0]), int(goal[1]))
    while cur != (int(start[0]), int(start[1])):
        path.append(cur)
        cur = tuple(int(c) for c in prev[cur])
    path.append(cur)
    return np.array([[x, y, z] for x, y in path[::-1]], np.float64) * RES + 0.5 * RES + LOWER


def concat(batches):
    nm = max(b.n_seg_max for b in batches)
    pm = max(b.p_max for b in batches)
    B = sum(b.batch for b in batches)
    n_seg = np.concatenate([b.n_seg for b in batches])
    x0, xd = np.concatenate([b.x0 for b in batches]), np.concatenate([b.xd for b in batches])
    T0, npl = np.zeros((B, nm)), np.zeros((B, nm), np.int32)
    planes, seeds = np.zeros((B, nm, pm, 4)), np.zeros((B, nm, 3))
    o = 0
    for b in batches:
        T0[o:o + b.batch, :b.n_seg_max] = b.T0
        npl[o:o + b.batch, :b.n_seg_max] = b.n_planes
        planes[o:o + b.batch, :b.n_seg_max, :b.p_max] = b.planes
        seeds[o:o + b.batch, :b.n_seg_max] = b.seeds
        o += b.batch
    npl[npl == 0] = 1   # unused knots: a neutral count (never read beyond n_seg)
    return abi.HostBatch(n_seg, x0, xd, T0, npl, planes, seeds=seeds)



def real_corridor_batch(n_paths=64, dims=(200, 200, 40)):
    """-> (HostBatch of the replay plans of all corridors, dict of facts about the generation)"""
    grid, _ = problems.make_voxel_map(dims, seed=7, n_pillars=170, n_boxes=70, n_rings=12)
    rng = np.random.default_rng(5)
    paths = []
    while len(paths) < n_paths:
        z = int(rng.integers(4, 30))
        free = np.argwhere(grid[:, :, z] == 0)
        a, b = free[rng.integers(len(free))], free[rng.integers(len(free))]
        if np.abs(a - b).sum() > 120:
            p = grid_path(grid, [a[0], a[1], z], [b[0], b[1], z])
            if p is not None:
                paths.append(p)
    tmp = tempfile.mkdtemp()
    fin, fout, exe = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin"), os.path.join(tmp, "gen")
    with open(fin, "wb") as f:
        f.write(struct.pack("<3id3di", *grid.shape, RES, *LOWER, len(paths)))
        for p in paths:
            f.write(struct.pack("<i", len(p)))
            f.write(np.ascontiguousarray(p, np.float64).tobytes())
        f.write(np.ascontiguousarray(grid, np.uint8).tobytes())
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests/cpp/test_corridor_gen.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "direct_amd/lib"), "-ldirect_ddp",
                           "-Wl,-rpath," + os.path.join(ROOT, "direct_amd/lib") + ":/opt/rocm/lib"])
    gen_out = subprocess.run([exe, fin, fout, "64"], capture_output=True, text=True)
    assert gen_out.returncode == 0, gen_out.stdout + gen_out.stderr
    raw, off = open(fout, "rb").read(), 0


    def take(fmt):
        nonlocal off
        v = struct.unpack_from(fmt, raw, off)
        off += struct.calcsize(fmt)
        return v


    modes = []
    for mode in range(2):
        cs = []
        for p in range(len(paths)):
            ok, n = take("<2i")
            cor = []
            for _ in range(n):
                (k,) = take("<i")
                cor.append((np.array(take("<%dd" % (4 * k))).reshape(k, 4), np.array(take("<3d")), np.array(take("<3d"))))
            cs.append((cor, ok))
        modes.append(cs)
    same = all(len(a[0]) == len(b[0]) and all(np.array_equal(x[0], y[0]) for x, y in zip(a[0], b[0])) for a, b in zip(*modes))
    all_cors = [c for c, ok in modes[1] if ok and len(c) >= 2]
    cors = [c for c in all_cors if max(len(q[0]) for q in c) <= abi.P_LIMIT]   # the DDP kernels take up to DIRECT_P_LIMIT planes per polytope
    batches = []
    for c in cors:
        pm = max(len(q[0]) for q in c)
        pl = np.zeros((len(c), pm, 4))
        for i, q in enumerate(c):
            pl[i, :len(q[0])] = q[0]
        cor = corridor_io.Corridor(0, [len(q[0]) for q in c], pl, [q[2] for q in c], [q[1] for q in c])
        batches.append(corridor_io.replay_batch(cor, n_first=2))
    batch = concat(batches)
    meta = {"map": dims, "paths": len(paths), "path_voxels_mean": float(np.mean([len(p) for p in paths])),
            "corridor_generation": {"stdout": gen_out.stdout.strip().splitlines(), "one_by_one_equals_lock_step": bool(same),
                                    "polytopes_per_corridor_mean": float(np.mean([len(c) for c in cors])),
                                    "planes_per_polytope_mean": float(np.mean([len(q[0]) for c in cors for q in c])),
                                    "planes_per_polytope_max": int(max(len(q[0]) for c in all_cors for q in c)),
                                    "planes_per_polytope_quantiles_50_90_99": [float(v) for v in np.quantile([len(q[0]) for c in all_cors for q in c], [0.5, 0.9, 0.99])],
                                    "widest_polytope_per_corridor_quantiles_50_90_max": [float(v) for v in np.quantile([max(len(q[0]) for q in c) for c in all_cors], [0.5, 0.9, 1.0])],
                                    "corridors": len(all_cors), "corridors_with_a_polytope_above_P_LIMIT": len(all_cors) - len(cors)},
            "ddp_problems": int(batch.batch), "n_seg_max": int(batch.n_seg_max), "p_max": int(batch.p_max)}
    return batch, meta
