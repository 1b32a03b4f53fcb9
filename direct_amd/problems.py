"""Synthetic corridor batches for the configurations of BASELINE.json / SURVEY.md section 8(d).

The reference has no corridor fixtures; these generators mimic what its
callers feed polyCurveGeneration: seeds a few metres apart (poly_utils.cpp:413-431), polytopes as
sign-normalised half-spaces with outward normals (poly_utils.cpp:42-52, 146-166), start/goal at rest
(teach_repeat_planner.cpp:832-842), durations from initTimeAllocation (teach_repeat_planner.cpp:583-639)
and, for "no corridor", the authors' own neutral-plane trick (0,0,0,-1)
(teach_repeat_planner.cpp:866-879).

Random numbers: numpy PCG64, one stream per chunk of CHUNK problems keyed by (seed, chunk index), so
that any CHUNK-aligned shard of a batch is generated identically on every rank (config 5).
"""
import numpy as np

from . import abi

CHUNK = 256


def time_allocation(n_seg, start, goal, seeds, max_vel=2.0, max_acc=2.0):
    """initTimeAllocation (teach_repeat_planner.cpp:583-639) with v0 = 0, vectorised in numpy.

    points = [start, seeds[1..N-1], goal]; trapezoid profile per consecutive pair.
    """
    n_seg = np.asarray(n_seg)
    B, nmax = seeds.shape[0], seeds.shape[1]
    pts = np.zeros((B, nmax + 1, 3))
    pts[:, :nmax] = seeds
    pts[:, 0] = start
    for b in range(B):  # goal sits at index n_seg[b]
        pts[b, n_seg[b]] = goal[b]
    D = np.linalg.norm(pts[:, 1:] - pts[:, :-1], axis=2)
    acct = max_vel / max_acc
    accd = max_acc * acct * acct / 2
    dcct = max_vel / max_acc
    dccd = max_acc * dcct * dcct / 2
    with np.errstate(invalid="ignore"):
        t_short = 2.0 * np.sqrt(max_acc * D) / max_acc        # t2 + t3 with aV0 = 0
    t_long = acct + (D - accd - dccd) / max_vel + dcct
    T = np.where(D < accd + dccd, t_short, t_long)
    k = np.arange(nmax)[None, :]
    return np.where(k < n_seg[:, None], T, 0.0)


def neutral_planes(B, N, P=6):
    """Every plane (0,0,0,-1): position rows evaluate to -1, always satisfied (TRP:875-876)."""
    pl = np.zeros((B, N, P, 4))
    pl[..., 3] = -1.0
    return pl


def _rng(seed, chunk):
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence([int(seed), int(chunk)])))


def _free_space_chunk(rng, n, N):
    start = np.stack([rng.uniform(-14, 14, n), rng.uniform(-14, 14, n), rng.uniform(0.5, 2.5, n)], axis=1)
    ell = rng.uniform(1.5, 3.5, n)
    th = rng.uniform(0.0, 2 * np.pi, n)
    ph = rng.uniform(-0.05, 0.05, n)
    direc = np.stack([np.cos(th) * np.cos(ph), np.sin(th) * np.cos(ph), np.sin(ph)], axis=1)
    goal = start + (N * ell)[:, None] * direc
    k = np.arange(N)[None, :, None] / float(N)
    seeds = start[:, None, :] + k * (goal - start)[:, None, :]
    jit = rng.normal(0.0, 0.3, (n, N, 3))
    jit[:, 0] = 0.0
    seeds = seeds + jit
    return start, goal, seeds


def _corridor_chunk(rng, start, goal, seeds, p_max):
    """Config-3 polytopes: inflated AABB of seeds k, k+1 plus 0..6 random cutting planes."""
    n, N = seeds.shape[0], seeds.shape[1]
    pts = np.concatenate([seeds, goal[:, None, :]], axis=1)
    pts[:, 0] = start
    a, b = pts[:, :-1], pts[:, 1:]
    lo = np.minimum(a, b) - rng.uniform(0.6, 1.5, (n, N, 3))
    hi = np.maximum(a, b) + rng.uniform(0.6, 1.5, (n, N, 3))
    planes = np.zeros((n, N, p_max, 4))
    for d in range(3):
        planes[:, :, 2 * d, d] = 1.0
        planes[:, :, 2 * d, 3] = -hi[..., d]
        planes[:, :, 2 * d + 1, d] = -1.0
        planes[:, :, 2 * d + 1, 3] = lo[..., d]
    n_extra = rng.integers(0, p_max - 6 + 1, (n, N))
    nrm = rng.normal(size=(n, N, p_max - 6, 3))
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    clear = rng.uniform(0.3, 1.2, (n, N, p_max - 6))
    off = -np.maximum(np.einsum("bkpd,bkd->bkp", nrm, a), np.einsum("bkpd,bkd->bkp", nrm, b)) - clear
    use = np.arange(p_max - 6)[None, None, :] < n_extra[..., None]
    planes[:, :, 6:, :3] = np.where(use[..., None], nrm, 0.0)
    planes[:, :, 6:, 3] = np.where(use, off, 0.0)
    return planes, (6 + n_extra).astype(np.int32)


def make_batch(kind, batch, n_seg, seed=1000, first=0, p_max=None, dtype=np.float64,
               max_vel=2.0, max_acc=2.0):
    """kind: 'free' (config 2: neutral planes, P = 6) or 'corridor' (configs 3-5: P in [6,12]).

    Problems first .. first+batch-1 of the stream `seed`; `first` must be CHUNK-aligned.
    Returns an abi.HostBatch (durations from time_allocation, seeds included).
    """
    assert first % CHUNK == 0, "shards must be aligned to %d problems" % CHUNK
    if p_max is None:
        p_max = 6 if kind == "free" else 12
    starts, goals, seedss, planess, nps = [], [], [], [], []
    done = 0
    while done < batch:
        n = min(CHUNK, batch - done)
        rng = _rng(seed, (first + done) // CHUNK)
        st, gl, sd = _free_space_chunk(rng, CHUNK, n_seg)
        if kind == "corridor":
            pl, npl = _corridor_chunk(rng, st, gl, sd, p_max)
        else:
            pl, npl = neutral_planes(CHUNK, n_seg, p_max), np.full((CHUNK, n_seg), p_max, np.int32)
        starts.append(st[:n]); goals.append(gl[:n]); seedss.append(sd[:n])
        planess.append(pl[:n]); nps.append(npl[:n])
        done += n
    start, goal = np.concatenate(starts), np.concatenate(goals)
    seeds, planes, n_planes = np.concatenate(seedss), np.concatenate(planess), np.concatenate(nps)
    nseg = np.full(batch, n_seg, np.int32)
    T0 = time_allocation(nseg, start, goal, seeds, max_vel, max_acc)
    x0 = np.zeros((batch, 9)); x0[:, :3] = start
    xd = np.zeros((batch, 9)); xd[:, :3] = goal
    return abi.HostBatch(nseg, x0, xd, T0, n_planes, planes, seeds=seeds, dtype=dtype)


def make_config1(seed=1, n_seg=50, dtype=np.float64):
    """Config 1: one trajectory through 8 waypoints, N = 50, no corridor (neutral planes).

    W0 = (-8,-8,2) (launch init, global_planner.launch:7-9), then the ring walk of
    teach_repeat_planner.cpp:196-217 (radius U(15,25), dtheta U(0.5,2.5) from 1.25 pi, z U(0.5,1.8)).
    """
    rng = _rng(seed, 0)
    W = [np.array([-8.0, -8.0, 2.0])]
    th = 1.25 * np.pi
    for _ in range(7):
        th = th + rng.uniform(0.5, 2.5)
        rad = rng.uniform(15.0, 25.0)
        W.append(np.array([rad * np.cos(th), rad * np.sin(th), rng.uniform(0.5, 1.8)]))
    W = np.array(W)
    seg = np.linalg.norm(W[1:] - W[:-1], axis=1)
    cum = np.concatenate([[0.0], np.cumsum(seg)])
    s = np.linspace(0.0, cum[-1], n_seg + 1)
    pts = np.stack([np.interp(s, cum, W[:, d]) for d in range(3)], axis=1)
    seeds = pts[None, :n_seg].copy()
    start, goal = W[None, 0], W[None, -1]
    nseg = np.array([n_seg], np.int32)
    T0 = time_allocation(nseg, start, goal, seeds)
    x0 = np.zeros((1, 9)); x0[0, :3] = W[0]
    xd = np.zeros((1, 9)); xd[0, :3] = W[-1]
    return abi.HostBatch(nseg, x0, xd, T0, np.full((1, n_seg), 6, np.int32), neutral_planes(1, n_seg, 6),
                         seeds=seeds, dtype=dtype)


def algorithmic_words(n_planes, n_seg, infeasible=False, split=False):
    """Algorithmic words moved per DDP iteration of a batch:
    per knot 257 + 5 nc + 8 P (feasible) or 257 + 10 nc + 8 P (infeasible), nc = 6 P + 55.
    `infeasible`: one flag for the batch or one per trajectory (the mode its iterations run in).
    `split`: return (backward words, forward words) instead of their sum - SURVEY's own split of the figure:
    backward R(nx + nu + nc + 4 P) W(nu + nu nx + nc) = 119 + 2 nc + 4 P, forward (ONE trial)
    R(nx + nu + nc + 4 P + nu + nu nx + nc) W(nx + nu + nc) = 138 + 3 nc + 4 P; the nc terms double in infeasible mode."""
    n_planes = np.asarray(n_planes)
    k = np.arange(n_planes.shape[1])[None, :] < np.asarray(n_seg)[:, None]
    nc = 6 * n_planes + 55
    dbl = np.where(np.broadcast_to(np.asarray(infeasible, bool).reshape(-1, 1), n_planes.shape), 2, 1)
    wb = 119 + 2 * dbl * nc + 4 * n_planes
    wf = 138 + 3 * dbl * nc + 4 * n_planes
    if split:
        return int(np.where(k, wb, 0).sum()), int(np.where(k, wf, 0).sum())
    return int(np.where(k, wb + wf, 0).sum())


def make_voxel_map(dims=(120, 120, 24), seed=7, n_pillars=60, n_boxes=25, n_rings=6):
    """Synthetic occupancy grid in the spirit of the reference's random map publisher (pillars and rings of
    global_planner/src/utils/random_complex_generator.cpp:40-150; 0 = free, 1 = obstacle, index [x][y][z]).
    Returns the uint8 grid and a list of free seed voxels spread over the map."""
    rng = _rng(seed, 0)
    X, Y, Z = dims
    g = np.zeros(dims, np.uint8)
    for _ in range(n_pillars):
        x, y = int(rng.integers(2, X - 4)), int(rng.integers(2, Y - 4))
        w, hgt = int(rng.integers(1, 4)), int(rng.integers(Z // 3, Z))
        g[x:x + w, y:y + w, :hgt] = 1
    for _ in range(n_boxes):
        x, y, z = int(rng.integers(0, X - 8)), int(rng.integers(0, Y - 8)), int(rng.integers(0, Z - 3))
        g[x:x + int(rng.integers(2, 8)), y:y + int(rng.integers(2, 8)), z:z + int(rng.integers(1, 4))] = 1
    for _ in range(n_rings):  # vertical rings: obstacles one voxel thick around a free hole
        x, y0, z0 = int(rng.integers(4, X - 4)), int(rng.integers(2, Y - 12)), int(rng.integers(1, max(2, Z - 10)))
        r = int(rng.integers(3, 6))
        yy, zz = np.meshgrid(np.arange(Y), np.arange(Z), indexing="ij")
        d = np.maximum(np.abs(yy - (y0 + r)), np.abs(zz - (z0 + r)))
        g[x][(d == r)] = 1
    free = np.argwhere(g == 0)
    seeds = free[rng.choice(len(free), 256, replace=False)].astype(np.int32)
    return g, seeds
