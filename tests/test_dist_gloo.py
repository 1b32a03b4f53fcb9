"""N > 1 path on CPU: world_size-2 gloo processes shard a batch, solve their shard (with the oracle
standing in for the device here) and run the config-5 gather; sharded == unsharded."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, n_seg, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from direct_amd import abi, distributed, problems
    from oracle import refapi
    first, count = distributed.shard_range(total, rank, world)
    batch = problems.make_batch("corridor", count, n_seg, seed=77, first=first)
    res, _ = refapi.solve_batch(abi.phase0_params(), batch, n_threads=1)
    i, c = distributed.local_best(res.cost, res.rtn)
    block = torch.from_numpy(np.concatenate([res.bez[i].ravel(), res.T[i].ravel()]))
    cost, gidx, owner, blk = distributed.gather_best(c, first + i, block)
    q.put((rank, cost, gidx, owner, blk.numpy().copy(), first, count))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_equals_unsharded():
    sys.path.insert(0, ROOT)
    from direct_amd import abi, distributed, problems
    from oracle import refapi
    total, n_seg, world = 2 * problems.CHUNK, 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, n_seg, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = problems.make_batch("corridor", total, n_seg, seed=77)
    res, _ = refapi.solve_batch(abi.phase0_params(), full)
    i, c = distributed.local_best(res.cost, res.rtn)
    want = np.concatenate([res.bez[i].ravel(), res.T[i].ravel()])
    for rank, cost, gidx, owner, blk, first, count in outs:
        assert gidx == i and cost == c, (gidx, i, cost, c)
        assert owner == i // (total // world)
        assert np.array_equal(blk, want)
    assert sorted(o[5] for o in outs) == [0, total // world]


def test_shard_generation_is_position_independent():
    sys.path.insert(0, ROOT)
    from direct_amd import problems
    full = problems.make_batch("corridor", 3 * problems.CHUNK, 4, seed=5)
    part = problems.make_batch("corridor", problems.CHUNK, 4, seed=5, first=2 * problems.CHUNK)
    assert np.array_equal(full.planes[2 * problems.CHUNK:], part.planes)
    assert np.array_equal(full.T0[2 * problems.CHUNK:], part.T0)
    assert np.array_equal(full.n_planes[2 * problems.CHUNK:], part.n_planes)
