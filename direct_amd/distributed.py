"""Multi-GPU sharding of a batch of independent corridors (BASELINE config 5).

The path shards trivially: problems are independent (the reference news a fresh optimiser per call,
teach_repeat_planner.cpp:853-854), so rank g solves the contiguous shard [g*B/G, (g+1)*B/G) with no
data-path collective.  The only exchange is the config-5 reduction "cheapest feasible trajectory of
the whole batch, materialised on every rank": one all_gather of each rank's local best
(cost, global index) and one all_gather of the local-best trajectory block -- tiny, latency-bound
messages over RCCL/xGMI (backend "nccl") or gloo in the CPU tests.  No reference counterpart.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import problems


def shard_range(total, rank, world, align=problems.CHUNK):
    """Contiguous, `align`-aligned shard of `total` problems for `rank` of `world`."""
    per = total // world
    assert per * world == total and per % align == 0, "batch must split into aligned equal shards"
    return rank * per, per


def local_best(cost, rtn):
    """Index and cost of the cheapest problem with rtn >= 0 (ties: lowest index); (-1, inf) if none."""
    cost = np.asarray(cost, np.float64)
    ok = np.asarray(rtn) >= 0
    if not ok.any():
        return -1, float("inf")
    c = np.where(ok, cost, np.inf)
    i = int(np.argmin(c))
    return i, float(c[i])


def gather_best(best_cost, best_global_index, block, device=None, force_collective=False):
    """All ranks learn the winner: returns (cost, global index, owner rank, block tensor).

    best_cost / best_global_index: this rank's local best; block: 1-D tensor with its trajectory
    (Bezier coefficients + durations).  One all_gather of 2 doubles and one of the block per rank.
    force_collective: run the two all_gathers at world size 1 as well (an initialised process group is needed): the
    N > 1 code path executed on a single GPU (bench.py, DIRECT_BENCH_FORCE_DIST=1).
    """
    world = dist.get_world_size() if dist.is_initialized() else 1
    device = block.device if device is None else device
    mine = torch.tensor([best_cost, float(best_global_index)], dtype=torch.float64, device=device)
    if world == 1 and not (force_collective and dist.is_initialized()):
        return best_cost, best_global_index, 0, block
    allv = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    blocks = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(blocks, block.contiguous())
    costs = torch.stack(allv)[:, 0].cpu().numpy()
    idxs = torch.stack(allv)[:, 1].cpu().numpy()
    valid = idxs >= 0
    if not valid.any():
        return float("inf"), -1, -1, block
    key = np.where(valid, costs, np.inf)
    # ties broken by the smaller global index so that the answer does not depend on the sharding
    owner = int(np.lexsort((idxs, key))[0])
    return float(costs[owner]), int(idxs[owner]), owner, blocks[owner]
