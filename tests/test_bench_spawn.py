"""`python bench.py --gpus N` must start N ranks by itself (the driver launches it that way).  No GPU here, so
the --dry mode is used: the same self-spawn, torchrun rendezvous on 127.0.0.1 (gloo), shard generation and gather,
with nothing solved.  Sharded result == what one process computes over the whole stream."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(n, extra=("--config", "3", "--batch", "256", "--nseg", "4")):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry"] + list(extra),
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_self_spawns_two_ranks():
    sys.path.insert(0, ROOT)
    from direct_amd import problems
    two = run(2)
    assert two["dry"] and two["n_gpus"] == 2 and two["dist_world_size"] == 2 and two["value"] is None
    full = problems.make_batch("corridor", 512, 4, seed=1000)
    cost = full.T0.sum(axis=1)
    i = int(np.argmin(cost))
    assert two["gather"]["best_index"] == i and two["gather"]["best_cost"] == float(cost[i])
    assert two["gather"]["owner"] == i // 256
    want = float(np.concatenate([full.T0[i], full.seeds[i].ravel()]).sum())
    assert abs(two["gather"]["block_checksum"] - want) < 1e-9


def test_bench_eight_ranks_strong_scaling_split():
    """The launch the driver uses on an 8-GPU node, and --scaling strong: a fixed total split evenly, shard r =
    problems [r B/8, (r + 1) B/8) of the same stream - the gather's winner is the winner of the whole stream."""
    sys.path.insert(0, ROOT)
    from direct_amd import problems
    eight = run(8, ("--config", "3", "--nseg", "4", "--scaling", "strong", "--total", "2048"))
    assert eight["n_gpus"] == 8 and eight["dist_world_size"] == 8 and eight["scaling"] == "strong" and eight["batch_per_gpu"] == 256
    full = problems.make_batch("corridor", 2048, 4, seed=1000)
    cost = full.T0.sum(axis=1)
    i = int(np.argmin(cost))
    assert eight["gather"]["best_index"] == i and eight["gather"]["owner"] == i // 256
