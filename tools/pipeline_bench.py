"""Back-to-back independent batches on TWO handles / TWO streams against one handle (the reference contract: independent
optimiser objects per call, TRP:853-854).  The hot kernel is a grid of persistent waves: the second handle's waves become
resident as the first's leave, so launch i + 1's first epochs fill the chip that launch i's slowest chains leave idle.
usage: pipeline_bench.py [B] [R] [yield_k] [H]   (yield_k: waves kept per unfinished trajectory, 0 = the default handles;
H handles / streams in rotation, default 2; prints JSON lines)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
if len(sys.argv) > 3:
    os.environ["DIRECT_DDP_YIELD"] = sys.argv[3]
H = int(sys.argv[4]) if len(sys.argv) > 4 else 2
from direct_amd import abi, devmem, problems, solver  # noqa: E402

dev = torch.device("cuda:0")
N = 100
batch = problems.make_batch("free", B, N, seed=1000, dtype=np.float32)
hs = [solver.DdpSolver(B, N, batch.p_max, np.float32) for _ in range(H)]
streams = [torch.cuda.Stream(device=dev) for _ in range(H)]
for s, st in zip(hs, streams):
    s.set_stream(st.cuda_stream)
g0 = hs[0].solve(abi.phase0_params(), batch)
b1 = batch.phase1_inputs(g0)
din = devmem.DeviceBatch(b1, dev)
outs = [devmem.DeviceResult(B, N, np.float32, dev) for _ in range(H)]
ref = devmem.DeviceResult(B, N, np.float32, dev)
torch.cuda.synchronize()
for name, p in (("fixed20", abi.phase1_params(iter_max=20, fixed_iters=1)), ("natural", abi.phase1_params())):
    hs[0].solve_device(p, din.cin, ref.cout)
    torch.cuda.synchronize()
    want = {k: v.clone() for k, v in ref.t.items()}
    its = int(want["fwd_passes"].sum().item())
    res = {}
    for mode in ("serial", "pipelined"):
        for _ in range(2):   # warm-up
            for i in range(H):
                hs[i if mode == "pipelined" else 0].solve_device(p, din.cin, outs[i].cout)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(R):
            hs[i % H if mode == "pipelined" else 0].solve_device(p, din.cin, outs[i % H].cout)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        same = all(torch.equal(outs[j].t[k], want[k]) for j in range(H) for k in want)
        res[mode] = dict(ms_per_batch=dt * 1e3 / R, M_iter_per_s=its * R / dt / 1e6, bit_identical=bool(same),
                         sched_error=[int(h.sched_error()) for h in hs])
    print(json.dumps(dict(workload=name, B=B, batches=R, handles=H, yield_k=os.environ.get("DIRECT_DDP_YIELD", "0"), iterations_per_batch=its, **res)))
