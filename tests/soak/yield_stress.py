"""Stress of DIRECT_FLAG_YIELD (not a pytest test): for several batch sizes R pipelined launches on two yielding handles must
reproduce one serial launch of an ordinary handle bit for bit, natural exits and fixed-20, scheduler error flags clear.
usage: python tests/soak/yield_stress.py [launches per size] [sizes, comma separated] [launches of every size but the last]
                                         [keep] [churn=N]
(the third argument separates "the earlier stages existed" - handles and streams created and closed - from "the earlier stages
ran for minutes" when looking for what makes the last stage stall: DESIGN.md 7.6.  `keep`: the handles of the earlier stages stay
open until the process ends - nothing is freed, no address is re-used; `churn=N`: N one-trajectory handles are created and closed
in front of the last stage and nothing else happens before it - the smallest "a handle was closed before" there is)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from direct_amd import abi, devmem, problems, solver  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 40
os.makedirs("gpurun_out", exist_ok=True)
dev = torch.device("cuda:0")
total = 0
SIZES = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else None
R_ALL, R_PRE = R, (int(sys.argv[3]) if len(sys.argv) > 3 else R)
KEEP = "keep" in sys.argv[4:]
CHURN = max([int(a.split("=")[1]) for a in sys.argv[4:] if a.startswith("churn=")] + [0])
kept = []
for B, kind, dt in ((1, "corridor", np.float32), (33, "free", np.float32), (700, "corridor", np.float64), (3100, "free", np.float32),
                    (4096, "corridor", np.float32), (4600, "free", np.float32), (9000, "corridor", np.float32)):
    if SIZES is not None and B not in SIZES:
        continue
    last = B == (max(SIZES) if SIZES is not None else 9000)
    R = R_ALL if last else R_PRE
    if last and CHURN:
        tiny = problems.make_batch("free", 1, 100, seed=1).astype(np.float32)
        for _ in range(CHURN):
            solver.DdpSolver(1, 100, tiny.p_max, np.float32, flags=abi.FLAG_YIELD).close()
    b = problems.make_batch(kind, B, 100, seed=6000 + B).astype(dt)
    one = solver.DdpSolver(B, 100, b.p_max, dt)
    g0 = one.solve(abi.phase0_params(), b)
    din = devmem.DeviceBatch(b.phase1_inputs(g0), dev)
    ref = devmem.DeviceResult(B, 100, dt, dev)
    hs = [solver.DdpSolver(B, 100, b.p_max, dt, flags=abi.FLAG_YIELD) for _ in range(2)]
    st = [torch.cuda.Stream(device=dev) for _ in range(2)]
    for h, s in zip(hs, st):
        h.set_stream(s.cuda_stream)
    outs = [devmem.DeviceResult(B, 100, dt, dev) for _ in range(4)]
    slow = []
    for p in (abi.phase1_params(iter_max=20, fixed_iters=1), abi.phase1_params(iter_max=40)):
        one.solve_device(p, din.cin, ref.cout)
        torch.cuda.synchronize()
        for base in range(0, R, 4):
            tg = time.perf_counter()
            for i in range(4):
                hs[i % 2].solve_device(p, din.cin, outs[i].cout)
            torch.cuda.synchronize()
            tg = time.perf_counter() - tg
            slow.append(tg)
            for i in range(4):
                for k in ref.t:
                    if not torch.equal(outs[i].t[k], ref.t[k]):
                        d = torch.nonzero((outs[i].t[k] != ref.t[k]).reshape(B, -1).any(dim=1)).flatten()
                        idx = d[:5].cpu().numpy()
                        np.savez_compressed("gpurun_out/yield_mismatch.npz", fwd=outs[i].t["fwd_passes"].cpu().numpy(), fwd_ref=ref.t["fwd_passes"].cpu().numpy(),
                                            rtn=outs[i].t["rtn"].cpu().numpy(), rtn_ref=ref.t["rtn"].cpu().numpy(), iters=outs[i].t["iter_used"].cpu().numpy(),
                                            iters_ref=ref.t["iter_used"].cpu().numpy(), cost=outs[i].t["cost"].cpu().numpy(), cost_ref=ref.t["cost"].cpu().numpy(),
                                            others=np.stack([outs[j].t["fwd_passes"].cpu().numpy() for j in range(4)]), launch=base + i, fixed=p.fixed_iters)
                        raise SystemExit("MISMATCH B=%d %s fixed=%d launch %d field %s: %d trajectories, first %s; rtn %s vs %s, fwd_passes %s vs %s, sched_error %s"
                                         % (B, kind, p.fixed_iters, base + i, k, len(d), idx, outs[i].t["rtn"][idx].cpu().numpy(), ref.t["rtn"][idx].cpu().numpy(),
                                            outs[i].t["fwd_passes"][idx].cpu().numpy(), ref.t["fwd_passes"][idx].cpu().numpy(), [(h.sched_error(), h.sched_debug()) for h in hs])
                                         + " fwd_passes min %d" % int(outs[i].t["fwd_passes"].min().item()))
            total += 4
        assert one.sched_error() == 0 and all(h.sched_error() == 0 for h in hs)
    for h in hs + [one]:
        if KEEP and not last:
            kept.append(h)
        else:
            h.close()
    print("B = %d (%s, %s): %d pipelined launches per workload reproduce the serial launch; groups of 4 launches: median %.0f ms, max %.0f ms, over 1 s: %d"
          % (B, kind, np.dtype(dt).name, R, 1e3 * float(np.median(slow)), 1e3 * max(slow), sum(1 for v in slow if v > 1.0)), flush=True)
print("ok: %d launches" % total)
