"""Serial control of yield_stress.py (not a pytest test): R launches of ONE ordinary handle on one stream, natural exits and
fixed-20, each compared with the first bit for bit; reports scheduler time-outs.  usage: serial_stress.py [R] [B] [flags]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from direct_amd import abi, devmem, problems, solver  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda:0")
b = problems.make_batch("corridor", B, 100, seed=6000 + B).astype(np.float32)
one = solver.DdpSolver(B, 100, b.p_max, np.float32, flags=flags)
one.set_stream(torch.cuda.Stream(device=dev).cuda_stream)
g0 = one.solve(abi.phase0_params(), b)
din = devmem.DeviceBatch(b.phase1_inputs(g0), dev)
ref = devmem.DeviceResult(B, 100, np.float32, dev)
outs = [devmem.DeviceResult(B, 100, np.float32, dev) for _ in range(4)]
bad = 0
for p in (abi.phase1_params(iter_max=20, fixed_iters=1), abi.phase1_params(iter_max=40)):
    one.solve_device(p, din.cin, ref.cout)
    torch.cuda.synchronize()
    for base in range(0, R, 4):
        for i in range(4):
            one.solve_device(p, din.cin, outs[i].cout)
        torch.cuda.synchronize()
        for i in range(4):
            if not all(torch.equal(outs[i].t[k], ref.t[k]) for k in ref.t):
                bad += 1
                print("MISMATCH fixed=%d launch %d sched_error %s %s" % (p.fixed_iters, base + i, one.sched_error(), one.sched_debug()), flush=True)
print("serial B = %d flags = %d: %d launches per workload, %d mismatches, sched_error %d" % (B, flags, R, bad, one.sched_error()))
