"""Throughput of the output-sampling kernel k_sample on device-resident arrays.
usage: sample_bench.py [B] [N] [f32|f64]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from direct_amd import abi, problems, solver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dt_np = np.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else np.float32
t_dt = torch.float64 if dt_np == np.float64 else torch.float32
nb = min(B, 4096)
b = problems.make_batch("free", nb, N, seed=1000, dtype=dt_np)
s = solver.DdpSolver(nb, N, b.p_max, dt_np)
g0, g1 = s.plan(abi.phase0_params(), abi.phase1_params(iter_max=20, fixed_iters=1), b)
s.close()
rep = B // nb
dev = torch.device("cuda", 0)
bez = torch.from_numpy(np.tile(g1.bez, (rep, 1, 1))).to(dev)
T = torch.from_numpy(np.tile(g1.T, (rep, 1))).to(dev)
nseg = torch.from_numpy(np.tile(b.n_seg, rep)).to(dev)
Bt = nb * rep
s = solver.DdpSolver(Bt, N, b.p_max, dt_np)
s.set_stream(torch.cuda.current_stream().cuda_stream)
for derivs, dt in ((0, 0.1), (2, 0.1), (2, 0.02)):
    cap = int(np.ceil(g1.T.sum(1).max() / dt)) + 2 * N
    o = {k: torch.zeros(Bt, cap, 3, dtype=t_dt, device=dev) for k in ("pos", "vel", "acc")}
    cnt = torch.zeros(Bt, dtype=torch.int32, device=dev)
    length = torch.zeros(Bt, dtype=t_dt, device=dev)
    vmax = torch.zeros(Bt, dtype=t_dt, device=dev)
    amax = torch.zeros(Bt, dtype=t_dt, device=dev)
    cin, cout = abi.SampleIn(), abi.SampleOut()
    cin.batch, cin.n_seg_max, cin.capacity, cin.derivs, cin.mem = Bt, N, cap, derivs, abi.MEM_DEVICE
    cin.n_seg, cin.bez, cin.T, cin.dt = nseg.data_ptr(), bez.data_ptr(), T.data_ptr(), dt
    cout.count, cout.pos, cout.length = cnt.data_ptr(), o["pos"].data_ptr(), length.data_ptr()
    if derivs:
        cout.vel, cout.acc, cout.vmax, cout.amax = o["vel"].data_ptr(), o["acc"].data_ptr(), vmax.data_ptr(), amax.data_ptr()
    ms = []
    for _ in range(6):
        s.sample_device(cin, cout)
        ms.append(s.sample_last_ms())
    pts = int(cnt.sum().item())
    isz = np.dtype(dt_np).itemsize
    byts = pts * 3 * isz * (1 + derivs) + Bt * N * 19 * isz
    m = float(np.median(ms[1:]))
    print("B=%d N=%d %s derivs=%d dt=%.2f: %d points, %.3f ms -> %.1f Mpoints/s, %.1f GB/s algorithmic (%.1f%% of 8 TB/s)"
          % (Bt, N, np.dtype(dt_np).name, derivs, dt, pts, m, pts / m / 1e3, byts / m / 1e6, byts / m / 1e6 / 80))
    del o
s.close()
