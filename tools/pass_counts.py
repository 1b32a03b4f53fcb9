"""Exact dynamic instruction counts of the two sweeps: one stepwise backward pass and one forward pass (k_pass) of
bench.py's phase-1 workload under `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace`, divided by
the knots each sweep executed (direct_ddp_last_launch_info).  usage: run this file under rocprofv3, then
`python tools/pass_counts.py report <rocprof-dir>`; the visits are printed by the first step."""
import csv
import glob
import json
import sys

import numpy as np

sys.path.insert(0, ".")
if len(sys.argv) > 2 and sys.argv[1] == "report":
    vis = json.load(open("/tmp/pass_visits.json"))
    f = glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_pass" in r["Kernel_Name"]]
    disp = sorted({int(r["Dispatch_Id"]) for r in rows})
    for name, d, key in (("backward", disp[0], "bwd"), ("forward", disp[1], "fwd")):
        c = {}
        for r in rows:
            if int(r["Dispatch_Id"]) == d:
                c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        n = vis[key]
        print("%s sweep: %d knot visits;" % (name, n), " ".join("%s/knot %.0f" % (k.replace("SQ_INSTS_", ""), v / n) for k, v in sorted(c.items())))
    sys.exit(0)
from direct_amd import abi, problems, solver
B, N = 4096, 100
b = problems.make_batch("free", B, N, seed=1000)
s = solver.DdpSolver(B, N, b.p_max, np.float32)
g0 = s.solve(abi.phase0_params(), b)
b1 = b.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, b.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
pf = abi.phase1_params(iter_max=20, fixed_iters=1)
s.begin(pf, b1)
s.iterate(6)       # a few iterations in: typical line searches
s.backward()
vb = s.launch_info()
s.forward()
vf = s.launch_info()
json.dump({"bwd": vb["bwd_knot_visits"], "fwd": vf["fwd_knot_visits"]}, open("/tmp/pass_visits.json", "w"))
print("visits", vb["bwd_knot_visits"], vf["fwd_knot_visits"])
