"""Ad-hoc GPU sanity run: per-pass + whole-solve parity against the oracle, then a timing probe."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from direct_amd import abi, problems, solver
from oracle import refapi

p0, p1 = abi.phase0_params(), abi.phase1_params()
def rel(a, b): return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))

def per_pass(kind, N, dtype):
    b = problems.make_batch(kind, 3, N, seed=7)
    s = solver.DdpSolver(3, N, b.p_max, dtype)
    s.begin(p0, b)
    r = [refapi.Stepper(p0, b, i) for i in range(3)]
    out = {}
    out["x0"] = max(rel(s.get(abi.FIELD_X)[i], r[i].get(abi.FIELD_X)) for i in range(3))
    out["c0"] = max(rel(s.get(abi.FIELD_C)[i][:, :r[i].ncmax], r[i].get(abi.FIELD_C)) for i in range(3))
    s.backward(); [q.backward() for q in r]
    for f, n in ((abi.FIELD_KU, "ku"), (abi.FIELD_KUU, "Ku"), (abi.FIELD_KS, "ks"), (abi.FIELD_KY, "ky")):
        out[n] = max(rel(s.get(f)[i], r[i].get(f)) for i in range(3))
    out["opterr"] = max(abs(s.scalars()["opterr"][i] / r[i].scalars()["opterr"] - 1) for i in range(3))
    s.forward(); [q.forward() for q in r]
    for f, n in ((abi.FIELD_X, "x"), (abi.FIELD_U, "u"), (abi.FIELD_S, "s"), (abi.FIELD_Y, "y")):
        out[n] = max(rel(s.get(f)[i], r[i].get(f)) for i in range(3))
    sc = s.scalars()
    out["step"] = [(int(sc["step"][i]), int(r[i].scalars()["step"])) for i in range(3)]
    out["cost"] = max(abs(sc["cost"][i] / r[i].scalars()["cost"] - 1) for i in range(3))
    print("per-pass", kind, N, np.dtype(dtype).name, {k: (v if isinstance(v, list) else "%.2e" % v) for k, v in out.items()}, flush=True)
    s.close()

def whole(kind, N, nb, dtype):
    b = problems.make_batch(kind, nb, N, seed=11)
    s = solver.DdpSolver(nb, N, b.p_max, dtype)
    r0, r1 = refapi.plan_batch(p0, p1, b)
    g0, g1 = s.plan(p0, p1, b)
    print("plan", kind, N, np.dtype(dtype).name, "rtn0 eq", int((g0.rtn == r0.rtn).sum()), "/", nb,
          "it0 eq", int((g0.iter_used == r0.iter_used).sum()), "rtn1 eq", int((g1.rtn == r1.rtn).sum()),
          "it1 eq", int((g1.iter_used == r1.iter_used).sum()),
          "cost1 rel max %.2e" % np.abs(g1.cost / r1.cost - 1).max(), "T rel %.2e" % rel(g1.T, r1.T),
          "bez rel %.2e" % rel(g1.bez, r1.bez), flush=True)
    s.close()

def timing(kind, B, N, dtype, iters=20):
    b = problems.make_batch(kind, B, N, seed=1000)
    s = solver.DdpSolver(B, N, b.p_max, dtype)
    t = time.time(); g0 = s.solve(p0, b); t0 = time.time() - t
    ms0, _ = s.last_kernel_ms()
    b1 = b.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, b.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
    pf = abi.phase1_params(iter_max=iters, fixed_iters=1)
    t = time.time(); g1 = s.solve(pf, b1); t1 = time.time() - t
    ms1, _ = s.last_kernel_ms()
    its = int(g1.fwd_passes.sum())
    print("timing", kind, B, N, np.dtype(dtype).name, "phase0 kernel %.1f ms (fwd %d) wall %.2fs; fixed-%d phase1 kernel %.1f ms -> %.3f M iter/s (wall %.2fs)"
          % (ms0, int(g0.fwd_passes.sum()), t0, iters, ms1, its / ms1 / 1e3, t1), "rtn hist", np.unique(g1.rtn, return_counts=True), flush=True)
    pn = abi.phase1_params()
    g2 = s.solve(pn, b1); ms2, _ = s.last_kernel_ms()
    print("   natural phase1 kernel %.1f ms iters %d -> %.3f M iter/s; rtn" % (ms2, int(g2.fwd_passes.sum()), g2.fwd_passes.sum() / ms2 / 1e3),
          np.unique(g2.rtn, return_counts=True), "iter mean %.1f max %d" % (g2.iter_used.mean(), g2.iter_used.max()), flush=True)
    s.close()

if __name__ == "__main__":
    print("abi", solver.lib().direct_ddp_abi_version(), solver.LIB_PATH, flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == "timing":
        timing("free", 4096, 100, np.float32)
        timing("corridor", 4096, 100, np.float32)
        timing("free", 4096, 100, np.float64)
        sys.exit(0)
    for dt in (np.float64, np.float32):
        per_pass("free", 5, dt)
        per_pass("corridor", 8, dt)
    whole("corridor", 12, 16, np.float64)
    whole("free", 20, 16, np.float64)
    whole("corridor", 12, 16, np.float32)
    timing("free", 4096, 100, np.float32)
    timing("corridor", 4096, 100, np.float32)
    timing("free", 4096, 100, np.float64)
