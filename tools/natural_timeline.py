"""Wall-clock timeline of the NATURAL-EXIT launch of config 2 (needs a -DDDP_TIMELINE -DDDP_TIMELINE_DEPTH=64 build:
bash tools/fastbuild.sh timeline -DDDP_TIMELINE -DDDP_TIMELINE_DEPTH=64;
DIRECT_DDP_LIB=direct_amd/lib/dev_timeline.so python tools/natural_timeline.py [B] [out.json]).
Where the slowest chains spend their time (busy, waiting for their next ticket), how fast an iteration is against the number
of trajectories still running, and a replay of the measured durations through models of the scheduler: strict tickets
(what runs), and long chains KEPT by their wave (no wait for the ticket counter), chosen with hindsight (the true iteration counts) or by the phase-0 cost."""
import ctypes as C, heapq, json, sys
import numpy as np
sys.path.insert(0, ".")
from direct_amd import abi, problems, solver

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N, D = 100, 64
b = problems.make_batch("free", B, N, seed=1000)
s = solver.DdpSolver(B, N, b.p_max, np.float32)
g0 = s.solve(abi.phase0_params(), b)
b1 = b.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, b.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
for _ in range(2):
    g1 = s.solve(abi.phase1_params(), b1)
ms = s.last_kernel_ms()[0]
li = s.launch_info()
lib = solver.lib()
tl = np.zeros((B, D, 4), np.uint64)
lib.direct_ddp_debug_timeline.argtypes = [C.c_void_p, C.c_void_p]
assert lib.direct_ddp_debug_timeline(s.h, tl.ctypes.data) == 0
tl = tl.astype(np.int64)
n = g1.fwd_passes.astype(int)
assert n.max() <= D
valid = np.arange(D)[None, :] < n[:, None]
t0 = tl[:, :, 0][valid].min()
st, mid, en = [np.where(valid, (tl[:, :, i] - t0) / 100.0, np.nan) for i in range(3)]  # us
dur = en - st
busy = np.nansum(dur, 1)
fin = np.array([en[i, n[i] - 1] for i in range(B)])
span = fin.max()
slots = li["resident_waves"]
gaps = st[:, 1:] - en[:, :-1]
idle = np.nansum(gaps, 1)
order = np.argsort(fin)[::-1]
live_at = lambda t: int((fin > t).sum())
out = {"batch": B, "kernel_ms": ms, "span_us": float(span), "resident_waves": slots, "iterations_total": int(n.sum()),
       "iterations_q50_q90_max": [int(v) for v in np.quantile(n, [0.5, 0.9, 1.0])],
       "work_bound_us": float(busy.sum() / slots),
       "finish_us_q10_q50_q90_q99": [float(v) for v in np.quantile(fin, [0.1, 0.5, 0.9, 0.99])],
       "corr_phase0_cost_iterations": float(np.corrcoef(g0.cost, n)[0, 1]),
       "corr_iterations_finish": float(np.corrcoef(n, fin)[0, 1]),
       "slowest_chains": [dict(traj=int(i), iterations=int(n[i]), finish_us=float(fin[i]), busy_us=float(busy[i]), idle_us=float(idle[i]),
                               first_start_us=float(st[i, 0])) for i in order[:8]]}
# iteration duration against the number of trajectories still running when it started
lv = np.array([[live_at(st[i, e]) if e < n[i] else -1 for e in range(D)] for i in order[:64]])
dd = np.array([dur[i] for i in order[:64]])
bins = [(3500, 4097), (2500, 3500), (1500, 2500), (800, 1500), (400, 800), (100, 400), (0, 100)]
out["iteration_us_of_the_64_slowest_by_live_trajectories"] = {"%d-%d" % (lo, hi): [float(np.nanmean(dd[(lv >= lo) & (lv < hi)])) if ((lv >= lo) & (lv < hi)).any() else None,
                                                                                  float(np.nanmean(np.array([gaps[i] for i in order[:64]])[(lv[:, 1:] >= lo) & (lv[:, 1:] < hi)])) if ((lv[:, 1:] >= lo) & (lv[:, 1:] < hi)).any() else None]
                                                             for lo, hi in bins}


def simulate(keep):
    """waves draw tickets (epoch, trajectory) in order; durations are the measured ones.  keep[b]: the wave that finishes an
    iteration of b goes on with b's next one at once (the ticket it skips is void)."""
    done = np.zeros(B, int); started = np.zeros(B, int)
    ptr = 0
    total = B * int(n.max())
    ev = [(0.0, w, -1, -1) for w in range(slots)]
    heapq.heapify(ev)
    waiting = {}
    T = 0.0
    while ev:
        t, w, fb, fe = heapq.heappop(ev)
        T = max(T, t)
        if fb >= 0:
            done[fb] = fe + 1
            if fb in waiting:
                w2, e2 = waiting.pop(fb)
                heapq.heappush(ev, (t + dur[fb, e2], w2, fb, e2))
            elif keep[fb] and fe + 1 < n[fb] and started[fb] == fe + 1:
                started[fb] = fe + 2
                heapq.heappush(ev, (t + dur[fb, fe + 1], w, fb, fe + 1))
                continue
        while ptr < total:
            e, bb = divmod(ptr, B)
            ptr += 1
            if e >= n[bb] or started[bb] > e:
                continue
            started[bb] = e + 1
            if done[bb] >= e:
                heapq.heappush(ev, (t + dur[bb, e], w, bb, e))
            else:
                waiting[bb] = (w, e)
            break
    return T


none = np.zeros(B, bool)
out["model_us"] = {"tickets": simulate(none), "keep_all": simulate(~none)}
for frac in (0.5, 0.25, 0.125):
    k = int(B * frac)
    o = np.zeros(B, bool); o[np.argsort(n)[::-1][:k]] = True
    p = np.zeros(B, bool); p[np.argsort(g0.cost)[::-1][:k]] = True
    out["model_us"]["keep_hindsight_top_%g" % frac] = simulate(o)
    out["model_us"]["keep_phase0_cost_top_%g" % frac] = simulate(p)
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    np.savez_compressed(sys.argv[2].replace(".json", "_raw.npz"), st=st.astype(np.float32), mid=mid.astype(np.float32), en=en.astype(np.float32), n=n, cost0=g0.cost)
