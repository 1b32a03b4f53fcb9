"""Wall-clock timeline of the fixed-20 launch of config 2, per trajectory and outer iteration (needs a -DDDP_TIMELINE build:
bash tools/fastbuild.sh timeline -DDDP_TIMELINE; DIRECT_DDP_LIB=direct_amd/lib/dev_timeline.so python tools/timeline.py [B] [out.json]).
Prints where the launch's time goes (work bound, slowest chains, idle gaps between a trajectory's iterations) and replays
the measured iteration durations through a discrete-event model of the ticket scheduler and of alternatives."""
import ctypes as C, heapq, json, sys
import numpy as np
sys.path.insert(0, ".")
from direct_amd import abi, problems, solver

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N, IT = 100, 20
b = problems.make_batch("free", B, N, seed=1000)
s = solver.DdpSolver(B, N, b.p_max, np.float32)
g0 = s.solve(abi.phase0_params(), b)
b1 = b.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, b.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
pf = abi.phase1_params(iter_max=IT, fixed_iters=1)
for _ in range(2):
    g1 = s.solve(pf, b1)
ms = s.last_kernel_ms()[0]
li = s.launch_info()
lib = solver.lib()
tl = np.zeros((B, 32, 4), np.uint64)
lib.direct_ddp_debug_timeline.argtypes = [C.c_void_p, C.c_void_p]
assert lib.direct_ddp_debug_timeline(s.h, tl.ctypes.data) == 0
tl = tl[:, :IT].astype(np.int64)
t0 = tl[:, :, 0].min()
st, mid, en, sp = (tl[:, :, 0] - t0) / 100.0, (tl[:, :, 1] - t0) / 100.0, (tl[:, :, 2] - t0) / 100.0, tl[:, :, 3]  # us
dur = en - st
busy = dur.sum(1)
span = en.max()
slots = li["resident_waves"]
gaps = st[:, 1:] - en[:, :-1]
last = np.argsort(en[:, -1])[-8:][::-1]
out = {
    "batch": B, "kernel_ms": ms, "span_us": float(span), "resident_waves": slots, "shared_sweep": li["shared_sweep"],
    "helper_front_knots": li.get("helper_front_knots", 0), "bwd_knot_visits": li["bwd_knot_visits"],
    "work_bound_us": float(busy.sum() / slots), "busy_us_q50_q90_max": [float(v) for v in np.quantile(busy, [0.5, 0.9, 1.0])],
    "iteration_us_mean_q50_q90_max": [float(dur.mean())] + [float(v) for v in np.quantile(dur, [0.5, 0.9, 1.0])],
    "bwd_us_mean": float((mid - st).mean()), "fwd_us_mean": float((en - mid).mean()),
    "first_start_us_q50_q90_max": [float(v) for v in np.quantile(st[:, 0], [0.5, 0.9, 1.0])],
    "idle_between_iterations_us_mean_per_trajectory": float(gaps.sum(1).mean()),
    "split_knot_share": float(sp.sum() / max(li["bwd_knot_visits"], 1)),
    "last_finishers": [dict(traj=int(i), finish_us=float(en[i, -1]), busy_us=float(busy[i]), first_start_us=float(st[i, 0]),
                            idle_us=float(gaps[i].sum()), bwd_us=float((mid[i] - st[i]).sum()), split_knots=int(sp[i].sum())) for i in last],
    "corr_consecutive_iteration_durations": float(np.corrcoef(dur[:, :-1].ravel(), dur[:, 1:].ravel())[0, 1]),
    "corr_first_half_second_half_busy": float(np.corrcoef(dur[:, :IT // 2].sum(1), dur[:, IT // 2:].sum(1))[0, 1]),
}


def simulate(policy, K=0, theta=0.0):
    """waves draw work; durations are the measured ones.  policy: 'tickets' (strict (epoch, trajectory) order, the wave that
    draws a ticket whose predecessor runs waits for it), 'ahead' (a wave that finishes (e, b) goes on with (e + 1, b) when that
    ticket is at most K rounds ahead of the draw pointer [and the iteration just finished took more than theta x the mean])"""
    done = np.zeros(B, int)          # iterations finished
    started = np.zeros(B, int)       # iterations started (claimed)
    tfin = np.zeros(B)               # finish time of the trajectory's last iteration
    mean = dur.mean()
    ptr = 0                          # next ticket
    ev = [(0.0, w, -1, -1) for w in range(slots)]  # (time free, wave, trajectory just finished, epoch)
    heapq.heapify(ev)
    waiting = {}                     # trajectory -> (wave, epoch) blocked on its predecessor
    T = 0.0
    total = B * IT
    while ev:
        t, w, fb, fe = heapq.heappop(ev)
        T = max(T, t)
        if fb >= 0:
            done[fb] = fe + 1
            tfin[fb] = t
            if fb in waiting:        # the wave that waited for this iteration starts now
                w2, e2 = waiting.pop(fb)
                heapq.heappush(ev, (t + dur[fb, e2], w2, fb, e2))
            elif policy == "ahead" and fe + 1 < IT and started[fb] == fe + 1:
                tk = (fe + 1) * B + fb
                if tk >= ptr and tk - ptr <= K * B and dur[fb, fe] > theta * mean:
                    started[fb] = fe + 2
                    heapq.heappush(ev, (t + dur[fb, fe + 1], w, fb, fe + 1))
                    continue
        while ptr < total:           # draw
            e, bb = divmod(ptr, B)
            ptr += 1
            if started[bb] > e:
                continue             # taken by a run-ahead
            started[bb] = e + 1
            if done[bb] >= e:
                heapq.heappush(ev, (t + dur[bb, e], w, bb, e))
            else:
                waiting[bb] = (w, e)
            break
    return T


out["model_us"] = {"tickets": simulate("tickets")}
for K in (1, 2, 4, 20):
    out["model_us"]["ahead_K%d" % K] = simulate("ahead", K)
    out["model_us"]["ahead_K%d_heavy1.2" % K] = simulate("ahead", K, 1.2)
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    np.savez_compressed(sys.argv[2].replace(".json", "_raw.npz"), st=st.astype(np.float32), mid=mid.astype(np.float32), en=en.astype(np.float32), sp=sp)
