#!/usr/bin/env python
"""Headline benchmark: DDP iterations per second over a batch (BASELINE.json metric).

Default workload = BASELINE config 2: 4096 free-space corridors PER GPU, N = 100 segments, float storage,
the reference's polynomial-segment model (9 states / 10 controls).  `--config 3|4|5`
selects the other single-GPU-shaped configurations (5 = the per-GPU shard of config 5: 16384 polyhedron
corridors per GPU); --batch/--nseg/--kind/--dtype override individual fields and config.workload always
names what was actually run.

One "step" = one pass of the hot path over the batch: polyCurveGeneration for every corridor
(ddp_optimizer.cpp:5-438) in its phase-1 configuration (feasible IPDDP, launch-file weights), warm started
from the phase-0 result and run for a FIXED 20 iterations with the early exits disabled so that every
implementation does identical work.  Inputs are resident in HBM before the timed region,
outputs stay in HBM.  After the timed region the same batch is solved once more with the reference's natural
exits (secondary figure: rtn histogram, iterations to exit).

`python bench.py --gpus N` with N > 1 and no torchrun environment re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, one rank per GPU over
RCCL; launched by torchrun directly it reads RANK / LOCAL_RANK / WORLD_SIZE.  Ranks shard the problem stream
(weak scaling, no data-path collective); the config-5 gather of the best trajectory (direct_ddp_gather_best:
two ncclAllGather through the library's C entry point) runs once after the timed region.
"""
import argparse
import glob
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
F64_VECTOR_PEAK_TF = 78.6  # MI355X fp64 vector peak (same guide)
FIXED_ITERS = 20
CONFIGS = {  # BASELINE.json configs[1..4]
    2: dict(kind="free", batch=4096, nseg=100, dtype="f32", name="config 2"),
    3: dict(kind="corridor", batch=4096, nseg=100, dtype="f32", name="config 3"),
    4: dict(kind="corridor", batch=16384, nseg=300, dtype="f64", name="config 4"),
    5: dict(kind="corridor", batch=16384, nseg=100, dtype="f32", name="config 5 (per-GPU shard of 131072 / 8)"),
    # config 4's shape (N = 300, double storage, 16384 per GPU) on the free-space generator: every trajectory leaves phase 0
    # feasible, so the timed iterations run FULL-LENGTH forward rollouts (config 4's own corridors stay in infeasible mode,
    # where every line-search trial dies at the fraction-to-boundary rule after a few knots: VERDICT r04 missing #1)
    6: dict(kind="free", batch=16384, nseg=300, dtype="f64", name="config 4, feasible-mode variant (free-space generator)"),
}


def usable_cpus():
    """Host threads this process may actually run on: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(batch1, params, sample):
    """The oracle ("port": Eigen-free restatement of the reference, the original cannot be built
    here) on a bounded sample of the same workload, OpenMP over the usable host cores.  Two thread
    counts are tried (all usable threads, and half of them in case they are SMT siblings); the better
    one is reported with the thread count it used."""
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "threads")
    from oracle import refapi
    refapi.build()
    sub = batch1.select(np.arange(sample))
    usable = usable_cpus()
    refapi.solve_batch(params, sub.select(np.arange(min(sample, 2 * usable))), n_threads=usable)  # warm up threads + arenas
    best = None
    for nt in sorted({usable, max(1, usable // 2)}, reverse=True):
        t = time.perf_counter()
        res, _ = refapi.solve_batch(params, sub, n_threads=nt)
        dt = time.perf_counter() - t
        v = float(res.fwd_passes.sum() / dt)
        if best is None or v > best[0]:
            best = (v, nt, dt)
    t1 = time.perf_counter()
    res1, _ = refapi.solve_batch(params, sub.select(np.arange(min(8, sample))), n_threads=1)
    dt1 = time.perf_counter() - t1
    return {"value": best[0], "unit": "iter/s", "cores": int(best[1]), "kind": "port",
            "sample": "%d of the %d corridors of rank 0's batch, same fixed-%d-iteration phase-1 solve, fp64, "
                      "OpenMP schedule(dynamic,1), %d threads (host reports %d CPUs, %d usable); %.1f s wall"
                      % (sample, batch1.batch, FIXED_ITERS, best[1], os.cpu_count() or 1, usable, best[2]),
            "single_thread_value": float(res1.fwd_passes.sum() / dt1)}


def matching_profile(pattern, workload):
    """Newest committed profiles/<pattern> whose recorded "workload" equals this run's; None otherwise.
    A profile of a different batch / N / dtype / kind says nothing about this launch."""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("workload") == workload:
            d["_file"] = os.path.relpath(f, ROOT)
            return d
    return None


def live_hbm_traffic(workload, timeout_s=150):
    """HBM bytes of ONE timed launch from the PMC counters, measured NOW on this box: two separate rocprofv3 passes
    (FETCH_SIZE and WRITE_SIZE cannot share one) over tools/prof_one.py, which runs this very workload (phase 0, then the
    fixed-iteration phase-1 launch: the LAST k_iterate* dispatch of the process, checked against the HIP-event time the
    script prints).  Units and the gfx950 correction as /opt/skills/guides/MI355X_MICROARCH.md prescribes (counter x 1 KiB;
    FETCH_SIZE reports half of the bytes read - the factors tools/hbm_calib measured for this access width are taken from
    the newest profiles/r*_hbm_calib.json, 0.5 / 1.0 otherwise).  None when rocprofv3 is missing or a pass fails."""
    import csv
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None
    cal = {"fetch_factor_dword": 0.5, "write_factor_dword": 1.0}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_calib.json")), reverse=True):
        try:
            cal.update({k: v for k, v in json.load(open(f)).items() if k in cal})
            break
        except (OSError, ValueError):
            continue
    out, ms = {}, {}
    tmp = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        for name in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, name)
            cmd = [rp, "--pmc", name, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "q", "--", sys.executable,
                   os.path.join(ROOT, "tools", "prof_one.py"), workload["kind"], workload["dtype"], str(workload["batch"]),
                   str(workload["nseg"]), str(workload["fixed_iters"])]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=timeout_s)
            ev = [float(l.split()[2]) for l in r.stdout.splitlines() if l.startswith("kernel ms")]
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not ev or not files:
                return None
            by = {}
            for row in csv.DictReader(open(files[0])):
                if "k_iterate" in row["Kernel_Name"] and row["Counter_Name"] == name:
                    by[int(row["Dispatch_Id"])] = by.get(int(row["Dispatch_Id"]), 0.0) + float(row["Counter_Value"])
            if not by:
                return None
            out[name] = by[max(by)]  # the last k_iterate* dispatch = the timed launch
            ms[name] = ev[-1]
    except (subprocess.TimeoutExpired, OSError, ValueError, KeyError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch_b = out["FETCH_SIZE"] * 1024.0 / cal["fetch_factor_dword"]
    write_b = out["WRITE_SIZE"] * 1024.0 / cal["write_factor_dword"]
    return {"traffic_bytes_per_launch": fetch_b + write_b, "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
            "kernel_ms_under_pmc": [ms["FETCH_SIZE"], ms["WRITE_SIZE"]], "calibration": cal}


def live_sq_counters(workload, timeout_s=150):
    """The SQ counters behind roofline_compute, measured NOW on this box: two more rocprofv3 --pmc passes over
    tools/prof_one.py (instruction mix; pipe activity and LDS bank conflicts), last k_iterate* dispatch, checked against
    the HIP-event time like the traffic passes.  Returns the derived figures of tools/pmc_collect.py, or None."""
    import csv
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None
    sets = (["SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_LDS"],
            ["SQ_ACTIVE_INST_VALU", "SQ_BUSY_CU_CYCLES", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES"])
    c, ms = {}, []
    tmp = tempfile.mkdtemp(prefix="bench_sq_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        for i, names in enumerate(sets):
            d = os.path.join(tmp, "p%d" % i)
            cmd = [rp, "--pmc"] + names + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "q", "--", sys.executable,
                   os.path.join(ROOT, "tools", "prof_one.py"), workload["kind"], workload["dtype"], str(workload["batch"]),
                   str(workload["nseg"]), str(workload["fixed_iters"])]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=timeout_s)
            ev = [float(l.split()[2]) for l in r.stdout.splitlines() if l.startswith("kernel ms")]
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not ev or not files:
                return None
            by = {}
            for row in csv.DictReader(open(files[0])):
                if "k_iterate" in row["Kernel_Name"]:
                    q = by.setdefault(row["Counter_Name"], {})
                    q[int(row["Dispatch_Id"])] = q.get(int(row["Dispatch_Id"]), 0.0) + float(row["Counter_Value"])
            for n, q in by.items():
                c[n] = q[max(q)]
            ms.append(ev[-1])
    except (subprocess.TimeoutExpired, OSError, ValueError, KeyError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    need = ("SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_ACTIVE_INST_VALU",
            "SQ_BUSY_CU_CYCLES", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT")
    if any(k not in c for k in need):
        return None
    it = float(workload["batch"] * workload["fixed_iters"])
    arith = c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_ADD_F64"]
    return {"kernel_ms_under_pmc": ms, "counters": c,
            "derived": {"valu_insts_per_ddp_iteration": c["SQ_INSTS_VALU"] / it,
                        "lds_insts_per_ddp_iteration": c.get("SQ_INSTS_LDS", 0.0) / it,
                        "f64_arith_frac_of_valu": arith / c["SQ_INSTS_VALU"],
                        "f64_flops_per_ddp_iteration": (2 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_ADD_F64"]) * 64 / it,
                        "valu_busy_frac_per_simd": c["SQ_ACTIVE_INST_VALU"] / c["SQ_BUSY_CU_CYCLES"],
                        "lds_busy_frac_per_cu": c["SQ_LDS_IDX_ACTIVE"] / c["SQ_BUSY_CU_CYCLES"],
                        "lds_bank_conflict_frac_of_lds_cycles": c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0)}}


def live_cluster_pmc(timeout_s=200):
    """L1 line accesses and HBM fetch of k_convex, measured NOW: two rocprofv3 --pmc passes over the cluster generator's
    own 64-seed run (tests/soak/cluster_bench.py 64: a 2-seed warm-up and six 64-seed calls), summed over k_convex's
    dispatches.  Returns the dict corridor_clusters_line reads from a committed profile, or None."""
    import csv
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    script = os.path.join(ROOT, "tests", "soak", "cluster_bench.py")
    if not os.path.exists(rp) or not os.path.exists(script):
        return None
    tmp = tempfile.mkdtemp(prefix="bench_cl_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = {}
    try:
        for name in ("TCP_TOTAL_CACHE_ACCESSES_sum", "FETCH_SIZE"):
            d = os.path.join(tmp, name)
            r = subprocess.run([rp, "--pmc", name, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "q", "--", sys.executable, script, "64"],
                               cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            tot = 0.0
            for row in csv.DictReader(open(files[0])):
                if "k_convex" in row["Kernel_Name"] and row["Counter_Name"] == name:
                    tot += float(row["Counter_Value"])
            if tot <= 0:
                return None
            out[name] = tot
    except (subprocess.TimeoutExpired, OSError, ValueError, KeyError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {"k_convex": out, "generation_calls_of_64_seeds": 6, "_file": None}


def hbm_copy_gbs(torch, dev):
    """Measured HBM bandwidth of a device-to-device copy (read + write bytes / time): the second, measured
    denominator SURVEY.md 8(d) asks for next to the 8 TB/s datasheet figure."""
    n = 1 << 28  # 2 x 1 GiB of float32
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del a, b
    torch.cuda.empty_cache()
    return 2.0 * n * 4 / (ms * 1e-3) / 1e9


def label_model_line(torch, dev, B, with_cpu):
    """The model BASELINE.json's metric string literally names (12-state / 4-control quadrotor, N = 100), reported
    SEPARATELY: it has no reference counterpart (include/direct_quad.h).  Same batch size as the main workload, float
    storage, a fixed 10 iLQR iterations per step, inputs resident in HBM."""
    from direct_amd import quad
    N, iters = 100, 10
    x0, xg = quad.label_problems(B, seed=1000)
    p = quad.default_params(iter_max=iters, fixed_iters=1)
    q = quad.QuadSolver(B, N, np.float32, device=dev.index or 0)
    q.set_stream(torch.cuda.current_stream().cuda_stream)
    tx0, txg = torch.from_numpy(x0.astype(np.float32)).to(dev), torch.from_numpy(xg.astype(np.float32)).to(dev)
    cost, it = torch.zeros(B, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
    ms = []
    for rep in range(7):
        q.solve_device(p, B, tx0.data_ptr(), txg.data_ptr(), cost.data_ptr(), it.data_ptr())
        torch.cuda.synchronize()
        if rep >= 2:
            ms.append(q.last_kernel_ms())
    n_it = int(it.sum().item())
    k_ms = float(np.mean(ms))
    byts = quad.ALGORITHMIC_WORDS_PER_KNOT_ITER * 4 * N * n_it
    out = {"note": "NO REFERENCE COUNTERPART: the 12-state / 4-control quadrotor of BASELINE.json's metric string does not exist "
                   "in ntu-caokun/DIRECT (SURVEY.md section 0); Gauss-Newton iLQR, explicit Euler dt 0.05, parity only against "
                   "this repo's own CPU checker",
           "value": n_it / (k_ms * 1e-3), "unit": "iter/s", "workload": "%d trajectories, N=%d knots, f32 storage, fixed %d iterations" % (B, N, iters),
           "kernel_ms": k_ms, "roofline": {"bound": "hbm", "achieved": byts / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": byts / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": byts}}
    q.close()
    if with_cpu:
        from oracle import quadapi
        t = time.perf_counter()
        r = quadapi.solve_batch(p, N, x0[:64], xg[:64], n_threads=1)
        out["cpu_baseline"] = {"value": float(r["iters"].sum() / (time.perf_counter() - t)), "unit": "iter/s", "cores": 1,
                               "kind": "port", "sample": "64 of the trajectories, this repo's own CPU checker (no reference exists)"}
    return out


L1_PEAK_GBS = 39300.0  # vector L1: 64 B per clock and CU x 256 CUs x 2.4 GHz (MI355X_MICROARCH.md: 32 KiB L1 per CU, 2.4 GHz)


def corridor_clusters_line(dev, with_cpu, live=True):
    """SURVEY.md 8(f-4), reported SEPARATELY: seed voxels -> voxel clusters -> polytope planes on the device
    (include/direct_cluster.h; polyhedron_generator + poly_utils.cpp:127-206, 282-389), 64 seeds on a synthetic
    200 x 200 x 40 map, clusters left on the device, planes back to the host.  Not part of `value`.
    roofline: the dominant kernel (k_convex, ray-cast convexity tests: byte gathers from the map and the summed-area
    table) is bound by the vector L1, not by HBM - its cache-line accesses (TCP_TOTAL_CACHE_ACCESSES x 64 B, newest
    committed profiles/r*_cluster_pmc.json: static) over its share of this run's kernel time.
    cpu_baseline: the REFERENCE's own serialConvexTest (oracle/_ref, compiled from cluster_engine_cpu.cpp) inside the
    restated loops of cluster_server_cpu.cpp, one thread, on a bounded sample of the same seeds."""
    from direct_amd import cluster, problems
    dims = (200, 200, 40)
    grid, seeds = problems.make_voxel_map(dims, seed=7, n_pillars=170, n_boxes=70, n_rings=12)
    seeds = seeds[:64]
    gen = cluster.ClusterGenerator(dims, max_batch=64, cluster_capacity=50000, candidate_capacity=10000, device=dev.index or 0)
    gen.set_map(grid)
    gen.polygon_generation(seeds[:2], fetch_clusters=False)
    ts, hp, r, kms = [], None, None, None
    for _ in range(3):
        t = time.perf_counter()
        r = gen.polygon_generation(seeds, 1000, 50, fetch_clusters=False)
        kms = gen.last_ms()
        hp = gen.hull_planes(0.2, np.array([-20.0, -20.0, 0.0]), batch=len(seeds), plane_capacity=128, vertex_capacity=512)
        ts.append(time.perf_counter() - t)
    gen.close()
    out = {"workload": "64 seeds, 200x200x40 voxel map (1.6 % obstacles)", "ms_per_seed": min(ts) * 1e3 / len(seeds),
           "wall_ms": min(ts) * 1e3, "generation_kernels_ms": kms, "cluster_voxels_mean": float(r["cluster_num"].mean()),
           "planes_mean": float(hp["n_planes"].mean()),
           "planes_max": int(hp["n_planes"].max()), "seeds_ok": int(((r["rtn"] == 0) & (hp["rtn"] == 0)).sum()),
           "note": "bit-identical to the reference's CPU path (clusters) / exact facet planes pinned against its quickhull: tests/test_gpu_cluster.py, tests/test_gpu_hull.py"}
    pmc = live_cluster_pmc() if live else None
    pmc_kind = "live: TCP_TOTAL_CACHE_ACCESSES / FETCH_SIZE passes of tests/soak/cluster_bench.py 64 on this box, in this run"
    for f in ([] if pmc is not None else sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cluster_pmc.json")), reverse=True)):
        pmc_kind = ("static: committed PMC profile of tests/soak/cluster_bench.py 64 (L1 accesses and HBM fetch per call), "
                    "this run's kernel time")
        try:
            pmc = json.load(open(f))
            pmc["_file"] = os.path.relpath(f, ROOT)
            break
        except (OSError, ValueError):
            continue
    if pmc is not None and "k_convex" in pmc and "TCP_TOTAL_CACHE_ACCESSES_sum" in pmc["k_convex"] and kms:
        calls = pmc.get("generation_calls_of_64_seeds", 6)   # tests/soak/cluster_bench.py 64: 6 full calls + a 2-seed warm-up
        share = 0.72   # k_convex's share of the generation's kernel time: from the newest committed rocprofv3 --stats summary
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cluster_kernel_stats.csv")), reverse=True):
            try:
                import csv
                rows = list(csv.DictReader(open(f)))
                gen_k = ("k_convex", "k_resolve", "k_compact", "k_inflate", "k_mark", "k_apply", "k_chunk_box", "k_flags_init", "k_sat_scan", "k_emit")
                tot = sum(float(q["TotalDurationNs"]) for q in rows if any(n in q["Name"] for n in gen_k))
                cv = sum(float(q["TotalDurationNs"]) for q in rows if "k_convex" in q["Name"])
                if tot > 0:
                    share = cv / tot
                break
            except (OSError, ValueError, KeyError):
                continue
        byts = pmc["k_convex"]["TCP_TOTAL_CACHE_ACCESSES_sum"] * 64.0 / calls
        ach = byts / (kms * share * 1e-3) / 1e9
        out["roofline"] = {"bound": "l1", "kernel": "k_convex", "achieved": ach, "peak": L1_PEAK_GBS, "unit": "GB/s", "frac": ach / L1_PEAK_GBS,
                           "traffic": pmc["k_convex"].get("FETCH_SIZE", 0) * 1024.0 * 2 / calls if "FETCH_SIZE" in pmc["k_convex"] else None,
                           "traffic_kind": pmc_kind, "source": pmc["_file"]}
    if with_cpu:
        from oracle import clusterapi as ca
        ca.use_reference_convex_test(ca.ref_lib() is not None)
        m = 4
        t = time.perf_counter()
        for sd in seeds[:m]:
            ca.polygon_generation(grid, sd, 1000, 50)
        cpu_ms = (time.perf_counter() - t) * 1e3 / m
        kind = "reference" if ca.ref_lib() is not None else "port"
        ca.use_reference_convex_test(False)
        out["cpu_baseline"] = {"value": cpu_ms, "unit": "ms/seed", "cores": 1, "kind": kind,
                               "sample": "%d of the 64 seeds: serialConvexTest of polyhedron_generator/src/cluster_engine_cpu.cpp (oracle/_ref) "
                                         "inside the restated loops of cluster_server_cpu.cpp:257-528, one thread" % m}
    return out


def single_plan_line(dev, with_cpu):
    """The reference's actual call pattern: ONE corridor, planned when the user clicks (teach_repeat_planner.cpp:895-921:
    phase 0, UpdateTime, phase 1), host buffers in and out.  The corridor is a real one: a 12-segment replay plan of the
    voxel-map -> corridorGeneration chain (direct_amd/data/real_corridor_n12.npz, tools/make_real_corridor_fixture.py),
    double storage like the reference.  Wall time of direct_ddp_plan_batch with batch = 1 (H2D of the corridor, both
    phases' kernels, D2H of the coefficients), median of 20 calls; the oracle on one host thread beside it.  Never `value`."""
    from direct_amd import abi, solver
    f = os.path.join(ROOT, "direct_amd", "data", "real_corridor_n12.npz")
    if not os.path.exists(f):
        return None
    d = np.load(f)
    batch = abi.HostBatch(d["n_seg"], d["x0"], d["xd"], d["T0"], d["n_planes"], d["planes"], seeds=d["seeds"])
    p0, p1 = abi.phase0_params(), abi.phase1_params()
    s = solver.DdpSolver(1, batch.n_seg_max, batch.p_max, np.float64, device=dev.index or 0)
    s.plan(p0, p1, batch)
    ts, ks = [], []
    for _ in range(20):
        t = time.perf_counter()
        g0, g1 = s.plan(p0, p1, batch)
        ts.append(time.perf_counter() - t)
        ks.append(s.last_kernel_ms()[0])
    li = s.launch_info()
    s.close()
    its = int(g0.fwd_passes.sum() + g1.fwd_passes.sum())
    ms = float(np.median(ts)) * 1e3
    out = {"workload": "1 real corridor, N=%d segments, widest polytope %d planes, f64 storage, two-phase plan (phase 0 + UpdateTime + phase 1), host buffers"
                       % (int(batch.n_seg[0]), int(batch.p_max)),
           "ms_per_plan": ms, "ms_per_plan_min": float(np.min(ts)) * 1e3, "phase1_kernel_ms": float(np.median(ks)),
           "iterations": its, "iter_per_s": its / (ms * 1e-3), "rtn": [int(g0.rtn[0]), int(g1.rtn[0])],
           "schedule": {k: li.get(k) for k in ("dynamic", "shared_search", "shared_sweep", "single_steps", "resident_waves")},
           "source": str(d["source"])}
    if with_cpu:
        from oracle import refapi
        refapi.build()
        refapi.plan_batch(p0, p1, batch)
        t = time.perf_counter()
        reps = 5
        for _ in range(reps):
            r0, r1 = refapi.plan_batch(p0, p1, batch)
        cms = (time.perf_counter() - t) * 1e3 / reps
        out["cpu_baseline"] = {"value": cms, "unit": "ms/plan", "cores": 1, "kind": "port",
                               "sample": "the same plan, oracle/direct_ref.c, one host thread, mean of %d" % reps,
                               "iterations": int(r0.fwd_passes.sum() + r1.fwd_passes.sum()),
                               "same_rtn_and_iterations": bool(r1.rtn[0] == g1.rtn[0] and r1.iter_used[0] == g1.iter_used[0]
                                                               and r0.rtn[0] == g0.rtn[0] and r0.iter_used[0] == g0.iter_used[0])}
    return out


def pipelined_line(torch, dev, batch1, np_dt, params_fixed, batches=8):
    """Secondary: back-to-back independent batches on TWO handles with DIRECT_FLAG_YIELD, a stream each, against the same
    batches on one handle (the reference's contract: an independent optimiser object per call, TRP:853-854).  The second
    handle's persistent waves become resident as the first's leave: launch i + 1's first epochs fill the CUs that launch
    i's slowest chains leave idle (natural exits: the chip empties while 48-iteration chains finish alone).  Results are
    checked bit for bit against a serial launch.  Never `value`."""
    from direct_amd import abi, devmem, solver
    B, N = batch1.batch, batch1.n_seg_max
    hs = [solver.DdpSolver(B, N, batch1.p_max, np_dt, device=dev.index or 0, flags=abi.FLAG_YIELD) for _ in range(2)]
    one = solver.DdpSolver(B, N, batch1.p_max, np_dt, device=dev.index or 0)
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    for h, st in zip(hs + [one], streams):
        h.set_stream(st.cuda_stream)
    din = devmem.DeviceBatch(batch1, dev)
    outs = [devmem.DeviceResult(B, N, np_dt, dev) for _ in range(2)]
    ref = devmem.DeviceResult(B, N, np_dt, dev)
    out = {"handles": 2, "batches": batches, "flag": "DIRECT_FLAG_YIELD"}
    same_all = True
    for name, p in (("fixed", params_fixed), ("natural_exit", abi.phase1_params())):
        one.solve_device(p, din.cin, ref.cout)
        torch.cuda.synchronize()
        want = {k: v.clone() for k, v in ref.t.items()}
        its = int(want["fwd_passes"].sum().item())
        rec = {}
        for mode in ("serial", "pipelined"):
            pick = (lambda i: hs[i % 2]) if mode == "pipelined" else (lambda i: one)
            for i in range(2):
                pick(i).solve_device(p, din.cin, outs[i].cout)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(batches):
                pick(i).solve_device(p, din.cin, outs[i % 2].cout)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            same = all(torch.equal(outs[j].t[k], want[k]) for j in range(2) for k in want)
            same_all = same_all and same
            rec[mode] = {"iter_per_s": its * batches / dt, "ms_per_batch": dt * 1e3 / batches}
        rec["iterations_per_batch"] = its
        out[name] = rec
    out["bit_identical_to_serial"] = bool(same_all)
    out["sched_error"] = int(any(h.sched_error() for h in hs + [one]))
    for h in hs + [one]:
        h.close()
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def dry_run(args, cfg, rank, world, torch, dist, distributed, problems):
    """The N > 1 plumbing without a device: spawn, gloo rendezvous, position-independent shard generation and
    the config-5 gather on a placeholder cost (the shard's first duration) - nothing is solved or measured."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    B, N = cfg["batch"], cfg["nseg"]
    first = rank * B
    batch = problems.make_batch(cfg["kind"], B, N, seed=1000, first=first)
    cost = batch.T0.sum(axis=1)
    li, lc = distributed.local_best(cost, np.zeros(B, np.int32))
    block = torch.from_numpy(np.concatenate([batch.T0[li], batch.seeds[li].ravel()]))
    bc, bidx, owner, blk = distributed.gather_best(lc, first + li, block)
    if rank == 0:
        print(json.dumps({"metric": "ddp_iterations_per_sec", "value": None, "dry": True, "n_gpus": world, "scaling": args.scaling,
                          "batch_per_gpu": B,
                          "dist_world_size": dist.get_world_size() if world > 1 else 1,
                          "gather": {"best_cost": bc, "best_index": bidx, "owner": owner,
                                     "block_checksum": float(blk.double().sum())}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configs[1..4]; 6 = config 4's shape in feasible mode")
    ap.add_argument("--batch", type=int, default=0, help="corridors per GPU (0 = the config's)")
    ap.add_argument("--nseg", type=int, default=0)
    ap.add_argument("--kind", default="", choices=["", "free", "corridor"])
    ap.add_argument("--dtype", default="", choices=["", "f32", "f64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the natural-exit run and the HBM copy microbench")
    ap.add_argument("--with-secondary", action="store_true",
                    help="N > 1 ranks: run the secondary blocks as well (by default a multi-GPU run times the headline step, "
                         "runs the gather, and leaves: ~1 min of single-plan / label-model / cluster blocks on rank 0 would "
                         "keep N - 1 leased GPUs waiting in a barrier)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two rocprofv3 --pmc passes of the same workload in a "
                         "subprocess, ~30 s); the newest committed profile of the workload is quoted instead")
    ap.add_argument("--dry", action="store_true",
                    help="no GPU: exercise launch, rendezvous (gloo), sharding and the gather on placeholder costs; "
                         "prints a line with value null (CPU test of the N > 1 plumbing, never a measurement)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="0 = max(512, 256 x usable host threads), capped by the batch")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the config's batch PER GPU (default, SURVEY.md 8e); strong: a fixed TOTAL of --total corridors "
                         "split evenly over the GPUs")
    ap.add_argument("--total", type=int, default=131072, help="total corridors of a strong-scaling run (config 5's 131072)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher of N ranks (one per GPU, RCCL)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)

    cfg = dict(CONFIGS[args.config])
    for k in ("batch", "nseg", "kind", "dtype"):
        if getattr(args, k):
            cfg[k] = getattr(args, k)
    if args.scaling == "strong":  # fixed total, B / G per rank (shards stay aligned to the generator's chunk of 256)
        if args.total % (args.gpus * 256):
            raise SystemExit("--total must be a multiple of 256 x --gpus")
        cfg["batch"] = args.total // args.gpus
    custom = any(cfg[k] != CONFIGS[args.config][k] for k in ("batch", "nseg", "kind", "dtype"))

    import torch  # first: the library then binds to the HIP runtime torch has already loaded
    import torch.distributed as dist
    from direct_amd import abi, devmem, distributed, problems, solver

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry:
        return dry_run(args, cfg, rank, world, torch, dist, distributed, problems)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the library has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # DIRECT_BENCH_FORCE_DIST=1: every branch of the N > 1 path (RCCL process group, the all-reduces, the object
    # broadcast, the library's communicator) runs at world size 1 too - so that each of its lines has executed on a GPU
    # before the first multi-GPU run (tests/test_gpu_configs.py); the timed region is the same
    dist_on = world > 1 or os.environ.get("DIRECT_BENCH_FORCE_DIST", "0") not in ("", "0")
    if world > 1 and not args.with_secondary:
        args.no_secondary = True
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    np_dt = np.float32 if cfg["dtype"] == "f32" else np.float64
    B, N = cfg["batch"], cfg["nseg"]
    first = rank * B  # weak scaling: rank r solves problems [r*B, (r+1)*B) of the stream
    batch = problems.make_batch(cfg["kind"], B, N, seed=1000, first=first, dtype=np_dt)
    s = solver.DdpSolver(B, N, batch.p_max, np_dt, device=local)
    s.set_stream(torch.cuda.current_stream().cuda_stream)

    # phase 0 once (untimed) to obtain the warm start of the timed phase-1 workload
    g0 = s.solve(abi.phase0_params(), batch)
    batch1 = batch.phase1_inputs(g0)  # TRP:911-921 as direct_ddp_plan_batch chains it (monomial hand-off: include/direct_ddp.h)
    params = abi.phase1_params(iter_max=FIXED_ITERS, fixed_iters=1)
    workload = {"kind": cfg["kind"], "batch": B, "nseg": N, "dtype": cfg["dtype"], "fixed_iters": FIXED_ITERS}

    # The CPU leg runs BEFORE the GPU's timed region so that the GPU is the last thing busy in the process
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        sample = args.cpu_sample or max(512, 256 * usable_cpus())  # ~10-20 s of host work at ~0.8 k iter/s per thread
        cpu = cpu_baseline(batch1.astype(np.float64), params, min(sample, B))

    dbatch = devmem.DeviceBatch(batch1, dev)  # inputs resident in HBM
    outs = devmem.DeviceResult(B, N, np_dt, dev)
    cin, cout = dbatch.cin, outs.cout

    def step():
        s.solve_device(params, cin, cout)

    def sync_all():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(s.last_kernel_ms()[0])  # HIP events around the hot kernel on its own stream
    sync_all()
    dt = time.perf_counter() - t0
    if dist_on:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if s.sched_error():
        raise SystemExit("the ticket scheduler reported an error: results are invalid")
    launch = s.launch_info()  # how the timed launch was scheduled + the sweep work it executed
    rank_kernel_ms = [float(np.mean(kernel_ms))]
    if dist_on:  # stragglers show: every rank's mean kernel time
        km = torch.zeros(world, dtype=torch.float64, device=dev)
        km[rank] = rank_kernel_ms[0]
        dist.all_reduce(km, op=dist.ReduceOp.SUM)
        rank_kernel_ms = [float(v) for v in km.tolist()]
    iters_step = int(outs["fwd_passes"].sum().item())
    # the mode every trajectory's timed iterations ran in (taken now: the secondary solves reuse the output arrays)
    infeas_mask = outs["infeas_out"].cpu().numpy().astype(bool)
    total = torch.tensor([float(iters_step)], dtype=torch.float64, device=dev)
    if dist_on:
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
    iters_all = float(total.item()) * args.steps

    # config-5 reduction (functional check, untimed): cheapest feasible trajectory on every rank.
    # (a) through torch.distributed, (b) through the library's C entry point on its own RCCL communicator.
    cost_h, rtn_h = outs["cost"].cpu().numpy(), outs["rtn"].cpu().numpy()
    li, lc = s.best_cost(outs["cost"].data_ptr(), outs["rtn"].data_ptr(), mem=abi.MEM_DEVICE, batch=B)
    hi, hc = distributed.local_best(cost_h, rtn_h)
    assert li == hi, (li, hi)
    blk_src = max(li, 0)
    block = torch.cat([outs["bez"][blk_src].reshape(-1), outs["T"][blk_src].reshape(-1)])
    tg = time.perf_counter()
    bc, bidx, owner, blk = distributed.gather_best(lc, first + li if li >= 0 else -1, block, force_collective=dist_on)
    torch.cuda.synchronize()
    gather = {"torch_ms": (time.perf_counter() - tg) * 1e3, "best_cost": bc, "best_index": bidx, "owner": owner,
              "forced_dist_path": bool(dist_on and world == 1)}
    devices = [local]
    if dist_on:  # every torch.distributed exchange the line needs happens BEFORE the library's own communicator is tried
        got = [None] * world
        dist.all_gather_object(got, (rank, local, torch.cuda.get_device_name(local)))
        devices = got
        gather["dist_world_size"] = dist.get_world_size()
    uid = [None]
    try:
        uid = [s.rccl_unique_id() if rank == 0 else None]
    except Exception as e:  # noqa: BLE001
        gather.update({"c_abi_error": repr(e)})
    if dist_on:
        dist.broadcast_object_list(uid, src=0)  # None: rank 0 could not talk to RCCL, nobody tries

    def c_abi_gather(res):
        """(b): the library's C entry point on its own RCCL communicator.  Runs in a thread with a deadline: a
        communicator that cannot be built must cost the run this block, never the bench line."""
        try:
            comm = s.rccl_comm_create(uid[0], world, rank)
            wb = torch.zeros(N, 18, dtype=outs["bez"].dtype, device=dev)
            wT = torch.zeros(N, dtype=outs["T"].dtype, device=dev)
            for rep in range(2):  # the second call is the timed one (the first builds RCCL's channels)
                tg = time.perf_counter()
                ci, cc, cown = s.gather_best(comm, world, rank, outs["cost"].data_ptr(), outs["rtn"].data_ptr(),
                                             outs["bez"].data_ptr(), outs["T"].data_ptr(), first, mem=abi.MEM_DEVICE, batch=B,
                                             out_bez=wb.data_ptr(), out_T=wT.data_ptr())
                c_ms = (time.perf_counter() - tg) * 1e3
            s.rccl_comm_destroy(comm)
            same = (ci == bidx and cc == bc and cown == owner
                    and bool(torch.equal(torch.cat([wb.reshape(-1), wT.reshape(-1)]), blk.to(dev))))
            res.update({"c_abi_rccl_ms": c_ms, "c_abi_matches_torch": bool(same), "rccl_ranks": world})
        except Exception as e:  # noqa: BLE001 - reported, not hidden
            res.update({"c_abi_error": repr(e)})

    # RCCL writes a version banner to the C stdout when a communicator is created: keep this process's stdout
    # for the ONE JSON line by pointing fd 1 at stderr while the library talks to RCCL
    rccl_hung = False
    if uid[0] is not None:
        import ctypes
        import threading
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        res = {}
        th = threading.Thread(target=c_abi_gather, args=(res,), daemon=True)
        th.start()
        th.join(timeout=120.0)
        rccl_hung = th.is_alive()
        ctypes.CDLL(None).fflush(None)
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
        gather.update(res if not rccl_hung else
                      {"c_abi_error": "no answer from the library's RCCL communicator within 120 s; secondaries skipped"})
    if rccl_hung:  # the handle's stream may be blocked behind a collective that never completes: nothing more runs on it
        args.no_secondary = True

    # secondary: the same step sustained for 100 more launches (~4 s of uninterrupted GPU work: clocks settle, and an
    # outside sampler such as the driver's SMI poll gets to see the device busy; the headline value is NOT taken here)
    sustained = None
    if not args.no_secondary:
        ts = time.perf_counter()
        for _ in range(100):
            step()
        torch.cuda.synchronize()
        sustained = (time.perf_counter() - ts) * 1e3 / 100
    # secondary: the SAME fixed-iteration solve with host buffers in and out (what a reference-side caller of the C-ABI
    # sees: H2D of the corridors, k_begin, the hot kernel, k_finish, D2H of the results) - PCIe-inclusive, never `value`
    e2e = None
    if not args.no_secondary:
        s.solve(params, batch1)  # first call: page-in of the host arrays
        te = time.perf_counter()
        reps = 3
        for _ in range(reps):
            he = s.solve(params, batch1)
        e2e_s = (time.perf_counter() - te) / reps
        e2e = {"iter_per_s": float(he.fwd_passes.sum() / e2e_s), "ms_per_call": e2e_s * 1e3,
               "what": "direct_ddp_solve_batch on host (numpy) arrays: pageable H2D + k_begin + hot kernel + k_finish + D2H, "
                       "mean of %d calls" % reps}
    # secondary: the same batch with the reference's natural exits (DDP:335-396)
    natural, hbm_copy = None, None
    if not args.no_secondary:
        pn = abi.phase1_params()
        s.solve_device(pn, cin, cout)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        s.solve_device(pn, cin, cout)
        torch.cuda.synchronize()
        dtn = time.perf_counter() - t1
        fp, rt = outs["fwd_passes"].cpu().numpy(), outs["rtn"].cpu().numpy()
        natural = {"iter_per_s": float(fp.sum() / dtn), "ms": dtn * 1e3, "kernel_ms": s.last_kernel_ms()[0],
                   "iterations_mean": float(fp.mean()), "iterations_p50": float(np.median(fp)),
                   "iterations_max": int(fp.max()), "iterations_min": int(fp.min()),
                   "rtn_histogram": {str(int(v)): int(c) for v, c in zip(*np.unique(rt, return_counts=True))}}
        hbm_copy = hbm_copy_gbs(torch, dev)
    label, clusters, single, pipelined = None, None, None, None
    if not args.no_secondary and rank == 0:
        try:
            pipelined = pipelined_line(torch, dev, batch1, np_dt, params)
        except Exception as ex:  # a secondary block never takes the line down
            pipelined = {"error": str(ex)[:200]}
        try:
            single = single_plan_line(dev, not args.no_cpu_baseline and world == 1)
        except Exception as ex:  # a secondary block never takes the line down
            single = {"error": str(ex)[:200]}
        label = label_model_line(torch, dev, B, not args.no_cpu_baseline and world == 1)
        try:
            clusters = corridor_clusters_line(dev, not args.no_cpu_baseline and world == 1, live=not args.no_live_traffic)
        except Exception as ex:  # a secondary block never takes the line down
            clusters = {"error": str(ex)[:200]}

    if rank == 0:
        # SURVEY.md 8(d)'s algorithmic words, priced on the sweep work the launch really EXECUTED: every backward knot
        # visit at the backward figure (119 + 2 nc + 4 P), and at most ONE forward trial-knot per backward knot visit
        # ("one accepted line-search trial") at the forward figure (138 + 3 nc + 4 P) - but never more forward knots
        # than were executed: in infeasible mode the fraction-to-boundary rule cuts most trials short after a few
        # knots, and the unit-formula (one full forward trial per iteration) then over-counts by 2 x (config 4, r03).
        # A trajectory that is still infeasible after the fixed iterations moves y / ky as well: the nc terms double.
        wb, wf = problems.algorithmic_words(batch1.n_planes, batch1.n_seg, infeasible=infeas_mask, split=True)
        knots = int(batch1.n_seg.sum())
        isz = np.dtype(np_dt).itemsize
        fwd_counted = min(launch["fwd_knot_visits"], launch["bwd_knot_visits"])
        bytes_per_launch = (launch["bwd_knot_visits"] * (wb / knots) + fwd_counted * (wf / knots)) * isz
        bytes_unit_formula = (wb + wf) * isz * FIXED_ITERS  # one launch = FIXED_ITERS iterations of B corridors, one full trial each
        avg_ms = float(np.mean(kernel_ms))
        traffic, traffic_kind = None, None
        if world == 1 and not args.no_live_traffic and not args.no_secondary:
            traffic = live_hbm_traffic(workload)
            if traffic is not None:  # the PMC passes must have run the launch that was timed above (10 %)
                if max(abs(m / avg_ms - 1.0) for m in traffic["kernel_ms_under_pmc"]) > 0.10:
                    traffic = None
                else:
                    traffic["_file"] = None
                    traffic_kind = "live: FETCH_SIZE / WRITE_SIZE passes of this workload on this box, in this run (tools/prof_one.py under rocprofv3)"
        if traffic is None:
            traffic = matching_profile("r*_hbm_traffic.json", workload)
            traffic_kind = None if traffic is None else "static: PMC counters of a committed profile of this workload, not measured in this run"
        # never a numerator above what the counters saw cross the HBM interface (the kernels move less than SURVEY's
        # figure where they can: the slack / dual gains never leave the wave)
        numer, numer_kind = bytes_per_launch, "algorithmic bytes of the executed sweep work (SURVEY.md 8d words x executed knot visits)"
        if traffic is not None and traffic["traffic_bytes_per_launch"] < numer:
            numer, numer_kind = float(traffic["traffic_bytes_per_launch"]), "measured HBM traffic (below the algorithmic figure)"
        achieved = numer / (avg_ms * 1e-3) / 1e9
        value = iters_all / dt
        sq, sq_source = None, None
        if world == 1 and not args.no_live_traffic and not args.no_secondary:
            sq = live_sq_counters(workload)
            if sq is not None and max(abs(m / avg_ms - 1.0) for m in sq["kernel_ms_under_pmc"]) > 0.10:
                sq = None
            if sq is not None:
                sq_source = "live"
        if sq is None:
            sq = matching_profile("r*_sq_counters.json", workload)
            sq_source = None if sq is None else "static: " + sq["_file"]
        fwd_passes_launch = max(1, iters_step)
        accepted = launch.get("accepted_line_searches")
        line = {
            "metric": "ddp_iterations_per_sec", "value": value, "unit": "iter/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            # arithmetic type of the path: double for both storage types
            "dtype": "f64", "storage_dtype": cfg["dtype"], "data": "synthetic",
            "config": {"workload": "%s%s: %d %s corridors per GPU, N=%d segments, %s storage, polynomial-segment IPDDP "
                                   "(9 states / 10 controls), phase-1 weights, fixed %d iterations, warm start from phase 0"
                                   % (cfg["name"], " (modified)" if custom else "", B,
                                      "free-space" if cfg["kind"] == "free" else "polyhedron", N, cfg["dtype"], FIXED_ITERS),
                       "batch_per_gpu": B, "n_seg": N, "fixed_iters": FIXED_ITERS, "parallelism": "shard%d" % world,
                       "storage_dtype": cfg["dtype"], "kind": cfg["kind"], "devices": devices,
                       "total_corridors": B * world,
                       # what the library actually did with this launch (direct_ddp_last_launch_info, rank 0)
                       "schedule": {k: launch.get(k) for k in ("dynamic", "shared_search", "shared_sweep", "pair_trials", "single_steps",
                                                               "n_buffers", "resident_waves")}},
            # bound "hbm" is the roofline BASELINE.json's north_star stipulates; the counters say the kernel is
            # instruction-issue bound, which roofline_compute prices
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None if traffic is None else traffic["traffic_bytes_per_launch"],
                         "traffic_kind": traffic_kind,
                         "traffic_source": None if traffic is None else traffic["_file"],
                         "traffic_over_algorithmic": None if traffic is None else traffic["traffic_bytes_per_launch"] / bytes_per_launch,
                         "peak_measured_copy": hbm_copy, "frac_of_measured_copy": None if not hbm_copy else achieved / hbm_copy,
                         "kernel": "k_iterate_dyn (ticket-scheduled k_iterate)", "kernel_ms": avg_ms,
                         "numerator": numer_kind,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         # the same words under the unit formula's assumption (one FULL forward trial per iteration)
                         "algorithmic_bytes_unit_formula": bytes_unit_formula,
                         "infeasible_mode_frac": float(infeas_mask.mean()),
                         # what the iterations of the timed launch were: line searches that accepted a step / that ended
                         # with fp_failed (DDP:760-762: the iterate does not move), forward trial-knots per backward knot
                         "accepted_step_frac": None if accepted is None else accepted / fwd_passes_launch,
                         "fp_failed_frac": None if accepted is None else 1.0 - accepted / fwd_passes_launch,
                         "fwd_per_bwd_knot": launch["fwd_knot_visits"] / max(1, launch["bwd_knot_visits"]),
                         # sweep work the launch really executed (forward trials cut short by the fraction-to-boundary
                         # rule count the knots they reached)
                         "bwd_knot_visits": launch["bwd_knot_visits"], "fwd_trial_knot_visits": launch["fwd_knot_visits"],
                         "algorithmic_knot_iterations": int(batch1.n_seg.sum()) * FIXED_ITERS,
                         "traffic_bytes_per_executed_knot_visit": None if traffic is None else
                         traffic["traffic_bytes_per_launch"] / max(1, launch["bwd_knot_visits"] + launch["fwd_knot_visits"])},
            "kernel_ms_per_rank": rank_kernel_ms,
            "kernel_ms_rank_spread": (float(max(rank_kernel_ms) - min(rank_kernel_ms)) / float(np.mean(rank_kernel_ms))) if rank_kernel_ms else None,
            "iters_per_step_rank0": iters_step, "gather": gather,
        }
        if sq is not None and "f64_flops_per_ddp_iteration" in sq["derived"] and "f64_arith_frac_of_valu" in sq["derived"]:
            fl = sq["derived"]["f64_flops_per_ddp_iteration"]
            tf = fl * (iters_step / (avg_ms * 1e-3)) / 1e12
            line["roofline_compute"] = {"bound": "fp64 vector", "achieved_tflops_f64": tf, "peak": F64_VECTOR_PEAK_TF,
                                        "frac": tf / F64_VECTOR_PEAK_TF, "valu_busy": sq["derived"]["valu_busy_frac_per_simd"],
                                        "f64_arith_frac_of_valu": sq["derived"]["f64_arith_frac_of_valu"],
                                        "valu_insts_per_ddp_iteration": sq["derived"]["valu_insts_per_ddp_iteration"],
                                        "lds_busy": sq["derived"].get("lds_busy_frac_per_cu"),
                                        "lds_bank_conflict_frac_of_lds_cycles": sq["derived"].get("lds_bank_conflict_frac_of_lds_cycles"),
                                        "source": sq_source,
                                        "note": "flops per DDP iteration counted by rocprofv3 (SQ_INSTS_VALU_{FMA,MUL,ADD}_F64 of the "
                                                "timed launch, all 64 lanes of an instruction counted) x this run's kernel rate"}
        if sustained is not None:
            line["sustained_ms_per_step_100"] = sustained
        if e2e is not None:
            line["e2e_host_buffers"] = e2e
        if natural is not None:
            line["natural_exit"] = natural
        if pipelined is not None:
            line["pipelined"] = pipelined
        if single is not None:
            line["single_plan_latency"] = single
        if label is not None:
            line["label_model"] = label
            line["corridor_clusters"] = clusters
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if rccl_hung:  # a thread is stuck inside RCCL: an orderly shutdown would wait for it
        sys.stderr.flush()
        os._exit(0)
    s.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
