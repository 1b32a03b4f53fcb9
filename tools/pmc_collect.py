"""Aggregates rocprofv3 --pmc passes (csv output) of ONE workload into the JSON files bench.py reads.

usage: pmc_collect.py <passes_dir> <out_prefix> <kind> <dtype> <batch> <nseg> <fixed_iters>
<passes_dir>/<pass>/ holds one rocprofv3 run each (counter_collection.csv + kernel_trace.csv).  The timed launch is
selected BY POSITION: tools/prof_one.py runs phase 0 first and the fixed-iteration phase-1 solve last, so the timed
launch is the LAST k_iterate* dispatch of the process (r02 selected "the largest dispatches", which picked the
50-iteration phase-0 launch wherever phase 0 never reaches feasibility: config 4).  The selection is then CHECKED:
the kernel-trace duration of the selected dispatch must agree within 10 % with the "kernel ms" line the same process
printed from its HIP events (<out_prefix>_pmc_run<i>.log), and - when <out_prefix>_plain.log exists - with the
un-profiled run; otherwise the script exits non-zero and writes nothing.  Writes <out_prefix>_sq_counters.json and, when FETCH_SIZE / WRITE_SIZE passes are
present, <out_prefix>_hbm_traffic.json (units and corrections: MI355X_MICROARCH.md, HBM section; the
calibration factors for THIS access width come from <out_prefix>_hbm_calib.json when it exists)."""
import csv
import glob
import json
import os
import sys

d, prefix, kind, dtype, batch, nseg, iters = sys.argv[1:8]
batch, nseg, iters = int(batch), int(nseg), int(iters)
workload = {"kind": kind, "batch": batch, "nseg": nseg, "dtype": dtype, "fixed_iters": iters}
counters, kernel_ms = {}, []
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "k_iterate" in r["Kernel_Name"]]
    by = {}
    for r in rows:
        by.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        by[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for name, disp in by.items():
        last = max(disp, key=int)  # the last k_iterate* dispatch of the process = the timed launch
        counters[name] = disp[last]
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "k_iterate" in r["Kernel_Name"]]
    if rows:
        r = max(rows, key=lambda q: int(q["Dispatch_Id"]))
        kernel_ms.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)


def printed_ms(path):
    """the `kernel ms <x>` line of tools/prof_one.py (HIP events around the timed launch, same process)"""
    try:
        for l in open(path):
            if l.startswith("kernel ms"):
                return float(l.split()[2])
    except OSError:
        pass
    return None


event_ms = [m for m in (printed_ms(f) for f in sorted(glob.glob(prefix + "_pmc_run*.log"))) if m is not None]
plain_ms = printed_ms(prefix + "_plain.log")
if not kernel_ms or not event_ms:
    sys.exit("pmc_collect: no k_iterate dispatch / no 'kernel ms' line found - nothing written")
worst = max(abs(t / e - 1.0) for t, e in zip(kernel_ms, event_ms))
if worst > 0.10:
    sys.exit("pmc_collect: selected dispatch %s ms vs HIP-event time %s ms of the same runs: wrong launch selected"
             % (kernel_ms, event_ms))
if plain_ms is not None and max(abs(t / plain_ms - 1.0) for t in kernel_ms) > 0.10:
    sys.exit("pmc_collect: kernel ms under PMC %s vs un-profiled %.2f ms differ by more than 10 %%" % (kernel_ms, plain_ms))
c = counters
out = {"command": "rocprofv3 --pmc <counters of one pass> --kernel-trace --output-format csv -- python tools/prof_one.py "
                  "%s %s %d %d %d (separate passes; the fixed-iteration phase-1 launch of bench.py's workload)"
                  % (kind, dtype, batch, nseg, iters),
       "workload": workload, "counters": c, "kernel_ms_under_pmc": kernel_ms, "kernel_ms_hip_events_same_runs": event_ms,
       "kernel_ms_unprofiled": plain_ms,
       "dispatch_selection": "last k_iterate* dispatch of tools/prof_one.py; checked against the HIP-event time (10 %)"}
ddp_iters = float(batch * iters)
der = {}
if "SQ_INSTS_VALU" in c:
    der["valu_insts_per_ddp_iteration"] = c["SQ_INSTS_VALU"] / ddp_iters
if "SQ_INSTS_LDS" in c:
    der["lds_insts_per_ddp_iteration"] = c["SQ_INSTS_LDS"] / ddp_iters
if all(k in c for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU")):
    arith = c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_ADD_F64"]
    der["f64_arith_frac_of_valu"] = arith / c["SQ_INSTS_VALU"]
    # flops with all 64 lanes of every instruction counted (an upper bound: lane utilisation is < 1)
    der["f64_flops_per_ddp_iteration"] = (2 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_MUL_F64"]
                                          + c["SQ_INSTS_VALU_ADD_F64"]) * 64 / ddp_iters
if all(k in c for k in ("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CU_CYCLES")):
    # SQ_ACTIVE_INST_* count quad-cycles summed over the SIMDs of a CU; SQ_BUSY_CU_CYCLES cycles per CU
    der["valu_busy_frac_per_simd"] = c["SQ_ACTIVE_INST_VALU"] * 4 / (c["SQ_BUSY_CU_CYCLES"] * 4)
if all(k in c for k in ("SQ_LDS_IDX_ACTIVE", "SQ_BUSY_CU_CYCLES")):
    der["lds_busy_frac_per_cu"] = c["SQ_LDS_IDX_ACTIVE"] / c["SQ_BUSY_CU_CYCLES"]
if all(k in c for k in ("SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU")):
    der["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64)
if all(k in c for k in ("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")):
    der["wait_inst_any_frac"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
if all(k in c for k in ("SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES")):
    der["wait_inst_lds_frac"] = c["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"]
out["derived"] = der
out["units"] = ("SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles, SQ_LDS_IDX_ACTIVE and SQ_BUSY_CU_CYCLES "
                "count cycles (MI355X_MICROARCH.md)")
json.dump(out, open(prefix + "_sq_counters.json", "w"), indent=1)
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    cal = {"fetch_factor_dword": 1.0, "write_factor_dword": 1.0}
    cf = prefix + "_hbm_calib.json"
    if os.path.exists(cf):
        cal.update({k: v for k, v in json.load(open(cf)).items() if k in cal})
    fetch_b = c["FETCH_SIZE"] * 1024.0 / cal["fetch_factor_dword"]
    write_b = c["WRITE_SIZE"] * 1024.0 / cal["write_factor_dword"]
    json.dump({"workload": workload, "FETCH_SIZE_kb": c["FETCH_SIZE"], "WRITE_SIZE_kb": c["WRITE_SIZE"],
               "calibration": cal, "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
               "traffic_bytes_per_launch": fetch_b + write_b,
               "note": "counter x 1024 B / (counter bytes per true byte measured by tools/hbm_calib for dword-per-lane "
                       "streams, separate --pmc passes); per k_iterate launch of the timed workload"},
              open(prefix + "_hbm_traffic.json", "w"), indent=1)
print(json.dumps(der))
