"""FETCH_SIZE / WRITE_SIZE of tools/hbm_calib's kernels (two rocprofv3 --pmc passes, csv) against their known byte
counts -> <out>_hbm_calib.json: counter bytes per true byte for dword- and dwordx4-per-lane streams, plus the
achieved GB/s of a plain run (the measured HBM peak).
usage: hbm_calib_collect.py <fetch_pass_dir> <write_pass_dir> <plain_run_log> <out_prefix>"""
import csv
import glob
import json
import os
import sys

fd, wd, log, prefix = sys.argv[1:5]
plain = [json.loads(l) for l in open(log) if l.startswith("{")]
known = {p["kernel"]: p["bytes_per_launch"] for p in plain}


def mean_counter(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    by = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            k = r["Kernel_Name"].split("(")[0]
            by.setdefault(k, {}).setdefault(r["Dispatch_Id"], 0.0)
            by[k][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: sum(v.values()) / len(v) for k, v in by.items()}


fetch, write = mean_counter(fd, "FETCH_SIZE"), mean_counter(wd, "WRITE_SIZE")
out = {"plain_run": plain, "FETCH_SIZE_kb": fetch, "WRITE_SIZE_kb": write}
for name, src, kern in (("fetch_factor_dword", fetch, "calib_read_dword"), ("fetch_factor_dwordx4", fetch, "calib_read_dwordx4"),
                        ("write_factor_dword", write, "calib_write_dword"), ("write_factor_dwordx4", write, "calib_write_dwordx4")):
    match = [v for k, v in src.items() if k.startswith(kern) and not k.startswith(kern + "x")]
    if match and kern in known:
        out[name] = match[0] * 1024.0 / known[kern]
out["hbm_peak_measured_GBs"] = max(p["GBs"] for p in plain)
json.dump(out, open(prefix + "_hbm_calib.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if "factor" in k or "peak" in k}))
