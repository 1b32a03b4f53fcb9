"""Static instruction counts per phase of the hot kernel (needs a -DDDP_MARKS -S build)."""
import re, sys
txt = open(sys.argv[1]).read()
kern = sys.argv[2] if len(sys.argv) > 2 else "_Z9k_iterateIfLi2E"
m = re.search(r"^%s\w*:.*?\.Lfunc_end\d+:" % kern, txt, re.S | re.M)
lines = m.group(0).split("\n")
cur, stats, order = "PROLOGUE", {}, []
for l in lines:
    mm = re.search(r"; DDP_MARK (\w+)", l)
    if mm:
        cur = mm.group(1)
        continue
    t = re.match(r"\s+([a-z_0-9]+)", l)
    if not t:
        continue
    op = t.group(1)
    if cur not in stats:
        stats[cur] = dict(v=0, s=0, ds=0, gl=0, rl=0, div=0, f64=0, wait=0); order.append(cur)
    d = stats[cur]
    if op.startswith("v_"): d["v"] += 1
    elif op.startswith("s_"): d["s"] += 1
    elif op.startswith("ds_"): d["ds"] += 1
    elif op.startswith("global_") or op.startswith("buffer_"): d["gl"] += 1
    if "readlane" in op or "writelane" in op: d["rl"] += 1
    if op.startswith(("v_div", "v_rcp", "v_rsq", "v_sqrt", "v_log", "v_exp", "v_ldexp", "v_frexp")): d["div"] += 1
    if "f64" in op: d["f64"] += 1
    if op == "s_waitcnt": d["wait"] += 1
print("%-9s %5s %5s %4s %3s %5s %4s %4s %4s" % ("phase", "VALU", "SALU", "DS", "GL", "lanes", "div", "f64", "wait"))
for k in order:
    d = stats[k]
    print("%-9s %5d %5d %4d %3d %5d %4d %4d %4d" % (k, d["v"], d["s"], d["ds"], d["gl"], d["rl"], d["div"], d["f64"], d["wait"]))
