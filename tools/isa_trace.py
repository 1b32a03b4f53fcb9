"""Dynamic instruction count per phase of ONE trip through a loop of the hot kernel, from a -DDDP_MARKS -S build.

usage: isa_trace.py file.s <kernel-prefix> <start-mark> [taken-label ...] [--p=TRIPS]
Walks the ISA from the first `; DDP_MARK <start-mark>` until that mark comes round again (or the function ends),
following fall-through at every s_cbranch_exec* (a lane-mask skip: the block runs unless no lane is active) and, at
uniform branches (scc / vcc), the direction given on the command line: LABEL = always taken, LABEL@N = taken at its Nth
encounter only (loop exits), every other conditional branch falls through.  Backward branches are loops: taken `TRIPS - 1` times (--p, default 3: the
plane loop of phase S for P = 6 at unroll 2).  Prints the uniform branches it met (so that the policy can be reviewed)
and VALU / SALU / LDS / VMEM counts per phase."""
import re
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--"))
path, kern, start = args[0], args[1], args[2]
taken = {}
for a in args[3:]:  # LABEL (always taken) or LABEL@N (taken at its Nth encounter only)
    lab, _, nth = a.partition("@")
    taken[lab] = int(nth) if nth else 0
seen = {}
trips = int(opts.get("p", 3))
txt = open(path).read()
m = re.search(r"^%s\w*:.*?\.Lfunc_end\d+:" % kern, txt, re.S | re.M)
lines = m.group(0).split("\n")
label_at = {}
for i, l in enumerate(lines):
    mm = re.match(r"^(\.LBB\d+_\d+):", l)
    if mm:
        label_at[mm.group(1)] = i
pc = [i for i, l in enumerate(lines) if "; DDP_MARK " + start in l][int(opts.get("occ", 1)) - 1]  # --occ=N: Nth occurrence of the mark
start_pc = pc
phase, stats, order, met, loops = start, {}, [], [], {}
first = True
steps = 0
while pc < len(lines) and steps < 200000:
    l = lines[pc]
    steps += 1
    mm = re.search(r"; DDP_MARK (\w+)", l)
    if mm:
        if (mm.group(1) == start and not first) or mm.group(1) == opts.get("stop", ""):
            break
        first = False
        phase = mm.group(1)
        pc += 1
        continue
    t = re.match(r"\s+([a-z_0-9]+)\s*(.*)", l)
    if not t:
        pc += 1
        continue
    op, rest = t.group(1), t.group(2)
    d = stats.setdefault(phase, dict(v=0, s=0, ds=0, gl=0, f64=0, cvt=0, rl=0, mov=0, cnd=0, i32=0))
    if phase not in order:
        order.append(phase)
    if op.startswith("v_"):
        d["v"] += 1
        if "f64" in op and not op.startswith("v_cvt") and not op.startswith("v_cmp"):
            d["f64"] += 1
        if op.startswith("v_cvt"):
            d["cvt"] += 1
        if "readlane" in op or "readfirstlane" in op or "dpp" in op:
            d["rl"] += 1
        if op.startswith("v_mov") or op.startswith("v_accvgpr"):
            d["mov"] += 1
        if op.startswith("v_cndmask"):
            d["cnd"] += 1
        if re.match(r"v_(add|sub|mul|mad|lshl|lshr|ashr|and|or|xor|min|max|bfe|bfi|add3|lshl_add|lshl_or|and_or|or3|subrev|mul_u32|mul_lo|mul_hi)_?[a-z]*_?[iu](32|24|16)", op) or op in ("v_lshlrev_b32_e32", "v_lshrrev_b32_e32", "v_and_b32_e32", "v_or_b32_e32", "v_lshl_add_u32", "v_lshl_or_b32", "v_and_or_b32", "v_or3_b32", "v_add3_u32", "v_xor_b32_e32", "v_bfe_u32", "v_ashrrev_i32_e32"):
            d["i32"] += 1
    elif op.startswith("s_"):
        d["s"] += 1
    elif op.startswith("ds_"):
        d["ds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        d["gl"] += 1
    if op == "s_branch":
        pc = label_at[rest.strip()]
        continue
    if op.startswith("s_cbranch"):
        tgt = rest.strip()
        back = label_at[tgt] < pc
        if back and label_at[tgt] >= start_pc:
            n = loops.get(tgt, 0)
            if n < trips - 1:
                loops[tgt] = n + 1
                pc = label_at[tgt]
                continue
            loops[tgt] = 0
        elif "exec" in op:
            if back and op.endswith("execnz"):  # a back edge over the lane mask (outer loop header): follow it
                pc = label_at[tgt]
                continue
        else:
            seen[tgt] = seen.get(tgt, 0) + 1
            take = tgt in taken and (taken[tgt] == 0 or taken[tgt] == seen[tgt])
            met.append((phase, op, tgt, take))
            if take:
                pc = label_at[tgt]
                continue
    if op == "s_endpgm" or op == "s_setpc_b64":
        break
    pc += 1
print("uniform branches met (phase, op, target, taken):")
import collections
for x, c in collections.Counter(met).items():
    print("  ", x, "x%d" % c)
print("%-8s %5s %5s %5s %4s | %5s %4s %5s %4s %4s %4s" % ("phase", "VALU", "SALU", "LDS", "VMEM", "f64", "cvt", "lane", "mov", "cnd", "i32"))
tot = dict(v=0, s=0, ds=0, gl=0, f64=0, cvt=0, rl=0, mov=0, cnd=0, i32=0)
for k in order:
    d = stats[k]
    for q in tot:
        tot[q] += d[q]
    print("%-8s %5d %5d %5d %4d | %5d %4d %5d %4d %4d %4d" % (k, d["v"], d["s"], d["ds"], d["gl"], d["f64"], d["cvt"], d["rl"], d["mov"], d["cnd"], d["i32"]))
d = tot
print("%-8s %5d %5d %5d %4d | %5d %4d %5d %4d %4d %4d" % ("TOTAL", d["v"], d["s"], d["ds"], d["gl"], d["f64"], d["cvt"], d["rl"], d["mov"], d["cnd"], d["i32"]))
