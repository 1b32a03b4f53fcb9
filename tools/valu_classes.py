"""Static classes of the VALU opcodes per phase of the hot kernel (a -DDDP_MARKS -S build: tools/isa_stats.sh out.s).
usage: valu_classes.py out.s [kernel-prefix] [out.json].  Classes: f64 arithmetic (fma / mul / add / min-max / transcendental
seeds), conversions, integer + address arithmetic, selects and compares, moves (plain, DPP, accvgpr), lane reads."""
import collections, json, re, sys
path = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "_Z13k_iterate_dynIfLi2E"
txt = open(path).read()
m = re.search(r"^%s\w*:.*?\.Lfunc_end\d+:" % kern, txt, re.S | re.M)


def cls(op, line):
    if "readlane" in op or "readfirstlane" in op or "writelane" in op: return "lane_read"
    if op.startswith("v_cvt"): return "convert"
    if op.startswith(("v_cndmask", "v_cmp", "v_cmpx")): return "f64_compare" if "f64" in op else "select_compare"
    if op.startswith(("v_mov", "v_accvgpr", "v_swap", "v_perm", "v_bfi", "v_alignbit")): return "move_dpp" if "dpp" in line else "move"
    if "f64" in op:
        if op.startswith(("v_fma", "v_fmac")): return "f64_fma"
        if op.startswith("v_mul"): return "f64_mul"
        if op.startswith("v_add"): return "f64_add"
        return "f64_other"
    if "f32" in op or "f16" in op: return "f32"
    return "int_address"


phases, order = collections.defaultdict(collections.Counter), []
ops = collections.defaultdict(collections.Counter)
cur = "PROLOGUE"
for l in m.group(0).split("\n"):
    mm = re.search(r"; DDP_MARK (\w+)", l)
    if mm:
        cur = mm.group(1)
        continue
    t = re.match(r"\s+(v_[a-z_0-9]+)", l)
    if not t: continue
    if cur not in order: order.append(cur)
    c = cls(t.group(1), l)
    phases[cur][c] += 1
    if not c.startswith("f64_") or c == "f64_compare": ops[cur][t.group(1) + (".dpp" if "dpp" in l else "")] += 1
cols = ["f64_fma", "f64_mul", "f64_add", "f64_other", "f64_compare", "convert", "int_address", "select_compare", "move", "move_dpp", "lane_read", "f32"]
print("%-9s %5s " % ("phase", "VALU") + " ".join("%7s" % c[:7] for c in cols))
groups = {"backward_knot": ["B_T2", "B_R1", "B_S", "B_S2", "B_H", "B_C", "B_G", "B_R2"], "forward_round_knot": ["F_D", "F_T", "F_R"]}
out = {"kernel": kern, "source": "static count per phase of the code object's ISA (one trip; phase S's plane loop counted once)", "phases": {}, "groups": {}}
for p in order:
    print("%-9s %5d " % (p, sum(phases[p].values())) + " ".join("%7d" % phases[p][c] for c in cols))
    out["phases"][p] = dict(valu=sum(phases[p].values()), classes=dict(phases[p]), top_non_arithmetic=ops[p].most_common(8))
for g, ps in groups.items():
    tot = collections.Counter()
    top = collections.Counter()
    for p in ps:
        tot.update(phases[p]); top.update(ops[p])
    n = sum(tot.values())
    print("%-9s %5d " % (g[:9], n) + " ".join("%7d" % tot[c] for c in cols))
    out["groups"][g] = dict(valu=n, classes=dict(tot), f64_arith_share=round(sum(tot[c] for c in cols[:4]) / max(n, 1), 3), top_non_arithmetic=top.most_common(12))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
