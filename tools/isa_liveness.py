"""Approximate VGPR liveness over a kernel's gfx9 assembly: max live registers per DDP_MARK phase."""
import re, sys
txt = open(sys.argv[1]).read()
kern = sys.argv[2]
m = re.search(r"^%s\w*:.*?\.Lfunc_end\d+:" % kern, txt, re.S | re.M)
lines = m.group(0).split("\n")
NODEF = ("ds_write", "global_store", "scratch_store", "buffer_store", "v_cmp", "s_", "ds_bpermute_dummy", "global_atomic", "flat_store")
def regs(tok):
    out = []
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok): out += list(range(int(a), int(b) + 1))
    tok2 = re.sub(r"\bv\[\d+:\d+\]", "", tok)
    out += [int(a) for a in re.findall(r"\bv(\d+)\b", tok2)]
    return out
ins = []  # (op, defs, uses, phase, label?)
labels = {}
phase = "PRO"
for l in lines:
    mm = re.search(r"; DDP_MARK (\w+)", l)
    if mm: phase = mm.group(1); continue
    lm = re.match(r"^(\.LBB\w+):", l)
    if lm: labels[lm.group(1)] = len(ins); continue
    l = l.split(";")[0].strip()
    if not l or l.startswith(".") or l.startswith(";"): continue
    parts = l.split(None, 1)
    op = parts[0]; args = parts[1] if len(parts) > 1 else ""
    ops = [a.strip() for a in args.split(",")]
    defs, uses = [], []
    if op.startswith(NODEF) or op.startswith("s_"):
        for o in ops: uses += regs(o)
        if op.startswith("v_cmp") : pass
    elif op.startswith(("v_readlane", "v_readfirstlane")):
        for o in ops[1:]: uses += regs(o)
    else:
        if ops: defs = regs(ops[0])
        for o in ops[1:]: uses += regs(o)
        if op.startswith(("v_mac", "v_fmac", "v_writelane", "v_movrel")) or "dpp" in op or "sdwa" in l: uses += defs
    tgt = None
    if op.startswith("s_cbranch") or op == "s_branch": tgt = ops[0]
    ins.append([op, set(defs), set(uses), phase, tgt])
n = len(ins)
succ = [[] for _ in range(n)]
for i, (op, d, u, ph, tgt) in enumerate(ins):
    if op == "s_branch":
        if tgt in labels: succ[i].append(labels[tgt])
        continue
    if op in ("s_endpgm",): continue
    if i + 1 < n: succ[i].append(i + 1)
    if tgt and tgt in labels: succ[i].append(labels[tgt])
livein = [set() for _ in range(n)]
changed = True
while changed:
    changed = False
    for i in range(n - 1, -1, -1):
        out = set()
        for s_ in succ[i]:
            if s_ < n: out |= livein[s_]
        new = (out - ins[i][1]) | ins[i][2]
        if new != livein[i]:
            livein[i] = new; changed = True
mx = {}; order = []
for i in range(n):
    ph = ins[i][3]
    if ph not in mx: mx[ph] = (0, 1 << 30); order.append(ph)
    c = len(livein[i])
    mx[ph] = (max(mx[ph][0], c), min(mx[ph][1], c))
print(" ".join("%s:%d/%d" % (p, mx[p][0], mx[p][1]) for p in order))
if len(sys.argv) > 3:
    ph = sys.argv[3]
    idx = [i for i in range(n) if ins[i][3] == ph]
    common = set.intersection(*[livein[i] for i in idx])
    print("always live in", ph, len(common), sorted(common))
    raw = [l.split(";")[0].strip() for l in lines]
    # next use / last def text for each
    flat = []
    phase = "PRO"
    for l in lines:
        mm = re.search(r"; DDP_MARK (\w+)", l)
        if mm: phase = mm.group(1); continue
        if re.match(r"^(\.LBB\w+):", l): continue
        t = l.split(";")[0].strip()
        if not t or t.startswith(".") or t.startswith(";"): continue
        flat.append((phase, t))
    for r in sorted(common):
        nxt = next((j for j in range(idx[-1], n) if r in ins[j][2]), None)
        prv = next((j for j in range(idx[0], -1, -1) if r in ins[j][1]), None)
        print("v%d  def[%s] %s   ||  use[%s] %s" % (r, flat[prv][0] if prv is not None else "-", flat[prv][1][:60] if prv is not None else "-",
              flat[nxt][0] if nxt is not None else "-", flat[nxt][1][:60] if nxt is not None else "-"))
if len(sys.argv) > 4 and sys.argv[4] == "peak":
    ph = sys.argv[3]
    idx = [i for i in range(n) if ins[i][3] == ph]
    pk = max(idx, key=lambda i: len(livein[i]))
    print("peak at instr", pk - idx[0], "of", len(idx), "live", len(livein[pk]))
    for j in range(max(idx[0], pk - 12), min(idx[-1], pk + 6)):
        print("  %3d %s" % (len(livein[j]), flat[j][1][:90]))
    print("live set:", sorted(livein[pk]))
    prof = [len(livein[i]) for i in idx]
    print("profile (every 20th):", prof[::20])
