"""Scratch accesses inside the sweeps, per kernel instantiation (needs a -DDDP_MARKS -S build, tools/isa_stats.sh).
A spill reload inside a sweep waits with vmcnt(0) and drains the software prefetch."""
import re
import sys
txt = open(sys.argv[1]).read()
for kern in ("_Z9k_iterateIfLi2E", "_Z9k_iterateIfLi3E", "_Z9k_iterateIdLi2E", "_Z9k_iterateIdLi3E", "_Z13k_iterate_dynIfLi2E",
             "_Z13k_iterate_dynIfLi3E", "_Z13k_iterate_dynIfLi4E", "_Z13k_iterate_dynIdLi2E", "_Z13k_iterate_dynIdLi3E"):
    m = re.search(r"^%s\w*:.*?\.Lfunc_end\d+:" % kern, txt, re.S | re.M)
    if not m:
        continue
    cur, cnt = "PRO", {}
    for l in m.group(0).split("\n"):
        mm = re.search(r"; DDP_MARK (\w+)", l)
        if mm:
            cur = mm.group(1)
            continue
        if re.match(r"\s+scratch_", l):
            cnt[cur] = cnt.get(cur, 0) + 1
    print("%-26s in-sweep scratch ops: %s" % (kern, {k: v for k, v in cnt.items() if k[0] in "BF" and k not in ("B_END", "F_END")}))
