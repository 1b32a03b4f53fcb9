"""LDS bank conflicts of the hot kernel per phase and source line, from the instrumented emulator (no GPU needed).
usage: python tools/lds_trace/run.py [free|corridor] [N] [iters] [f32|f64]   (build first: tools/lds_trace/build.sh)"""
import collections
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from direct_amd import abi, problems  # noqa: E402
from tests.emu import emuapi  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "free"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
st = sys.argv[4] if len(sys.argv) > 4 else "f32"
so = os.path.join(ROOT, "tools/lds_trace/libddp_emu_trace.so")
L = C.CDLL(so)
L.emu_begin.restype = C.c_void_p
L.emu_begin.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
for n in ("emu_backward", "emu_forward", "emu_end"):
    getattr(L, n).argtypes = [C.c_void_p]
    getattr(L, n).restype = None
L.emu_iterate.argtypes = [C.c_void_p, C.c_int]
L.emu_get_field.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
L.emu_set_field.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
L.emu_finish.argtypes = [C.c_void_p, C.c_void_p]
L.ddp_emu_report.argtypes = [C.c_char_p]
emuapi._LIB = L

batch = problems.make_batch(kind, 1, N, seed=1000)
p0 = abi.phase0_params()
dt = np.float32 if st == "f32" else np.float64
g0 = emuapi.solve_batch(p0, batch, dt, compute64=True)
b1 = batch.with_init(None, T0=np.where((g0.rtn == 2)[:, None], g0.T, batch.T0), infeas_in=g0.infeas_out, init_poly=g0.poly)
p1 = abi.phase1_params(iter_max=iters, fixed_iters=1)
s = emuapi.EmuSolver(p1, b1, dt, compute64=True)
L.ddp_emu_trace(1)
s.iterate(iters)
L.ddp_emu_trace(0)
out = "/tmp/lds_trace.txt"
L.ddp_emu_report(out.encode())
res = s.finish()
print("phase-0 rtn", g0.rtn, "infeasible mode in phase 1:", bool(g0.infeas_out[0]), "| traced", iters, "iterations, fwd passes", res.fwd_passes)

L.ddp_emu_layout.argtypes = [C.c_char_p]
L.ddp_emu_layout(b"/tmp/lds_layout.txt")
layout = sorted((int(l.split()[1]), l.split()[0]) for l in open("/tmp/lds_layout.txt"))
# the backward / forward halves of the LDS union overlap: resolve an offset within the phase's half
FWD = {"KUr", "ft"}
def member(off, phase):
    best = None
    for o, n in layout:
        if o <= off and n != "end" and ((n in FWD) == phase.startswith("F") or o < dict((n2, o2) for o2, n2 in layout)["We"]):
            if best is None or o >= best[0]:
                best = (o, n)
    return "%s+%d" % (best[1], off - best[0]) if best else "?"
rows = [l.split() for l in open(out)]
addrs = sorted({r[1] for r in rows})
# addr2line -i prints the inlining chain: keep the first frame inside ddp_wave.h
lines, cur = {}, []
text = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-symbolizer", "--obj=" + so, "--output-style=GNU", "-a", "-i", "--no-demangle", "-s"] + ["0x" + a for a in addrs], capture_output=True, text=True).stdout.split("\n")
key = None
for t in text:
    if t.startswith("0x"):
        key = "%x" % int(t, 16)
        lines[key] = []
    elif t and key:
        lines[key].append(os.path.basename(t.split(" ")[0]))
def where(a):
    fr = [x for x in lines.get(a, []) if x.startswith("ddp_wave.h") and not x.endswith(":0")]
    return ",".join(x.split(":")[1] for x in fr[:3]) if fr else "?"
ph = collections.OrderedDict()
for r in rows:
    p, a, rw, size, n, cyc, conf, lanes = r[0], r[1], r[2], int(r[3]), int(r[4]), int(r[5]), int(r[6]), int(r[7])
    ph.setdefault(p, []).append((conf, cyc, n, rw, size, where(a) + " " + member(int(r[8]), p) + ".." + member(int(r[9]), p), lanes))
order = ["B_L", "B_T2", "B_R1", "B_S", "B_S2", "B_H", "B_C", "B_G", "B_R2", "F_L", "F_D", "F_T", "F_R"]
tot_c = tot_k = 0
print("%-6s %8s %8s %8s %7s" % ("phase", "instr", "cycles", "conflict", "share"))
for p in order + [q for q in ph if q not in order]:
    if p not in ph:
        continue
    n = sum(x[2] for x in ph[p]); cyc = sum(x[1] for x in ph[p]); conf = sum(x[0] for x in ph[p])
    if p in order:
        tot_c += cyc; tot_k += conf
    print("%-6s %8d %8d %8d %6.1f%%" % (p, n, cyc, conf, 100.0 * conf / max(cyc, 1)))
print("sweeps: conflict cycles / LDS-array cycles = %d / %d = %.1f %%" % (tot_k, tot_c, 100.0 * tot_k / max(tot_c, 1)))
print("\ntop conflict sites (phase, R/W bytes, ddp_wave.h line(s), instructions, cycles, conflict cycles, lanes/instr):")
allr = [(x[0], p) + x[1:] for p in ph for x in ph[p] if p in order]
for conf, p, cyc, n, rw, size, w, lanes in sorted(allr, reverse=True)[:int(os.environ.get("LDS_TOP", "40"))]:
    if conf == 0:
        break
    print("  %-5s %s%-3d line %-44s n=%-6d cyc=%-7d conf=%-7d lanes=%.0f" % (p, rw, size, w, n, cyc, conf, lanes / max(n, 1)))
