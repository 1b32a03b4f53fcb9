"""Builds build_variants/cluster_counts.so: the library with ray counters in k_convex (rays tested, rays walked, DDA
steps) for tools/cluster_counts.py.  Patches a COPY of direct_cluster.hip; the product source has no counters."""
import os, subprocess, sys
sys.path.insert(0, ".")
from direct_amd import build as b
root = os.getcwd()
s = open(root + "/direct_amd/csrc/direct_cluster.hip").read()
def sub(s, old, new, n=1):
    assert s.count(old) == n, (s.count(old), old)
    return s.replace(old, new)
s = sub(s, "namespace {\n", "__device__ unsigned long long g_counts[4];\nnamespace {\n")
s = sub(s, "  for (;;) {\n    if (tMaxX < tMaxY) {\n      if (tMaxX < tMaxZ) { id += ix; tMaxX += tDX; }", "  atomicAdd(&g_counts[1], 1ull);\n  for (;;) {\n    atomicAdd(&g_counts[2], 1ull);\n    if (tMaxX < tMaxY) {\n      if (tMaxX < tMaxZ) { id += ix; tMaxX += tDX; }")
s = sub(s, "      need = ray_needs_walk(D, fl, cx, cy, cz, tgt, full);", "      need = ray_needs_walk(D, fl, cx, cy, cz, tgt, full);\n      atomicAdd(&g_counts[0], 1ull);")
s = sub(s, "    push(j < i ? ray_needs_walk(D, fl, cx, cy, cz, cd[j], full) : 0, j);", "    if (j < i) atomicAdd(&g_counts[3], 1ull);\n    push(j < i ? ray_needs_walk(D, fl, cx, cy, cz, cd[j], full) : 0, j);")
s = sub(s, 'extern "C" {\n', 'extern "C" {\nvoid direct_cluster_debug_counts(unsigned long long* out, int reset) {\n  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_counts), 32);\n  if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_counts), z, 32); }\n}\n')
os.makedirs("/tmp/cc/direct_amd/csrc", exist_ok=True); os.makedirs("/tmp/cc/include", exist_ok=True)
import shutil
for f in os.listdir(root + "/direct_amd/csrc"): shutil.copy(root + "/direct_amd/csrc/" + f, "/tmp/cc/direct_amd/csrc/" + f)
for f in os.listdir(root + "/include"): shutil.copy(root + "/include/" + f, "/tmp/cc/include/" + f)
open("/tmp/cc/direct_amd/csrc/direct_cluster.hip", "w").write(s)
os.makedirs(root + "/build_variants", exist_ok=True)
cmd = [b.hipcc()] + b.FLAGS + ["/tmp/cc/direct_amd/csrc/" + f for f in ("direct_ddp.hip", "direct_cluster.hip", "direct_quad.hip")] + ["-o", root + "/build_variants/cluster_counts.so"]
r = subprocess.run(cmd, capture_output=True, text=True)
print(r.returncode, [l for l in r.stderr.splitlines() if "error" in l][:5])
