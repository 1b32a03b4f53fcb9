"""Builds build_variants/iter_times.so: the library with a wall-clock stamp (100 MHz) per outer iteration in every
trajectory's state, for tools/iter_times.py.  Patches COPIES of the sources under /tmp."""
import os, shutil, subprocess, sys
sys.path.insert(0, ".")
from direct_amd import build as b
root = os.getcwd()
os.makedirs("/tmp/v3/direct_amd/csrc", exist_ok=True); os.makedirs("/tmp/v3/include", exist_ok=True)
for f in os.listdir(root + "/direct_amd/csrc"): shutil.copy(root + "/direct_amd/csrc/" + f, "/tmp/v3/direct_amd/csrc/" + f)
for f in os.listdir(root + "/include"): shutil.copy(root + "/include/" + f, "/tmp/v3/include/" + f)
def sub(s, old, new):
    assert s.count(old) == 1, old
    return s.replace(old, new)
s = open("/tmp/v3/direct_amd/csrc/ddp_wave.h").read()
s = sub(s, "  int neg_time, nseg, nc0, npos;", "  int neg_time, nseg, nc0, npos;\n  long long t_it[96];")
s = sub(s, "    fwd_pass(helper);\n    if (helper) return;", "    fwd_pass(helper);\n    if (helper) return;\n    if (st.fwd_passes < 96) st.t_it[st.fwd_passes] = (long long)wall_clock64();")
open("/tmp/v3/direct_amd/csrc/ddp_wave.h", "w").write(s)
h = open("/tmp/v3/direct_amd/csrc/direct_ddp.hip").read()
h = sub(h, "#if defined(DDP_TIMING)\n// debug builds only", """direct_status_t direct_ddp_debug_state(direct_ddp_handle_t h, void* dst, int32_t* stride) {
  HIP_TRY(hipMemcpy(dst, h->st, (size_t)h->B * sizeof(TrajState), hipMemcpyDeviceToHost));
  *stride = (int32_t)sizeof(TrajState);
  return DIRECT_OK;
}
#if defined(DDP_TIMING)
// debug builds only""")
open("/tmp/v3/direct_amd/csrc/direct_ddp.hip", "w").write(h)
os.makedirs(root + "/build_variants", exist_ok=True)
cmd = [b.hipcc()] + b.FLAGS + ["/tmp/v3/direct_amd/csrc/" + f for f in ("direct_ddp.hip", "direct_cluster.hip", "direct_quad.hip")] + ["-o", root + "/build_variants/iter_times.so"]
r = subprocess.run(cmd, capture_output=True, text=True)
print(r.returncode, [l for l in r.stderr.splitlines() if "error" in l][:5])
