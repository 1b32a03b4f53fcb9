"""Builds build_variants/chain.so: the library with per-trajectory cycle accounting (backward / forward cycles, rounds
run by the owner, rounds fetched from helpers) for tools/chain_stats.py.  Patches COPIES of the sources under /tmp."""
import os, shutil, subprocess, sys
sys.path.insert(0, ".")
from direct_amd import build as b
root = os.getcwd()
os.makedirs("/tmp/v2/direct_amd/csrc", exist_ok=True); os.makedirs("/tmp/v2/include", exist_ok=True)
for f in os.listdir(root + "/direct_amd/csrc"): shutil.copy(root + "/direct_amd/csrc/" + f, "/tmp/v2/direct_amd/csrc/" + f)
for f in os.listdir(root + "/include"): shutil.copy(root + "/include/" + f, "/tmp/v2/include/" + f)
def sub(s, old, new):
    assert s.count(old) == 1, old
    return s.replace(old, new)
s = open("/tmp/v2/direct_amd/csrc/ddp_wave.h").read()
s = sub(s, "  int neg_time, nseg, nc0, npos;", "  int neg_time, nseg, nc0, npos;\n  long long cyc_bwd, cyc_fwd; int n_rounds, n_fetched;")
s = sub(s, "    if (!helper) {\n      while (true) {  // DDP:297-310", "    long long t0_ = __builtin_readcyclecounter();\n    if (!helper) {\n      while (true) {  // DDP:297-310")
s = sub(s, "    fwd_pass(helper);\n    if (helper) return;", "    long long t1_ = __builtin_readcyclecounter();\n    fwd_pass(helper);\n    if (helper) return;\n    long long t2_ = __builtin_readcyclecounter();\n    st.cyc_bwd += t1_ - t0_; st.cyc_fwd += t2_ - t1_;")
s = sub(s, "      if (mine >= 0) {\n        if (nt == 2)", "      if (mine >= 0) {\n        if (!helper) st.n_rounds++;\n        if (nt == 2)")
s = sub(s, "          if (!fetch_results(hs, r_eval, pair ? 2 * r_eval - 1 : r_eval, pair ? 2 : 1, tag, o)) {", "          st.n_fetched++;\n          if (!fetch_results(hs, r_eval, pair ? 2 * r_eval - 1 : r_eval, pair ? 2 : 1, tag, o)) {")
s = sub(s, "    st.fwd_passes = 0;\n  }", "    st.fwd_passes = 0;\n    st.cyc_bwd = 0; st.cyc_fwd = 0; st.n_rounds = 0; st.n_fetched = 0;\n  }")
open("/tmp/v2/direct_amd/csrc/ddp_wave.h", "w").write(s)
h = open("/tmp/v2/direct_amd/csrc/direct_ddp.hip").read()
h = sub(h, "#if defined(DDP_TIMING)\n// debug builds only", """direct_status_t direct_ddp_debug_state(direct_ddp_handle_t h, void* dst, int32_t* stride) {
  HIP_TRY(hipMemcpy(dst, h->st, (size_t)h->B * sizeof(TrajState), hipMemcpyDeviceToHost));
  *stride = (int32_t)sizeof(TrajState);
  return DIRECT_OK;
}
#if defined(DDP_TIMING)
// debug builds only""")
open("/tmp/v2/direct_amd/csrc/direct_ddp.hip", "w").write(h)
os.makedirs(root + "/build_variants", exist_ok=True)
cmd = [b.hipcc()] + b.FLAGS + ["/tmp/v2/direct_amd/csrc/" + f for f in ("direct_ddp.hip", "direct_cluster.hip", "direct_quad.hip")] + ["-o", root + "/build_variants/chain.so"]
r = subprocess.run(cmd, capture_output=True, text=True)
print(r.returncode, [l for l in r.stderr.splitlines() if "error" in l][:5])
