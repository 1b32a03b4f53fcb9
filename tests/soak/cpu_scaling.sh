mkdir -p gpurun_out
python - > gpurun_out/cpu.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, ".")
import bench
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "usable", bench.usable_cpus())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
os.system("nproc; lscpu | head -20")
import time, numpy as np
from direct_amd import abi, problems
from oracle import refapi
b=problems.make_batch("free",1024,100,seed=1000)
p0=abi.phase0_params()
g0,_=refapi.solve_batch(p0,b,n_threads=bench.usable_cpus())
b1=b.with_init(None,T0=np.where((g0.rtn==2)[:,None],g0.T,b.T0),infeas_in=g0.infeas_out,init_poly=g0.poly)
pf=abi.phase1_params(iter_max=20,fixed_iters=1)
for nt in (1,8,32,64,128,256):
    sub=b1.select(np.arange(min(1024,max(8,4*nt))))
    t=time.time(); g,_=refapi.solve_batch(pf,sub,n_threads=nt); dt=time.time()-t
    print(nt,"threads: %d problems %.2fs  %.0f iter/s"%(sub.batch,dt, g.fwd_passes.sum()/dt), flush=True)
PY
