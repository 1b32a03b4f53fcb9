#!/bin/bash
# usage: asm.sh out.s [extra flags]; prints phase counts, private segment sizes and in-loop scratch ops
out=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDDP_WAVES_F32=3 -DDDP_WAVES_F64=3 -mllvm -amdgpu-load-store-vectorizer=0 -Xclang -target-feature -Xclang -load-store-opt -mllvm -amdgpu-sched-strategy=iterative-ilp -DDDP_MARKS -S --cuda-device-only -I/root/repo/include -I/root/repo/direct_amd/csrc "$@" /root/repo/direct_amd/csrc/direct_ddp.hip -o $out 2>&1 | grep -E "error" | head
python /root/repo/tools/phase_count.py $out
python - $out <<'PY'
import re,sys
txt=open(sys.argv[1]).read()
for m in re.finditer(r"\.name:\s+(_Z9k_iterate\w+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+)",txt,re.S):
    print(m.group(1)[:22],"scratch",m.group(2),"vgpr",m.group(3))
for kern in ("_Z9k_iterateIfLi2E", "_Z9k_iterateIfLi3E", "_Z9k_iterateIdLi2E", "_Z13k_iterate_dynIfLi2E", "_Z13k_iterate_dynIdLi2E", "_Z13k_iterate_dynIfLi3E"):
    m=re.search(r"^%s\w*:.*?\.Lfunc_end\d+:" % kern,txt,re.S|re.M)
    if not m: continue
    cur='PRO'; cnt={}
    for l in m.group(0).split('\n'):
        mm=re.search(r"; DDP_MARK (\w+)",l)
        if mm: cur=mm.group(1); continue
        if re.match(r"\s+scratch_",l): cnt[cur]=cnt.get(cur,0)+1
    print(kern, "scratch ops by phase:",cnt)
PY
