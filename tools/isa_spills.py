import re,sys
txt=open(sys.argv[1]).read()
for kn in ("_Z9k_iterateIfLi2E","_Z9k_iterateIdLi2E","_Z9k_iterateIfLi3E","_Z13k_iterate_dynIfLi2E","_Z13k_iterate_dynIdLi2E"):
    m=re.search(r"^%s\w*:.*?\.Lfunc_end\d+:"%kn,txt,re.S|re.M)
    if not m: continue
    cur='PRO'; cnt={}
    for l in m.group(0).split('\n'):
        mm=re.search(r"; DDP_MARK (\w+)",l)
        if mm: cur=mm.group(1); continue
        if re.match(r"\s+scratch_",l): cnt[cur]=cnt.get(cur,0)+1
    print(kn,"scratch ops by phase:",cnt)
