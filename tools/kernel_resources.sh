#!/bin/bash
# register / scratch / LDS use of every kernel of one source: tools/kernel_resources.sh conv2d_x3 ["-D..."]
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --cuda-device-only $2 -c practicaldeepstereo_nips2018_amd/csrc/$1.hip -o /tmp/kr_$1.co \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur={}
for line in sys.stdin:
    m=re.search(r'remark: (.*?) \[-Rpass', line)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith('Function Name:'):
        if cur: print(cur)
        cur={'fn':t.split(':',1)[1].strip()[:70]}
    else:
        k,_,v=t.partition(':'); k=k.strip()
        if k in ('VGPRs','AGPRs','TotalSGPRs','ScratchSize [bytes/lane]','VGPR Spill','SGPR Spill','Occupancy [waves/SIMD]','LDS Size [bytes/block]'): cur[k.split(' ')[0]]=v.strip()
if cur: print(cur)
"
