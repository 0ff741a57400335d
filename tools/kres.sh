#!/bin/bash
# kernel resource usage of one csrc file: tools/kres.sh attention_mfma.hip [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c /root/repo/multi-feature-vit_amd/csrc/$f -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); vals={}
    for k in ('VGPRs','AGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','LDS Size \[bytes/block\]','SGPRs'):
        m=re.search(k+r': (\d+)',l)
        if m: vals[k.split(' ')[0]]=m.group(1)
    if 'LDS Size' in l and cur:
        import subprocess
        name=subprocess.run(['c++filt',cur],capture_output=True,text=True).stdout.strip()[:110]
        print(vals, name)
"
