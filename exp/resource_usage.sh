#!/bin/bash
# Registers, scratch, LDS and occupancy of every kernel as the compiler reports them: bash exp/resource_usage.sh [file.hip] (runs here, no GPU needed)
SRC=${1:-/root/repo/dynamicslamtool_amd/csrc/mor_kernels.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -c -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-value $EXTRA -Rpass-analysis=kernel-resource-usage -x hip "$SRC" -o /tmp/ru_k.o 2> /tmp/ru.txt
python3 - <<'P'
import re
cur=None; res={}
for l in open('/tmp/ru.txt'):
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=re.sub(r'^_Z\d+','',m.group(1)); cur=re.sub(r'(ILi\d+EEv)?6MorDev.*','',cur); res[cur]={}
    for key,pat in (('vgpr',r' VGPRs: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('lds',r'LDS Size \[bytes/block\]: (\d+)'),('sgpr',r'TotalSGPRs: (\d+)')):
        m=re.search(pat,l)
        if m and cur: res[cur][key]=int(m.group(1))
for k,v in res.items(): print("%-18s vgpr %3d scratch %3d occ %d lds %6d sgpr %3d"%(k,v.get('vgpr',0),v.get('scratch',0),v.get('occ',0),v.get('lds',0),v.get('sgpr',0)))
P
