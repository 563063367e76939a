#!/usr/bin/env python3
"""Kernel table (VGPRs, AGPRs, scratch bytes per lane, waves per SIMD, static LDS) from hipcc's resource remarks:
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -c --cuda-device-only -Rpass-analysis=kernel-resource-usage X.hip -o /dev/null 2>&1 | python3 tools/kernel_resources.py
(profiles/r04_kernel_resources.txt is this over every source of coldrec_amd/csrc)."""
import sys,re,subprocess
name=None;rec={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: name=m.group(1); rec[name]={}
    for k,sh in (('VGPRs','V'),('AGPRs','A'),(r'ScratchSize \[bytes/lane\]','scr'),(r'Occupancy \[waves/SIMD\]','occ'),(r'LDS Size \[bytes/block\]','lds')):
        m=re.search(k+r': (\d+)',l)
        if m and name: rec[name][sh]=int(m.group(1))
for n,r in rec.items():
    d=subprocess.run(['c++filt',n],capture_output=True,text=True).stdout.strip()
    d=re.sub(r'\(anonymous namespace\)::','',d); d=re.sub(r'\(.*','',d); d=d.replace('void ','')
    print('%-58s %s'%(d[:58],' '.join('%s=%d'%kv for kv in r.items())))
