#!/bin/bash
# tools/kres.sh FILE.hip [extra flags]: per kernel VGPRs / AGPRs / spills / scratch / occupancy of one compilation unit
# (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel
cd "$(dirname "$0")/../aspire_amd/csrc"
f=$1; shift
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage "$@" -c "$f" -o /tmp/kres.o 2>&1 |
python3 -c '
import re,sys,subprocess
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r"Function Name: (\S+)",line)
    if m:
        cur={"name":m.group(1)}; rows.append(cur); continue
    if cur is None: continue
    for key,pat in (("vgpr",r" VGPRs: (\d+)"),("agpr",r"AGPRs: (\d+)"),("spill",r"VGPR Spill: (\d+)"),("scratch",r"ScratchSize \[bytes/lane\]: (\d+)"),("occ",r"Occupancy \[waves/SIMD\]: (\d+)"),("lds",r"LDS Size \[bytes/block\]: (\d+)"),("sgpr",r" SGPRs: (\d+)")):
        m=re.search(pat,line)
        if m: cur[key]=int(m.group(1))
names=subprocess.run(["/usr/bin/c++filt"]+[r["name"] for r in rows],capture_output=True,text=True).stdout.splitlines()
for r,nm in zip(rows,names):
    print("%-110s v=%3d a=%3d spill=%3d scratch=%4d occ=%d sgpr=%d"%(nm[:110],r.get("vgpr",-1),r.get("agpr",-1),r.get("spill",-1),r.get("scratch",-1),r.get("occ",-1),r.get("sgpr",-1)))
'
