#!/bin/bash
# kernel resource usage of a HIP source, one line per kernel:  tools/kres.sh file.hip [extra hipcc flags]
f=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I anemoi_core_amd/csrc -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m: cur={"name":m.group(1)}; continue
    m=re.search(r"remark: [^ ]+\s+(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)",l)
    if m and cur is not None:
        cur[m.group(1)]=m.group(2)
        if m.group(1).startswith("LDS"):
            print("%-70s vgpr %s agpr %s sgpr %s scratch %s spill %s occ %s"%(cur["name"][:70],cur.get("VGPRs"),cur.get("AGPRs"),cur.get("SGPRs"),cur.get("ScratchSize [bytes/lane]"),cur.get("VGPRs Spill"),cur.get("Occupancy [waves/SIMD]")))
'
