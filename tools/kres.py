#!/usr/bin/env python3
"""Print VGPR/AGPR/spill/occupancy/LDS per kernel of a .hip file: tools/kres.py csrc/gemm.hip [filter]"""
import re, subprocess, sys
src = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-c", src, "-o", "/tmp/kres.o",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = {}
for line in out.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
    else:
        k, _, v = t.partition(":")
        cur[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size"):
            if filt in cur["name"]:
                n = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
                n = n.replace("(anonymous namespace)::", "").split("(")[0][-70:]
                print(f"{n:70s} vgpr {cur.get('VGPRs'):>4} agpr {cur.get('AGPRs'):>4} spill {cur.get('VGPRs Spill'):>3} occ {cur.get('Occupancy [waves/SIMD]'):>2} scratch {cur.get('ScratchSize [bytes/lane]')}")
