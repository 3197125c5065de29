#!/usr/bin/env python3
"""Compact per-kernel resource table (VGPR/SGPR/scratch/occupancy) for a .hip file."""
import re, subprocess, sys
src = sys.argv[1]
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
                      "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-slp-vectorize", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = subprocess.run(["c++filt", t.split(":")[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur).replace("nsa::", "")
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
for k, r in rows.items():
    print(f"{k[:70]:70s} vgpr={r.get('VGPRs')} agpr={r.get('AGPRs')} sgpr={r.get('TotalSGPRs')} scratch={r.get('ScratchSize [bytes/lane]')} occ={r.get('Occupancy [waves/SIMD]')} lds={r.get('LDS Size [bytes/block]')}")
