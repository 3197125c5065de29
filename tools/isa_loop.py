#!/usr/bin/env python3
"""Per-basic-block instruction mix of the loop blocks of one kernel.

    python tools/isa_loop.py <file.hip> <mangled-name-substring> [min_block_len]
Compiles with the Makefile's flags (-S, device only) and prints VALU/SALU/DS/VMEM/MFMA counts per loop block.
"""
import collections
import re
import subprocess
import sys

src, key = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 8
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-slp-vectorize"]
subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-S", "--cuda-device-only", src, "-o", "/tmp/isa_loop.s"], check=True,
               stderr=subprocess.DEVNULL)
s = open("/tmp/isa_loop.s").read().split("\n")
start = next(i for i, l in enumerate(s) if re.match(r"^_Z\S+:", l) and key in l)
end = next(i for i in range(start, len(s)) if ".amdhsa_kernel" in s[i])
body = s[start:end]
labels = [i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)] + [len(body)]


def kind(b):
    if b.startswith("v_mfma"):
        return "mfma"
    if b.startswith("v_"):
        return "valu"
    if b.startswith("s_"):
        return "salu"
    if b.startswith("ds_"):
        return "ds"
    if b.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


for a, b in zip(labels[:-1], labels[1:]):
    blk = [x.strip().split()[0] for x in body[a + 1:b] if x.strip() and not x.strip().startswith((";", "."))]
    c = collections.Counter(kind(x) for x in blk)
    hdr = body[a]
    if ("Loop" in hdr or "Depth" in hdr) and len(blk) >= minlen:
        print(hdr[:58].ljust(58), len(blk), dict(c))
        if c["valu"] > 12:
            print("      ", collections.Counter(x for x in blk if kind(x) == "valu").most_common(12))
for l in s[end:end + 80]:
    if re.search(r"\.(num_vgpr|num_agpr|scratch|private_seg)", l) and key in l:
        print(l.strip()[-40:])
