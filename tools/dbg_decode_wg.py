#!/usr/bin/env python3
"""diagnostic of the decode workgroup kernel: one case per child process (a GPU fault in one case must not hide the others)
    python tools/dbg_decode_wg.py            -> runs every case in its own subprocess
    python tools/dbg_decode_wg.py <case>     -> one case"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = ["one_chunk", "eight_chunks", "nine_chunks", "sixteen_chunks", "many_chunks", "split_route"]

if len(sys.argv) == 1:
    bad = False
    for c in CASES:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True, timeout=120)
        tail = (r.stdout + r.stderr).strip().splitlines()[-3:]
        print(f"{c:16s} rc={r.returncode} :: " + " | ".join(tail), flush=True)
        bad = bad or r.returncode != 0
    sys.exit(1 if bad else 0)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import nsa_vibe_amd as nv  # noqa: E402
from oracle import nsa_oracle as orc  # noqa: E402

case = sys.argv[1]
rng = np.random.default_rng(0)
B, G, h, D, n, S_kv = 2, 1, 6, 64, 16, 4096
Q = rng.standard_normal((B, 1, G, h, D), dtype=np.float32)
K = rng.standard_normal((B, G, S_kv, D), dtype=np.float32)
V = rng.standard_normal((B, G, S_kv, D), dtype=np.float32)
rg = np.zeros((B, 1, G, n, 2), np.int32)
nchunks = {"one_chunk": 1, "eight_chunks": 8, "nine_chunks": 9, "sixteen_chunks": 16, "many_chunks": 40, "split_route": 16}[case]
for i in range(min(nchunks, n)):
    rg[:, 0, 0, i] = (128 * i, 128 * i + 64)
if nchunks > n:
    rg[:, 0, 0, 0] = (0, 64 * nchunks)
if case == "split_route":
    nv._lib.set_tuning("DECODE_WG", 0)
bf = lambda a: torch.from_numpy(a).cuda().bfloat16()  # noqa: E731
rb = lambda a: torch.from_numpy(a).bfloat16().float().numpy()  # noqa: E731
O = nv.selection_attention_hip(bf(Q), bf(K), bf(V), torch.from_numpy(rg).cuda())
torch.cuda.synchronize()
ref = orc.sel_attention_masked(rb(Q), rb(K), rb(V), rg)
err = np.abs(O.float().cpu().numpy() - ref)
print(f"max err {err.max():.3e}  finite {bool(np.isfinite(O.float().cpu().numpy()).all())}")
