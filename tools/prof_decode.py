#!/usr/bin/env python3
"""Decode-shaped hot path for profiling: B sequences at context S, one token each.
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/prof_decode.py [B] [S] [iters]
Also prints host wall time per step (python + launches) and GPU event time per step."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nsa_vibe_amd as nv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
S = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50

g = torch.Generator(device="cuda")
g.manual_seed(0)
meta = nv.build_block_meta(S, 32, 16, 64, 16, 512)
Q = torch.randn(B, 1, 2, 6, 64, device="cuda", generator=g).bfloat16()
Kc = torch.randn(B, 2, meta.S_cmp, 64, device="cuda", generator=g).bfloat16()
K = torch.randn(B, 2, S, 64, device="cuda", generator=g).bfloat16()
V = torch.randn(B, 2, S, 64, device="cuda", generator=g).bfloat16()
t = S - 1


O_buf = torch.empty(B, 1, 2, 6, 64, device="cuda", dtype=torch.bfloat16)
R_buf = torch.empty(B, 2, 16, 2, device="cuda", dtype=torch.int32)


def step():
    if os.environ.get("NSA_DECODE_SEPARATE"):
        p = nv.selection_scores(Q, Kc, meta, causal_skip=True)
        r = nv.select_topn_ranges(p[:, 0], meta, 16, t)
        return nv.selection_attention_hip(Q, K, V, r.unsqueeze(1))
    return nv.selection_decode_step(Q, Kc, K, V, meta, 16, t, out=O_buf, ranges_out=R_buf)


for _ in range(5):
    step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
a.record()
for _ in range(iters):
    step()
b.record()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"decode B={B} S={S}: host issue {1e6 * (t1 - t0) / iters:.1f} us/step, gpu {1e3 * a.elapsed_time(b) / iters:.1f} us/step, "
      f"wall {1e6 * (t2 - t0) / iters:.1f} us/step")
