#!/usr/bin/env python3
"""How much K/V would two adjacent query rows (t, t+1) of one (b,g) share if one wave processed both?  Uses the bench inputs."""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
import nsa_vibe_amd as nv  # noqa: E402

for S, B in ((4096, 2), (16384, 1), (65536, 1)):
    meta, Q, Kc, K, V = bench.make_inputs(nv, B, S, torch.device("cuda"), 1234)
    p = nv.selection_scores(Q, Kc, meta, 0.125, causal_skip=True)
    rg = nv.select_topn_ranges_batched(p, meta, 16, S, True, 2)  # [B,S,G,W,2]
    r = rg[0, :, 0].cpu().long()  # [S,W,2]
    n_blk = (S + 63) // 64
    cover = torch.zeros(S, n_blk + 1, dtype=torch.int32)
    for i in range(r.shape[1]):
        s, e = r[:, i, 0], r[:, i, 1]
        live = e > s
        idx = torch.arange(S)[live]
        cover[idx, (s[live] // 64)] += 1
        cover[idx, ((e[live] + 63) // 64)] -= 1
    blocks = (cover.cumsum(1)[:, :n_blk] > 0)  # selected 64-token blocks per row (partial last block counted whole)
    a, b = blocks[0::2], blocks[1::2]
    both = (a & b).sum().item()
    union = (a | b).sum().item()
    total = a.sum().item() + b.sum().item()
    print(f"S={S}: blocks/row {total / S:.2f}; pair union/sum = {union / total:.3f} (shared {2 * both / total:.3f})")
    q4 = blocks[: S // 4 * 4].view(S // 4, 4, n_blk)
    print(f"        4-row union/sum = {q4.any(1).sum().item() / q4.sum().item():.3f}")
