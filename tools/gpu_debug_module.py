#!/usr/bin/env python3
"""Debug aid: replay the g12 module golden on the GPU and, per decode step, compare the device selection chain with the
oracle evaluated on the module's own Q / K_cmp (is a mismatch vs the reference a near-tie flip or a kernel bug?)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nsa_vibe_amd as nv  # noqa: E402
from nsa_vibe_amd import nsa_attention as na  # noqa: E402
from oracle import nsa_oracle as orc  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "g12_module.npz"))
dim, H, G, dk, dv, l, d, ls, n, w = (int(x) for x in g["cfg"])
m = na.NSAAttention(dim, H, G, dk, dv, l=l, d=d, l_sel=ls, n_sel=n, w=w, selector="sequential")
m.load_state_dict({k[6:].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith("state_")})
with torch.no_grad():
    m.gate.fc2.bias.copy_(torch.tensor([-1000.0, 1000.0, -1000.0]))
m = m.cuda().eval()
x_pre, x_dec = torch.from_numpy(g["x_pre"]).cuda(), torch.from_numpy(g["x_dec"]).cuda()
kv = m.new_kv(x_pre.shape[0], 128, "cuda", torch.float32)
captured = {}
orig = na.selection_decode_step


def spy(Q, Kc, K, V, meta, n_top, t, **kw):
    O, r = orig(Q, Kc, K, V, meta, n_top, t, **kw)
    captured.update(Q=Q.clone(), Kc=Kc.clone(), K=K.clone(), V=V.clone(), meta=meta, t=t, O=O.clone(), r=r.clone())
    return O, r


na.selection_decode_step = spy
with torch.no_grad():
    out, kv = m(x_pre, kv, prefill=True)
    print("prefill err", float(np.abs(out.cpu().numpy() - g["out_pre_seq"]).max()))
    for i in range(x_dec.shape[0]):
        o, kv = m(x_dec[i], kv, prefill=False)
        err = float(np.abs(o.cpu().numpy() - g["out_dec_seq"][i]).max())
        c = captured
        mo = orc.build_block_meta(c["meta"].S_sel * ls if False else kv.meta_seq_len, l, d, ls, n, w)
        Qn, Kcn = c["Q"].cpu().numpy(), c["Kc"].cpu().numpy()
        pc = orc.compute_pcmp_all(Qn, Kcn, 1.0 / np.sqrt(dk))
        _, pg = orc.map_pcmp_to_pslc_and_pgrp(pc, mo)
        pg_dev = nv.selection_scores(c["Q"], c["Kc"], c["meta"]).cpu().numpy()
        r_or = orc.select_topn_ranges(pg[:, 0], mo, n, c["t"])
        same = orc.normalise_ranges(c["r"].cpu().numpy()) == orc.normalise_ranges(r_or)
        srt = np.sort(pg[:, 0], axis=-1)[..., ::-1]
        print(f"step {i:2d} t={c['t']:3d} n_cmp={Kcn.shape[2]:3d} meta_len={kv.meta_seq_len:3d} err={err:.2e} pgrp_dev_err={np.abs(pg_dev - pg).max():.1e} "
              f"ranges==oracle:{same}  dev={orc.normalise_ranges(c['r'].cpu().numpy())[0]}  top scores b0g0={np.array2string(pg[0, 0, 0], precision=4)}")
