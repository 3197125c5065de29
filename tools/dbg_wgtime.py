import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nsa_vibe_amd as nv
from nsa_vibe_amd import _lib
B, S = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator(device="cuda"); g.manual_seed(0)
meta = nv.build_block_meta(S, 32, 16, 64, 16, 512)
mk = lambda *s: torch.randn(*s, device="cuda", generator=g).bfloat16()
Q, K, V, dO = mk(B, S, 2, 6, 64), mk(B, 2, S, 64), mk(B, 2, S, 64), mk(B, S, 2, 6, 64)
rg = nv.select_topn_ranges_batched(torch.rand(B, S, 2, meta.S_sel, device="cuda", generator=g), meta, 16, S)
q, k, v = (x.clone().requires_grad_(True) for x in (Q, K, V))
for _ in range(3):
    q.grad = k.grad = v.grad = None
    nv.selection_attention_hip(q, k, v, rg).backward(dO)
    torch.cuda.synchronize()
L = ctypes.CDLL(os.path.join(ROOT, "nsa_vibe_amd", "libnsa_sel_hip.so"))
buf = np.zeros(4 * 65536, dtype=np.uint64)
print("rc", L.nsa_dbg_read(buf.ctypes.data_as(ctypes.c_void_p)))
nkb, nbg, ns = (S + 63) // 64, 2 * B, min(16, max(1, (S + 511) // 512))
n = min(65536, nkb * nbg * ns)
d = buf.reshape(-1, 4)[:n].astype(np.int64)
t0 = d[:, 0].min()
st, en, hits = (d[:, 0] - t0) / 100.0, (d[:, 1] - t0) / 100.0, d[:, 2]   # us (100 MHz clock)
dur = en - st
print(f"WGs {n}  kernel span {en.max():.1f} us  sum(dur) {dur.sum()/1e3:.1f} ms  mean dur {dur.mean():.1f} us  max dur {dur.max():.1f}")
print("concurrency avg", dur.sum() / en.max())
for lo, hi in ((0, 1), (1, 17), (17, 65), (65, 129), (129, 257), (257, 513)):
    m = (hits >= lo) & (hits < hi)
    if m.any():
        print(f"hits [{lo},{hi}): n={m.sum():5d} mean dur {dur[m].mean():8.1f} us  per-hit {dur[m].sum()/max(1,hits[m].sum()):.3f} us  start range {st[m].min():.0f}..{st[m].max():.0f}")
# timeline: concurrency in 20 bins
edges = np.linspace(0, en.max(), 21)
for a, b in zip(edges[:-1], edges[1:]):
    c = (np.minimum(en, b) - np.maximum(st, a)).clip(0).sum() / (b - a)
    print(f"  {a:7.0f}-{b:7.0f} us: {c:6.1f} WGs resident; started {((st>=a)&(st<b)).sum()}")
print("xcc counts", np.bincount(d[:, 3] & 15))
