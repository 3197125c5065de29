import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nsa_vibe_amd as nv
L = ctypes.CDLL(os.path.join(ROOT, "nsa_vibe_amd", "libnsa_sel_hip.so"))
g = torch.Generator(device="cuda"); g.manual_seed(0)
mk = lambda *s: torch.randn(*s, device="cuda", generator=g).bfloat16()
S = 65536; S_cmp = (S - 32) // 16 + 1
Q, Kc, Vc = mk(1, S, 2, 6, 64), mk(1, 2, S_cmp, 64), mk(1, 2, S_cmp, 64)
buf = np.zeros(8, dtype=np.uint64)
nv.batched_causal_attention_compressed(Q, Kc, Vc, 32, 16); torch.cuda.synchronize()
L.nsa_dbg_band_read(buf.ctypes.data_as(ctypes.c_void_p), 1)
nv.batched_causal_attention_compressed(Q, Kc, Vc, 32, 16); torch.cuda.synchronize()
L.nsa_dbg_band_read(buf.ctypes.data_as(ctypes.c_void_p), 1)
n = int(buf[7])
names = ["loop top (prev tail)", "wait DMA (vmcnt 0)", "LDS frag reads + lgkmcnt", "issue next DMA", "S MFMAs + x/max", "slow path? + exp/cvt + PV MFMAs"]
tot = sum(int(buf[i]) for i in range(6))
print(f"tiles sampled {n}; cycles per tile (s_memtime units) total {tot / n:.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:36s} {int(buf[i]) / n:8.1f}  ({100.0 * int(buf[i]) / tot:4.1f} %)")
