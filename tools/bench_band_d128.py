"""Band kernel timing at head size 128 (bf16, G=2, h=6)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nsa_vibe_amd as nv
g = torch.Generator(device="cuda"); g.manual_seed(0)
mk = lambda *s: torch.randn(*s, device="cuda", generator=g).bfloat16()
for B, S in ((8, 4096), (1, 65536)):
    G, h, D = 2, 6, 128
    Q, K, V = mk(B, S, G, h, D), mk(B, G, S, D), mk(B, G, S, D)
    Sc = (S - 32) // 16 + 1
    Kc, Vc = mk(B, G, Sc, D), mk(B, G, Sc, D)
    t = torch.arange(S)
    kw = int(torch.clamp(t + 1, max=512).sum()); kc = int(torch.where(t + 1 < 32, 0, (t + 1 - 32) // 16 + 1).sum())
    for name, fn, keys in (("win", lambda: nv.sliding_window_attention(Q, K, V, 512), kw), ("cmp", lambda: nv.batched_causal_attention_compressed(Q, Kc, Vc, 32, 16), kc)):
        for _ in range(3): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        print(f"D=128 {name} S={S} B={B}: {ms:.3f} ms  {4.0*B*G*h*D*keys/ms/1e9:.0f} TFLOP/s")
