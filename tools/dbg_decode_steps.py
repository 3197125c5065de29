import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nsa_vibe_amd as nv
S, B, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
dev = torch.device("cuda")
m = nv.NSAAttention(768, 12, 2, 64, 64, 32, 16, 64, 16, 512, selector="batched").to(dev).to(torch.bfloat16).eval()
x = torch.randn(B, S, 768, device=dev, dtype=torch.bfloat16)
with torch.no_grad():
    kv = m.new_kv(B, S + steps + 8, dev, torch.bfloat16)
    y, kv = m(x, kv, prefill=True)
    xt = torch.randn(B, 1, 768, device=dev, dtype=torch.bfloat16)
    ts = []
    for i in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y, kv = m(xt, kv, prefill=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
import numpy as np
a = np.array(ts) * 1e6
print("host-issue us: median %.1f max %.1f (step %d)   total us: median %.1f max %.1f (step %d)" % (np.median(a[:, 0]), a[:, 0].max(), a[:, 0].argmax(), np.median(a[:, 1]), a[:, 1].max(), a[:, 1].argmax()))
print("slow steps:", [(i, round(v)) for i, v in enumerate(a[:, 1]) if v > 3 * np.median(a[:, 1])][:20])
