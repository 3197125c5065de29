"""Experiment: replay ONE captured decode step (fixed position) as a HIP graph to see what launch-overhead-free decode would cost."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nsa_vibe_amd as nv
S, B = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(0)
dev = torch.device("cuda")
m = nv.NSAAttention(768, 12, 2, 64, 64, 32, 16, 64, 16, 512, selector="batched").to(dev).to(torch.bfloat16).eval()
x = torch.randn(B, S, 768, device=dev, dtype=torch.bfloat16)
with torch.no_grad():
    kv = m.new_kv(B, S + 64, dev, torch.bfloat16)
    m(x, kv, prefill=True)
    xt = torch.randn(B, 1, 768, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        m(xt, kv, prefill=False)
    t_fixed = kv.t
    def step():
        kv.t = t_fixed
        return m(xt, kv, prefill=False)[0]
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): step()
    torch.cuda.synchronize()
    print(f"eager (same position): {(time.perf_counter()-t0)/50*1e6:.1f} us/step")
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            y = step()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    print(f"graph replay: {(time.perf_counter()-t0)/50*1e6:.1f} us/step")
