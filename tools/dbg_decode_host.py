import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nsa_vibe_amd as nv
from nsa_vibe_amd import _lib
S, B, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
dev = torch.device("cuda")
m = nv.NSAAttention(768, 12, 2, 64, 64, 32, 16, 64, 16, 512, selector="batched").to(dev).to(torch.bfloat16).eval()
x = torch.randn(B, S, 768, device=dev, dtype=torch.bfloat16)
L = _lib.lib()
orig = L.nsa_layer_decode_step
acc = [0.0, 0]
def timed(*a):
    t0 = time.perf_counter()
    r = orig(*a)
    acc[0] += time.perf_counter() - t0
    acc[1] += 1
    return r
with torch.no_grad():
    kv = m.new_kv(B, S + steps + 8, dev, torch.bfloat16)
    y, kv = m(x, kv, prefill=True)
    xt = torch.randn(B, 1, 768, device=dev, dtype=torch.bfloat16)
    for _ in range(8):
        y, kv = m(xt, kv, prefill=False)
    torch.cuda.synchronize()
    L.nsa_layer_decode_step = timed
    t0 = time.perf_counter()
    for _ in range(steps):
        y, kv = m(xt, kv, prefill=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"per step: python loop {1e6*(t1-t0)/steps:.1f} us (of which C call {1e6*acc[0]/acc[1]:.1f} us), with final sync {1e6*(t2-t0)/steps:.1f} us")
