import os, sys, cProfile, pstats, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import nsa_vibe_amd as nv
dev = torch.device("cuda")
m = nv.NSAAttention(768, 12, 2, 64, 64, 32, 16, 64, 16, 512, selector="batched").to(dev).to(torch.bfloat16).eval()
S, B = 4096, 1
x = torch.randn(B, S, 768, device=dev, dtype=torch.bfloat16)
with torch.no_grad():
    kv = m.new_kv(B, S + 3000, dev, torch.bfloat16)
    m(x, kv, prefill=True)
    xt = torch.randn(B, 1, 768, device=dev, dtype=torch.bfloat16)
    for _ in range(50): m(xt, kv, prefill=False)
    torch.cuda.synchronize()
    # host time per call with the GPU far behind?  measure pure host: time 500 calls without sync
    t0 = time.perf_counter()
    for _ in range(500): m(xt, kv, prefill=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host {1e6*(t1-t0)/500:.1f} us/call enqueue, {1e6*(t2-t0)/500:.1f} us/call to completion")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(1000): m(xt, kv, prefill=False)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
