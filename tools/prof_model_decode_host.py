"""host time of a whole-model decode call (TinyLM.decode: one native call per token) against its time to completion"""
import os, sys, time, cProfile, pstats, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsa_vibe_amd.llama_block_nsa import TinyLM
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda")
torch.manual_seed(0)
lm = TinyLM(50257, 768, 12, 12, 2, 64, 64, 32, 16, 64, 16, 512, selector="batched").to(dev).to(torch.bfloat16).eval()
tok = torch.randint(0, 50257, (B, S), device=dev)
with torch.no_grad():
    caches = lm.new_caches(B, S + 1200, dev, torch.bfloat16)
    nxt = lm.prefill(tok, caches).argmax(-1)
    for _ in range(20): nxt = lm.decode(nxt, caches, return_next=True)[1]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): nxt = lm.decode(nxt, caches, return_next=True)[1]
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"S={S} B={B}: host enqueue {1e6*(t1-t0)/300:.0f} us/token, to completion {1e6*(t2-t0)/300:.0f} us/token")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300): nxt = lm.decode(nxt, caches, return_next=True)[1]
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
