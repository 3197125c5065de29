"""host time of one selection_decode_step call against its GPU time (B, S from argv)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, nsa_vibe_amd as nv
B, S = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
d = bench.decode_bench(nv, B, S, 30, dev)
print(f"B={B} S={S}: cold {d['ms_per_step_cold']*1e3:.1f} us, warm {d['ms_per_step_warm']*1e3:.1f} us per step (bench.decode_bench)")
# pure host: enqueue 300 steps on one cache set without waiting
torch.manual_seed(0)
G, h, D, n = 2, 6, 64, 16
meta = nv.build_block_meta(S, 32, 16, 64, n, 512)
Q = torch.randn(B, 1, G, h, D, device=dev).bfloat16()
S_cmp = (S - 32) // 16 + 1
Kc = torch.randn(B, G, S_cmp, D, device=dev).bfloat16()
K = torch.randn(B, G, S, D, device=dev).bfloat16(); V = torch.randn(B, G, S, D, device=dev).bfloat16()
for _ in range(20): nv.selection_decode_step(Q, Kc, K, V, meta, n, S - 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): nv.selection_decode_step(Q, Kc, K, V, meta, n, S - 1)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"   host enqueue {1e6*(t1-t0)/300:.1f} us/call, to completion {1e6*(t2-t0)/300:.1f} us/call")
