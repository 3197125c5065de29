#!/usr/bin/env python3
"""Timeline inside the one-launch decode step (sel_decode_fused.hip): s_memrealtime stamps of thread 0 of the middle row's workgroup(s),
mean of 20 steps.  Needs the library built with the stamps (on the GPU box, into a scratch copy):
    make -C nsa_vibe_amd/csrc clean && make -C nsa_vibe_amd/csrc TIMELINE=1 -j16
or in the container, beside the product:  make -C nsa_vibe_amd/csrc TIMELINE=1 BUILD=build_tl OUT=../../ab/timeline.so -j8   and run with NSA_HIP_LIB=ab/timeline.so
usage: python tools/decode_timeline2.py [cold] [BxS ...] (tuning switches through the NSA_HIP_* environment; cold: the steps rotate over bench.py's cache sets)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, nsa_vibe_amd as nv
from nsa_vibe_amd import _lib
dev = torch.device("cuda", 0)
L = _lib.lib()
if not hasattr(L, "nsa_debug_read_ts2"):
    sys.exit("libnsa_sel_hip.so was built without TIMELINE=1 (see the docstring)")
L.nsa_debug_read_ts2.argtypes = [ctypes.c_void_p]
names = ["start", "logits + records (thread 0)", "stores drained", "barrier", "ticket / barrier: finisher", "log-sum-exp", "group scores written", "barrier",
         "top-n", "barrier, ranges stored", "gather + merge + O"]
args = sys.argv[1:]
cold = bool(args) and args[0] == "cold"
args = args[1:] if cold else args
shapes = [tuple(int(v) for v in a.split("x")) for a in args] or [(64, 16384), (256, 16384), (64, 65536)]
for B, S in shapes:
    n_sets = min(32, -(-bench.COLD_BYTES_BETWEEN_USES // bench.decode_step_bytes(B, S)) + 1) if cold else 1
    meta, sets = bench.decode_cache_sets(nv, B, S, dev, n_sets)
    O = torch.empty(B, 1, bench.G, bench.H, bench.D, device=dev, dtype=torch.bfloat16)
    rg = torch.empty(B, bench.G, bench.N_SEL, 2, device=dev, dtype=torch.int32)
    for i in range(5):
        q1, Kc, K, V = sets[i % n_sets]
        nv.selection_decode_step(q1, Kc, K, V, meta, bench.N_SEL, S - 1, out=O, ranges_out=rg)
    torch.cuda.synchronize()
    acc = np.zeros(32); n = 20
    for i in range(n):
        ts = (ctypes.c_longlong * 32)()
        q1, Kc, K, V = sets[(5 + i) % n_sets]
        nv.selection_decode_step(q1, Kc, K, V, meta, bench.N_SEL, S - 1, out=O, ranges_out=rg)
        torch.cuda.synchronize()
        L.nsa_debug_read_ts2(ts)
        a = np.array(ts[:], dtype=np.float64)
        acc += (a - a[0]) * 0.01  # 100 MHz -> us
    acc /= n
    print(f"B={B} S={S} {'cold' if cold else 'warm'}  (us since the start stamp of the middle row; split forms: stamps 0-3 are whichever of the row's workgroups wrote last)")
    prev = 0.0
    for i, nm in enumerate(names):
        print(f"  {nm:32s} {acc[i]:7.2f}  (+{acc[i] - prev:5.2f})")
        prev = acc[i]
    del meta, sets, Kc, K, V
    torch.cuda.empty_cache()
