#!/usr/bin/env python3
"""Timeline inside the fused decode launch (one workgroup = one (b,g) row): s_memrealtime stamps at the phase boundaries, taken by
thread 0 of the middle workgroup and averaged over 20 steps.  Needs the library built with the stamps:
    make -C nsa_vibe_amd/csrc clean && make -C nsa_vibe_amd/csrc TIMELINE=1 -j8
(rebuild without TIMELINE afterwards: the stamps cost about 0.1 us each)."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, nsa_vibe_amd as nv
from nsa_vibe_amd import _lib
dev = torch.device("cuda", 0)
L = _lib.lib()
if not hasattr(L, "nsa_debug_read_ts"):
    sys.exit("libnsa_sel_hip.so was built without TIMELINE=1 (see the docstring)")
L.nsa_debug_read_ts.argtypes = [ctypes.c_void_p]
names = {0:"start",1:"ph1 done (w0)",2:"sync1",3:"stats+sync",4:"taps done (w0)",5:"sync",6:"top-n done",7:"sync -> attend",20:"  keys built",21:"  forced done",22:"  sorted",23:"  k_eff counted",24:"  radix done",25:"  picks done",10:"q loaded issue",11:"chunk table",12:"S mfma done",13:"softmax done",14:"V landed",15:"PV done",16:"partials synced",17:"end"}
for B, S in ((64, 16384), (64, 65536)):
    meta, Q, Kc, K, V = bench.make_inputs(nv, B, S, dev, 7)
    q1 = Q[:, -1:].contiguous(); del Q
    O = torch.empty(B, 1, bench.G, bench.H, bench.D, device=dev, dtype=torch.bfloat16)
    rg = torch.empty(B, bench.G, bench.N_SEL, 2, device=dev, dtype=torch.int32)
    for _ in range(5):
        nv.selection_decode_step(q1, Kc, K, V, meta, bench.N_SEL, S - 1, out=O, ranges_out=rg)
    torch.cuda.synchronize()
    acc = np.zeros(64)
    n = 20
    for _ in range(n):
        nv.selection_decode_step(q1, Kc, K, V, meta, bench.N_SEL, S - 1, out=O, ranges_out=rg)
        torch.cuda.synchronize()
        ts = (ctypes.c_longlong * 64)()
        L.nsa_debug_read_ts(ts)
        a = np.array(ts[:], dtype=np.float64)
        acc += (a - a[0]) * 0.01  # 100 MHz -> us
    acc /= n
    print(f"B={B} S={S}")
    prev = 0.0
    for i in [0,1,2,3,4,5,20,21,22,23,24,25,6,7,10,11,12,13,14,15,16,17]:
        print(f"  {names[i]:18s} {acc[i]:7.2f} us  (+{acc[i]-prev:5.2f})")
        prev = acc[i]
