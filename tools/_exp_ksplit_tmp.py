import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, nsa_vibe_amd as nv
dev = torch.device("cuda", 0)
for S, B in ((65536, 1), (65536, 4), (4096, 8), (16384, 2)):
    meta, Q, Kc, K, V = bench.make_inputs(nv, B, S, dev, 1234)
    p = nv.selection_scores(Q, Kc, meta, 0.125, causal_skip=True, leave_skipped=True)
    rg = nv.select_topn_ranges_batched(p, meta, bench.N_SEL, S)
    del p
    def t(v):
        nv._lib.set_tuning("SEL_KSPLIT", v)
        with torch.no_grad():
            return bench.time_events(lambda: nv.selection_attention_hip(Q, K, V, rg), 6, warm=2) * 1e3
    full = min(t(0), t(0))
    two = sum(min(t(20 + c + 1), t(20 + c + 1)) for c in range(2))
    four = sum(min(t(40 + c + 1), t(40 + c + 1)) for c in range(4))
    setup = min(t(91), t(91))
    nv._lib.set_tuning("SEL_KSPLIT", 0)
    print(f"S={S} B={B}: full {full:8.1f} us | 2-way sum {two:8.1f} ({two/full:.2f}x) | 4-way sum {four:8.1f} ({four/full:.2f}x) | no blocks at all (setup only) {setup:8.1f} us", flush=True)
    del Q, Kc, K, V
    torch.cuda.empty_cache()
