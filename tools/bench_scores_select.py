import sys, torch
sys.path.insert(0, '/root/repo')
import bench, nsa_vibe_amd as nv
dev = torch.device("cuda", 0)
for S, B in ((4096, 8), (4096, 1), (8192, 4), (16384, 2), (32768, 2), (65536, 1), (65536, 4)):
    meta, Q, Kc, K, V = bench.make_inputs(nv, B, S, dev, 1)
    f = lambda: nv.selection_scores_select(Q, Kc, meta, 16, mode="batched")
    def g():
        p = nv.selection_scores(Q, Kc, meta, causal_skip=True, leave_skipped=True)
        return nv.select_topn_ranges_batched(p, meta, 16, S)
    res = []
    for _ in range(2):
        nv._lib.set_tuning("SCORES_SELECT", -1); a = bench.time_events(f, 8, warm=2)
        nv._lib.set_tuning("SCORES_SELECT", 0); b = bench.time_events(f, 8, warm=2)
        c = bench.time_events(g, 8, warm=2)
        res.append((a, b, c))
    a, b, c = (min(r[i] for r in res) for i in range(3))
    print(f"S={S} B={B}: one launch {a*1e3:8.1f} us | two launches behind one call {b*1e3:8.1f} us | two python calls {c*1e3:8.1f} us", flush=True)
    del Q, Kc, K, V
