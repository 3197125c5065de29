#!/usr/bin/env python3
"""Benchmark of the selected-branch attention hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1 without WORLD_SIZE: bench.py starts the N ranks itself)

A "step" = one pass of the hot path over one batch of synthetic input for one layer:
    Q,K_cmp -> p_grp (softmax scores, Eq.9, Eq.10) -> deterministic top-n ranges -> selection attention
(two launches + the merge launch of the key-split form: the fused scorer with the top-n selection in its epilogue, then the attention kernel)
on the m7c_125m shape (dim 768: 12 heads, G=2, h=6, d_k=d_v=64; l=32 d=16 l'=64 n=16) at S=65536 (north_star's target
length, BASELINE.json configs[3]), bf16, B=16 sequences per GPU: K/V = 512 MiB, twice the Infinity Cache, so the inputs
really live in HBM.  Inputs are resident in HBM before the timed region.  (--seq/--batch select the other BASELINE shapes;
S=4096 B=8, S=16384 and S=65536 B=1 are reported under `extra`.)
Multi-GPU: the batch x group axis is sharded, every rank runs its own B sequences, no data-path collective exists on this
path (weak scaling); time = max over ranks.

The JSON line also carries
  roofline         dominant kernel of the step (the selection-attention launches; the step is scores + select in one launch, then the attention
                   and the merge launch of its key-split form from 32k keys on).  Top level = the roof SURVEY 8(d) names for the timed (prefill)
                   form: in-block QK^T + PV flops against the dense bf16 MFMA peak, a fraction <= 1 (qk_frac = the QK^T half).  The
                   memory-side views sit beside it, nested and as scalar keys: hbm_traffic_frac (rocprofv3 PMC bytes, profiles/r0x/traffic_*.json,
                   same shape, / HIP-event time vs 8 TB/s), l2_frac (the 64-key blocks the waves bring into LDS vs the ~34.5 TB/s aggregate);
                   algorithmic_gather_GBps (SURVEY 8(d)'s byte formula / time: above the HBM peak, L2 re-use) is a plain field.
  decode_roofline  the HBM-bound configuration north_star names: one decode step of B sequences at context S reads
                   sum_rows L_row*(Dk+Dv)*2 B of selected K/V plus the compressed keys (S_cmp*Dk*2 B per (b,g)) exactly once
                   (reads formula of nsa/core/nsa_attention.py:634-635, bytes formula of triton_sel_kernel/__init__.py:483);
                   achieved = those bytes / COLD step time vs the 8 TB/s HBM peak.  Cold = the steps rotate over independent cache sets
                   (>= 512 MiB loaded between two uses of a line); the warm figure (same set back to back: Infinity-Cache served) is kept
                   beside it.  Top level = the S = 65536 configuration with the highest cold fraction; every shape is in `extra`.
  cpu_baseline     the CPU oracle (a port of the reference path, validated against the reference) timed on this node's host
                   cores on a bounded sample of the same workload
  library          path and sha256 of the libnsa_sel_hip.so that ran (NSA_HIP_LIB can point a measurement at an A/B build)
  extra            decode tok/s (cold) and prefill ms at S in {4k,16k,64k} of the hot path, MFMA TFLOP/s of the attention kernel,
                   selection backward with its roofline, the sliding/compressed branch kernel, and the whole NSAAttention layer (native path)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16
L2_PEAK_GBPS = 34500.0  # same guide, "L2 (per XCD)": 32 MiB aggregate, ~34.5 TB/s

G, H, D = 2, 6, 64
L_CMP, D_CMP, L_SEL, N_SEL = 32, 16, 64, 16


def make_inputs(nv, B, S, device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    meta = nv.build_block_meta(S, L_CMP, D_CMP, L_SEL, N_SEL, 512)
    Q = torch.randn(B, S, G, H, D, device=device, generator=g).bfloat16()
    Kc = torch.randn(B, G, meta.S_cmp, D, device=device, generator=g).bfloat16()
    K = torch.randn(B, G, S, D, device=device, generator=g).bfloat16()
    V = torch.randn(B, G, S, D, device=device, generator=g).bfloat16()
    return meta, Q, Kc, K, V


def hot_path(nv, meta, Q, Kc, K, V, S):
    # scores (blocks both selectors mask to -inf at row t are not computed) + batched top-n (select_topn_ranges_batched semantics) in ONE
    # launch: a scorer workgroup selects the ranges of its 64 query rows right behind its second sweep (nsa_sel_scores_select) ...
    p_grp, ranges = nv.selection_scores_select(Q, Kc, meta, N_SEL, mode="batched")
    # ... then the selection attention over the ranges (they are an output of the path)
    with torch.no_grad():
        return ranges, nv.selection_attention_hip(Q, K, V, ranges)


def time_events(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    # median: equals the mean for these kernels except when the shared box inserts a rare multi-millisecond stall
    return float(np.median([a.elapsed_time(b) for a, b in evs]))  # ms


def stage_times(nv, meta, Q, Kc, K, V, S, iters):
    """per-stage HIP-event times.  The step runs three launches: scores, the select kernel, the selection-attention kernel (the last two
    behind one native call, nsa_sel_select_attn_fwd).  The attention kernel is the dominant one: it carries the roofline."""
    p_grp = nv.selection_scores(Q, Kc, meta, causal_skip=True, leave_skipped=True)
    ranges = nv.select_topn_ranges_batched(p_grp, meta, N_SEL, S)
    t_sc = time_events(lambda: nv.selection_scores(Q, Kc, meta, causal_skip=True, leave_skipped=True), iters)
    t_scsel = time_events(lambda: nv.selection_scores_select(Q, Kc, meta, N_SEL, mode="batched"), iters)
    t_sel = time_events(lambda: nv.select_topn_ranges_batched(p_grp, meta, N_SEL, S), iters)
    with torch.no_grad():
        t_att = time_events(lambda: nv.selection_attention_hip(Q, K, V, ranges), iters)
    t_sa = time_events(lambda: nv.select_and_attend(p_grp, Q, K, V, meta, N_SEL, mode="batched"), iters)
    L = (ranges[..., 1] - ranges[..., 0]).clamp_min(0).sum(-1).double()
    return t_sc, t_sel, t_att, t_sa, float(L.sum().item()), float(L.mean().item()), gathered_tiles(ranges, K.shape[2]), t_scsel


BLK_KEYS = 64                           # keys of one K/V block of the block-form kernel (sel_attn_blocks_mfma.hip)
BLK_ROWS = min(32, 4 * max(1, 16 // H))  # query rows one wave owns there (NT = 4 column tiles of 16 // h rows)


def gathered_tiles(ranges, S_kv, tpw=BLK_ROWS, keys=BLK_KEYS):
    """K/V blocks (of `keys` keys) the attention kernel really brings into LDS: the union over the tpw rows one wave owns"""
    Bq, Sq, Gq = ranges.shape[:3]
    sh = keys.bit_length() - 1
    nt = (S_kv + keys - 1) // keys
    total = 0.0
    for b in range(Bq):  # one sequence at a time: the cover map of a 64k sequence is 0.5 GB
        rb = ranges[b: b + 1]
        s, e = rb[..., 0].long().clamp(0, S_kv), rb[..., 1].long().clamp(0, S_kv)
        live = (e > s).to(torch.int16)
        diff = torch.zeros(1, Sq, Gq, nt + 1, dtype=torch.int16, device=ranges.device)
        diff.scatter_add_(3, (s >> sh).clamp(max=nt), live)
        diff.scatter_add_(3, (((e - 1).clamp_min(0) >> sh) + 1).clamp(max=nt), -live)
        cover = diff.cumsum(3, dtype=torch.int16)[..., :nt] > 0
        del diff
        Sp = Sq // tpw * tpw
        n = cover[:, :Sp].reshape(1, Sp // tpw, tpw, Gq, nt).any(2).sum()
        if Sp < Sq:
            n = n + cover[:, Sp:].any(1).sum()
        total += float(n.item())
        del cover
    return total


def attention_extremes(nv, device, iters=5):
    """SURVEY 8(d)'s two synthetic extremes for the selection-attention kernel (the reference's bench/bench_sel_triton.py:13-34): every row
    takes (a) ONE span [0, L) and (b) n equal spans, the same ones for all rows -- the rows of a wave share every K/V block they touch, the
    opposite corner from the selector's random picks (where at 64k a block serves one row of a wave and 6 of 16 MFMA columns).  What the
    kernel reaches here is its own MFMA ceiling; what the headline shape lacks relative to it is the workload's sparsity structure."""
    out = {}
    S_kv, L = 65536, N_SEL * L_SEL
    for B_, S_ in ((4, 16384), (16, 4096)):
        g = torch.Generator(device=device)
        g.manual_seed(5)
        Q = torch.randn(B_, S_, G, H, D, device=device, generator=g).bfloat16()
        K = torch.randn(B_, G, S_kv, D, device=device, generator=g).bfloat16()
        V = torch.randn(B_, G, S_kv, D, device=device, generator=g).bfloat16()
        cases = {}
        one = torch.zeros(B_, S_, G, N_SEL, 2, dtype=torch.int32, device=device)
        one[..., 0, 1] = L
        cases["single_span_0_1024"] = one
        eq = torch.zeros(B_, S_, G, N_SEL, 2, dtype=torch.int32, device=device)
        starts = torch.arange(N_SEL, device=device, dtype=torch.int32) * (S_kv // N_SEL)
        eq[..., 0] = starts
        eq[..., 1] = starts + L_SEL
        cases["16_equal_spans_of_64_same_for_all_rows"] = eq
        rnd = torch.zeros(B_, S_, G, N_SEL, 2, dtype=torch.int32, device=device)  # for contrast: 16 random blocks per row, no sharing
        blk = torch.rand(B_, S_, G, S_kv // L_SEL, device=device, generator=g).topk(N_SEL, dim=-1).indices.sort(dim=-1).values.int()
        rnd[..., 0] = blk * L_SEL
        rnd[..., 1] = blk * L_SEL + L_SEL
        cases["16_random_blocks_per_row_no_sharing"] = rnd
        for name, rg in cases.items():
            with torch.no_grad():
                ms = time_events(lambda: nv.selection_attention_hip(Q, K, V, rg), iters)
            fl = 4.0 * H * L * D * B_ * S_ * G
            tf = fl / (ms * 1e-3) / 1e12
            out[f"{name}_B{B_}_S{S_}"] = {"ms": ms, "rows": B_ * S_ * G, "keys_per_row": L, "tflops": tf, "attn_mfma_frac": tf / MFMA_BF16_PEAK_TFLOPS,
                                          "qk_frac": 0.5 * tf / MFMA_BF16_PEAK_TFLOPS}
        del Q, K, V, cases, one, eq, rnd, blk
        torch.cuda.empty_cache()
    return out


COLD_BYTES_BETWEEN_USES = 512 * 2 ** 20  # bytes other cache sets load between two uses of one set's lines: twice the 256 MiB Infinity Cache


def decode_cache_sets(nv, B, S_ctx, device, n_sets, seed=7):
    """n_sets independent decode states (q, K_cmp, K, V) of B sequences at context S_ctx -- the layers of a model, each with its own cache and
    its own query (so its own selected blocks).  Generated in bf16 on the device (no fp32 temporaries: K/V of B=256 @64k are 8 GiB per set)."""
    meta = nv.build_block_meta(S_ctx, L_CMP, D_CMP, L_SEL, N_SEL, 512)
    sets = []
    for i in range(n_sets):
        g = torch.Generator(device=device)
        g.manual_seed(seed + 1009 * i)
        mk = lambda *sh: torch.randn(*sh, device=device, generator=g, dtype=torch.bfloat16)  # noqa: E731
        sets.append((mk(B, 1, G, H, D), mk(B, G, meta.S_cmp, D), mk(B, G, S_ctx, D), mk(B, G, S_ctx, D)))
    return meta, sets


def decode_step_bytes(B, S_ctx):
    """bytes one decode step must read, each exactly once: sum_rows L_row*(Dk+Dv)*2 (selected K/V, L_row <= n*l') + B*G*S_cmp*Dk*2 (compressed keys)"""
    S_cmp = (S_ctx - L_CMP) // D_CMP + 1
    return B * G * (min(N_SEL * L_SEL, S_ctx) * (D + D) * 2 + S_cmp * D * 2)


def decode_bench(nv, B, S_ctx, steps, device, max_sets=32):
    """Decode-shaped hot path: B sequences at context S_ctx, one new token each (sequential-mode selector, preallocated K/V cache passed as a
    strided view -- no torch.cat append as in nsa/cache/kv_cache.py:28-30), timed the way a model's decode loop issues it
    (bench/bench_decode.py:123-136): steps back to back, each on ANOTHER layer's cache.

    COLD (the figure the roofline uses): the steps rotate over n_sets independent (q, K_cmp, K, V) sets -- n_sets chosen so that the other
    sets load >= 512 MiB between two uses of any line (twice the Infinity Cache; 12 sets at B=64 @16k, the layer count of m7c_125m), every
    set with its own query, hence its own 13 scored blocks: every byte comes from HBM.  WARM (kept for the record; rounds 1-3 reported it):
    the same set 20 times back to back -- the <= 201 MB a step touches then stay in the 256 MiB Infinity Cache."""
    step_bytes = decode_step_bytes(B, S_ctx)
    n_sets = min(max_sets, -(-COLD_BYTES_BETWEEN_USES // step_bytes) + 1)
    meta, sets = decode_cache_sets(nv, B, S_ctx, device, n_sets)
    t = S_ctx - 1
    O = torch.empty(B, 1, G, H, D, device=device, dtype=torch.bfloat16)
    rg = torch.empty(B, G, N_SEL, 2, device=device, dtype=torch.int32)
    nb = 20
    pos = [0]

    def cold_batch():  # scores -> sequential top-n -> attention in one native call (nsa_sel_decode_step) per step
        for _ in range(nb):
            q1, Kc, K, V = sets[pos[0] % n_sets]
            pos[0] += 1
            nv.selection_decode_step(q1, Kc, K, V, meta, N_SEL, t, out=O, ranges_out=rg)

    def warm_batch():
        q1, Kc, K, V = sets[0]
        for _ in range(nb):
            nv.selection_decode_step(q1, Kc, K, V, meta, N_SEL, t, out=O, ranges_out=rg)

    nrep = max(3, steps // nb * 3)
    ms_warm = time_events(warm_batch, nrep, warm=1) / nb
    ms_cold = time_events(cold_batch, nrep, warm=1) / nb
    Lsum = 0.0
    for q1, Kc, K, V in sets:  # selected tokens over all (b,g) rows, mean over the sets
        nv.selection_decode_step(q1, Kc, K, V, meta, N_SEL, t, out=O, ranges_out=rg)
        Lsum += float((rg[..., 1] - rg[..., 0]).clamp_min(0).sum().item())
    L = Lsum / n_sets
    gather_bytes = L * (D + D) * 2  # L_row * (Dk+Dv) * sizeof(bf16), K/V once per group (triton_sel_kernel/__init__.py:483)
    kcmp_bytes = float(B * G * meta.S_cmp * D * 2)  # the scorer reads every compressed key of every (b,g) once
    between = (n_sets - 1) * (gather_bytes + kcmp_bytes)
    del sets
    torch.cuda.empty_cache()
    return {"tok_per_s": B / (ms_cold * 1e-3), "ms_per_step": ms_cold, "ms_per_step_cold": ms_cold, "ms_per_step_warm": ms_warm,
            "tok_per_s_warm": B / (ms_warm * 1e-3), "steps_per_timed_batch": nb, "context": S_ctx, "batch": B,
            "cache_sets": n_sets, "bytes_loaded_between_two_uses_of_a_set": between, "cold": bool(between >= COLD_BYTES_BETWEEN_USES),
            "selected_tokens_per_row": L / (B * G), "gather_bytes": gather_bytes, "kcmp_bytes": kcmp_bytes,
            "kv_resident_bytes": float(n_sets * 2 * B * G * S_ctx * D * 2)}


def decode_roofline(d, traffic):
    """HBM roofline of one decode step: bytes that must be read once / COLD step time vs the 8 TB/s peak"""
    alg = d["gather_bytes"] + d["kcmp_bytes"]
    ach = alg / (d["ms_per_step_cold"] * 1e-3) / 1e9
    ach_w = alg / (d["ms_per_step_warm"] * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": traffic,
            "cold": d["cold"], "cache_sets": d["cache_sets"], "bytes_loaded_between_two_uses_of_a_set": d["bytes_loaded_between_two_uses_of_a_set"],
            "warm_same_set_GBps": ach_w, "warm_same_set_frac_not_an_hbm_figure": ach_w / HBM_PEAK_GBPS,
            "algorithmic_bytes_per_step": alg, "gather_bytes": d["gather_bytes"], "kcmp_bytes": d["kcmp_bytes"],
            "step_ms": d["ms_per_step_cold"], "step_ms_warm": d["ms_per_step_warm"],
            "workload": f"decode step, B={d['batch']} sequences at context {d['context']}, rotating over {d['cache_sets']} independent cache sets "
            f"(K/V resident: {d['kv_resident_bytes'] / 2 ** 20:.0f} MiB; {d['bytes_loaded_between_two_uses_of_a_set'] / 2 ** 20:.0f} MiB loaded between two uses of a set), "
            "one native call per step (scores -> top-n -> attention)",
            "formula": "sum_rows L_row*(Dk+Dv)*2 B + B*G*S_cmp*Dk*2 B, each read once (nsa_attention.py:634-635; "
                       "triton_sel_kernel/__init__.py:483)"}


def band_bench(nv, B, S, device, iters=5):
    """sliding-window (w=512) and compressed (l=32, d=16) branch kernels: ms and MFMA TFLOP/s (4*h*D flops per (row, key))"""
    g = torch.Generator(device=device)
    g.manual_seed(5)
    mk = lambda *sh: torch.randn(*sh, device=device, generator=g).bfloat16()  # noqa: E731
    S_cmp = (S - L_CMP) // D_CMP + 1
    Q, K, V, Kc, Vc = mk(B, S, G, H, D), mk(B, G, S, D), mk(B, G, S, D), mk(B, G, S_cmp, D), mk(B, G, S_cmp, D)
    t = torch.arange(S)
    keys_win = int(torch.clamp(t + 1, max=512).sum())
    keys_cmp = int(torch.where(t + 1 < L_CMP, 0, (t + 1 - L_CMP) // D_CMP + 1).sum())
    out = {}
    for name, fn, keys in (("win", lambda: nv.sliding_window_attention(Q, K, V, 512), keys_win),
                           ("cmp", lambda: nv.batched_causal_attention_compressed(Q, Kc, Vc, L_CMP, D_CMP), keys_cmp)):
        ms = time_events(fn, iters)
        out[name] = {"ms": ms, "tflops": 4.0 * B * G * H * D * keys / (ms * 1e-3) / 1e12}
    return out


def backward_bench(nv, meta, Q, K, V, S, iters=5):
    """selection attention forward + backward (autograd through the HIP kernels) on the bench workload, with the backward's roofline:
    algorithmic flops 10*h*L*D per row (S = Q K^T, dP = dO V^T, dV = P^T dO, dK = dS^T Q, dQ = dS K: five products of 2*h*L*D; the analytic
    backward of nsa/kernels/triton_sel_kernel/__init__.py:163-231) against the dense bf16 MFMA peak.  The kernels recompute S and dP in both the
    query-major dQ pass and the key-block-major dK/dV pass (14*h*L*D issued): no atomics, bitwise reproducible."""
    B = Q.shape[0]
    g = torch.Generator(device=Q.device)
    g.manual_seed(6)
    rg = nv.select_topn_ranges_batched(torch.rand(B, S, G, meta.S_sel, device=Q.device, generator=g), meta, N_SEL, S)
    dO = torch.randn(Q.shape, device=Q.device, generator=g).bfloat16()
    q, k, v = (x.clone().requires_grad_(True) for x in (Q, K, V))
    tf = tb = 0.0
    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    for i in range(iters + 2):
        q.grad = k.grad = v.grad = None
        a.record()
        O = nv.selection_attention_hip(q, k, v, rg)
        b.record()
        O.backward(dO)
        c.record()
        torch.cuda.synchronize()
        if i >= 2:
            tf += a.elapsed_time(b) / iters
            tb += b.elapsed_time(c) / iters
    L = float((rg[..., 1] - rg[..., 0]).clamp_min(0).sum().item())
    fl = 10.0 * H * L * D
    tfl = fl / (tb * 1e-3) / 1e12
    return {"fwd_ms": tf, "bwd_ms": tb, "bwd_over_fwd": tb / tf,
            "roofline_bwd": {"kernel": "bwd_delta + bwd_dq_rows + bwd_hitmap + bwd_dkdv + bwd_reduce (the backward launches of one selection_attention_hip call)",
                             "bound": "mfma", "achieved": tfl, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / MFMA_BF16_PEAK_TFLOPS,
                             "algorithmic_flops": fl, "issued_flops_with_recompute": 1.4 * fl, "kernel_ms": tb,
                             "traffic": pmc_traffic(f"sel_bwd_S{S}_B{B}")}}


def layer_bench(nv, B, S, device, steps=40):
    """the whole NSAAttention layer (cmp + sel + win branches, gate, projections) on the native path: prefill ms, decode tok/s"""
    torch.manual_seed(0)
    m = nv.NSAAttention(768, 12, G, D, D, L_CMP, D_CMP, L_SEL, N_SEL, 512, selector="batched").to(device).to(torch.bfloat16).eval()
    x = torch.randn(B, S, 768, device=device, dtype=torch.bfloat16)
    with torch.no_grad():
        best = 1e9
        for _ in range(4):
            kv = m.new_kv(B, S + steps + 16, device, torch.bfloat16)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, kv = m(x, kv, prefill=True)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        xt = torch.randn(B, 1, 768, device=device, dtype=torch.bfloat16)
        for _ in range(8):
            _, kv = m(xt, kv, prefill=False)
        # blocks of 10 steps, best block: a shared box shows occasional multi-millisecond stalls unrelated to the work
        dt = 1e9
        for _ in range(max(1, steps // 10)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                _, kv = m(xt, kv, prefill=False)
            torch.cuda.synchronize()
            dt = min(dt, (time.perf_counter() - t0) / 10)
    return {"prefill_ms": best * 1e3, "prefill_tok_per_s": B * S / best, "decode_us_per_step": dt * 1e6, "decode_tok_per_s": B / dt}


def layer_train_bench(nv, B, S, device, iters=5):
    """forward + backward of the NSAAttention layer with autograd (config 5 shape): fused projection GEMM, then every stage
    (RoPE/append, pooling, the three attention branches, gate/combine) is a differentiable native op with its backward kernel"""
    torch.manual_seed(0)
    m = nv.NSAAttention(768, 12, G, D, D, L_CMP, D_CMP, L_SEL, N_SEL, 512, selector="batched").to(device).to(torch.bfloat16).train()
    x = torch.randn(B, S, 768, device=device, dtype=torch.bfloat16, requires_grad=True)
    go = torch.randn(B, S, 768, device=device, dtype=torch.bfloat16)
    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    tfs, tbs = [], []
    for i in range(iters + 3):
        m.zero_grad(set_to_none=True)
        x.grad = None
        kv = m.new_kv(B, S, device, torch.bfloat16)
        a.record()
        out, _ = m(x, kv, prefill=True)
        b.record()
        out.backward(go)
        c.record()
        torch.cuda.synchronize()
        if i >= 3:
            tfs.append(a.elapsed_time(b))
            tbs.append(b.elapsed_time(c))
    tf, tb = float(np.median(tfs)), float(np.median(tbs))
    return {"fwd_ms": tf, "bwd_ms": tb, "tok_per_s": B * S / ((tf + tb) * 1e-3)}


def model_bench(B, S, device, steps=24):
    """m7c_125m TinyLM (12 x LlamaBlockNSA, GPT-2 vocabulary, random weights): prefill ms and decode tokens/s of the whole model"""
    from nsa_vibe_amd.llama_block_nsa import TinyLM

    torch.manual_seed(0)
    lm = TinyLM(50257, 768, 12, 12, G, D, D, L_CMP, D_CMP, L_SEL, N_SEL, 512, selector="batched").to(device).to(torch.bfloat16).eval()
    tok = torch.randint(0, 50257, (B, S), device=device)
    with torch.no_grad():
        best = 1e9
        for _ in range(3):
            caches = lm.new_caches(B, S + steps + 16, device, torch.bfloat16)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            logits = lm.prefill(tok, caches)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        nxt = logits.argmax(-1)
        for _ in range(8):
            nxt = lm.decode(nxt, caches, return_next=True)[1]
        dt = 1e9
        for _ in range(max(1, steps // 8)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                nxt = lm.decode(nxt, caches, return_next=True)[1]
            torch.cuda.synchronize()
            dt = min(dt, (time.perf_counter() - t0) / 8)
    return {"prefill_ms": best * 1e3, "decode_ms_per_token": dt * 1e3, "decode_tok_per_s": B / dt}


def cpu_baseline(S, B, seed=3, min_seconds=10.0):
    """The oracle (CPU restatement of the reference path, fp32) on the same workload.  Bounded sample: at S <= 8192 whole
    sequences, repeated until ~10 s of host time have been spent; beyond that every `stride`-th query row of one sequence
    (scores and attention on the sampled rows against the FULL K_cmp / K / V; the batched selector runs over all rows and is
    charged pro rata), repeated until ~10 s."""
    from oracle import nsa_oracle as orc

    orc.build()
    rng = np.random.default_rng(seed)
    meta = orc.build_block_meta(S, L_CMP, D_CMP, L_SEL, N_SEL, 512)
    S_cmp = meta.cmp_starts.size
    stride = 1 if S <= 8192 else S // 4096
    rows = np.arange(stride // 2, S, stride)
    Q = rng.standard_normal((1, rows.size, G, H, D), dtype=np.float32)
    Kc = rng.standard_normal((1, G, S_cmp, D), dtype=np.float32)
    K = rng.standard_normal((1, G, S, D), dtype=np.float32)
    V = rng.standard_normal((1, G, S, D), dtype=np.float32)
    tot = [0.0, 0.0, 0.0]
    nrep = 0
    p_full = np.zeros((1, S, G, meta.sel_starts.size), np.float32) if stride > 1 else None
    while sum(tot) < min_seconds and nrep < 64 * B:
        t0 = time.perf_counter()
        p_cmp = orc.compute_pcmp_all(Q, Kc, 1.0 / 8.0)
        _, p_grp = orc.map_pcmp_to_pslc_and_pgrp(p_cmp, meta)
        del p_cmp
        t1 = time.perf_counter()
        if stride > 1:
            p_full[0, rows] = p_grp[0]
            r = orc.select_topn_ranges_batched(p_full, meta, N_SEL, S)[:, rows]
        else:
            r = orc.select_topn_ranges_batched(p_grp, meta, N_SEL, S)
        t2 = time.perf_counter()
        orc.sel_attention_masked(Q, K, V, r)
        t3 = time.perf_counter()
        tot = [tot[0] + t1 - t0, tot[1] + (t2 - t1) / stride, tot[2] + t3 - t2]
        nrep += 1
    what = (f"{nrep} sequences of S={S} (every row)" if stride == 1 else
            f"{nrep} x {rows.size} query rows (every {stride}th row) of one S={S} sequence against the full K_cmp/K/V, selector pro rata")
    return {"value": nrep * rows.size / sum(tot), "unit": "tok/s", "cores": orc.num_threads(), "kind": "port",
            "sample": f"{what} (m7c, fp32), OpenMP over rows: scores {tot[0]:.2f}s select {tot[1]:.2f}s attention {tot[2]:.2f}s",
            "cpu_model": _cpu_model()}


def pmc_traffic(tag, key="traffic_bytes"):
    """HBM bytes per launch for a workload, measured with rocprofv3 PMC counters in separate passes (tools/pmc_traffic.sh) and
    committed under profiles/ (bench.py cannot run the profiler on itself); None if not measured."""
    for rnd in ("r04", "r03", "r02", "r01"):
        try:
            return float(json.load(open(os.path.join(ROOT, "profiles", rnd, f"traffic_{tag}.json")))[key])
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return None


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def train_mode(nv, args, dist, world, rank, device):
    """DDP training step of one NSAAttention layer (m7c shape, bf16): every rank B sequences, bucketed gradient all-reduce over
    RCCL overlapped with the backward by DistributedDataParallel; time = max over ranks, value = whole-job tokens/s."""
    from torch.nn.parallel import DistributedDataParallel as DDP

    B, S = args.batch, args.seq
    torch.manual_seed(0)  # identical initial weights on every rank
    layer = nv.NSAAttention(768, 12, G, D, D, L_CMP, D_CMP, L_SEL, N_SEL, 512, selector="batched").to(device).to(torch.bfloat16).train()

    class Wrap(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, x):
            return self.m(x, self.m.new_kv(x.shape[0], x.shape[1], x.device, x.dtype), prefill=True)[0]

    model = Wrap(layer)
    if dist is not None:
        model = DDP(model, device_ids=[device.index], bucket_cap_mb=25)
    opt = torch.optim.SGD(layer.parameters(), lr=1e-4)
    g = torch.Generator(device=device)
    g.manual_seed(1234 + rank)
    x = torch.randn(B, S, 768, device=device, generator=g).bfloat16()
    go = torch.randn(B, S, 768, device=device, generator=g).bfloat16()

    def step():
        opt.zero_grad(set_to_none=True)
        model(x).backward(go)
        opt.step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    ranks = rank_report(dist, world, elapsed, args.steps, device)
    if dist is not None:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        print(json.dumps({
            "metric": "nsa_layer_train_tok_per_s", "value": world * B * S / (elapsed / args.steps), "unit": "tok/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"one m7c_125m NSAAttention layer, forward+backward+SGD step, S={S}, B={B} per GPU, DDP "
                                   f"(gradient all-reduce of {sum(p.numel() for p in layer.parameters())} bf16 parameters per step)",
                       "global_batch": world * B, "seq_len": S, "parallelism": f"dp{world} (DistributedDataParallel over RCCL)"},
            "ranks": ranks}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def train_model_mode(nv, args, dist, world, rank, device):
    """BASELINE config 5: the m7c_125m TinyLM (12 x LlamaBlockNSA, dim 768, GPT-2 vocabulary as configs/m7c_125m_80g.yaml names it) trained
    on synthetic random tokens (train_showcase.py:1222): forward, cross-entropy, backward, gradient clipping at 1.0 and a fused AdamW
    step (lr 2e-4, weight decay 0.01), bf16 weights.  Every rank B sequences; DistributedDataParallel overlaps the bucketed RCCL
    all-reduce with the backward.  time = max over ranks, value = whole-job tokens/s."""
    from torch.nn.parallel import DistributedDataParallel as DDP

    from nsa_vibe_amd.llama_block_nsa import TinyLM

    B, S, V = args.batch, args.seq, args.vocab
    torch.manual_seed(0)  # identical initial weights on every rank
    lm = TinyLM(V, 768, args.layers, 12, G, D, D, L_CMP, D_CMP, L_SEL, N_SEL, 512, selector="batched").to(device).to(torch.bfloat16).train()
    model = DDP(lm, device_ids=[device.index], bucket_cap_mb=25, gradient_as_bucket_view=True) if dist is not None else lm
    opt = torch.optim.AdamW(lm.parameters(), lr=2e-4, weight_decay=0.01, fused=True)
    g = torch.Generator(device=device)
    g.manual_seed(1337 + rank)
    tok = torch.randint(0, V, (B, S + 1), device=device, generator=g)
    x, y = tok[:, :-1].contiguous(), tok[:, 1:].contiguous()
    loss_box = [None]

    def step():
        opt.zero_grad(set_to_none=True)
        logits = model(x)
        loss = torch.nn.functional.cross_entropy(logits.view(-1, V).float(), y.view(-1))
        loss.backward()
        torch.nn.utils.clip_grad_norm_(lm.parameters(), 1.0)
        opt.step()
        loss_box[0] = loss

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    first = float(loss_box[0]) if loss_box[0] is not None else float("nan")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    ranks = rank_report(dist, world, elapsed, args.steps, device)
    if dist is not None:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        n_par = sum(p.numel() for p in lm.parameters())
        print(json.dumps({
            "metric": "m7c_125m_train_tok_per_s", "value": world * B * S / (elapsed / args.steps), "unit": "tok/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"m7c_125m TinyLM ({args.layers} NSA blocks, dim 768, vocab {V}, {n_par / 1e6:.1f} M parameters): forward + "
                                   f"cross-entropy + backward + clip + fused AdamW, S={S}, B={B} per GPU, random tokens and weights",
                       "global_batch": world * B, "seq_len": S, "parallelism": f"dp{world} (DistributedDataParallel over RCCL)"},
            "ranks": ranks, "loss_after_warmup": first, "loss_last": float(loss_box[0]),
            "peak_mem_GiB": torch.cuda.max_memory_allocated(device) / 2 ** 30}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def rank_report(dist, world, elapsed_local, steps, device):
    """what lets the driver check a multi-rank line against itself: every rank contributes a 1 to an all-reduce over the measurement's
    own process group (ranks_seen must equal --gpus: asserted), and the per-rank step times are reduced to their min / max (the line's
    ms_per_step is the max).  Pattern: scripts/train_showcase.py:425-437, 718-723 (rank / world bookkeeping of the reference's trainer)."""
    ms = elapsed_local / steps * 1e3
    if dist is None:
        return {"ranks_seen": 1, "ms_per_step_rank_min": ms, "ms_per_step_rank_max": ms, "backend": "none (single process)", "rccl": False}
    one = torch.ones(1, device=device, dtype=torch.float64)
    dist.all_reduce(one)
    lo = torch.tensor([ms], device=device, dtype=torch.float64)
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    seen = int(round(one.item()))
    assert seen == world, f"all-reduce saw {seen} rank(s), the launcher promised {world}"
    backend = dist.get_backend()
    return {"ranks_seen": seen, "ms_per_step_rank_min": float(lo.item()), "ms_per_step_rank_max": float(hi.item()), "backend": backend,
            "rccl": backend == "nccl"}


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher environment: start the N ranks ourselves, BEFORE anything touches the GPU --
    a child `python -m torch.distributed.run` (one process per GPU, rendezvous on 127.0.0.1); this process only waits and exits
    with the child's code (never re-exec: a process that has initialised the GPU must not be replaced).  PG-init pattern of the
    reference's trainer: scripts/train_showcase.py:425-437."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def dry_main(args, world, rank):
    """--dry: the multi-rank plumbing of the bench (rendezvous, barrier, max-over-ranks timing, one JSON line from rank 0) on the CPU
    with the gloo backend and a stand-in step -- what the world-size-2 CPU test runs.  Never a measurement."""
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.backend)
    x = torch.ones(64, 64) * (rank + 1)

    def step():
        return (x @ x).sum()

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ranks = rank_report(dist if world > 1 else None, world, elapsed, args.steps, torch.device("cpu"))
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        print(json.dumps({"metric": "dry_run_plumbing_only", "value": world * args.batch * args.seq / max(elapsed / args.steps, 1e-9), "unit": "tok/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "dry run (CPU, no kernels): launcher / rendezvous / timing plumbing only",
                                     "global_batch": world * args.batch, "parallelism": f"{world} rank(s), backend {args.backend}"},
                          "ranks": ranks}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="sequences per GPU (default 16 at S=65536, 8 otherwise)")
    ap.add_argument("--seq", type=int, default=65536)
    ap.add_argument("--no-extra", action="store_true", help="skip the decode / 4k / 16k extras and the CPU baseline")
    ap.add_argument("--train", action="store_true",
                    help="instead of the hot path: forward+backward of the whole NSAAttention layer under DistributedDataParallel "
                         "(BASELINE config 5: gradient all-reduce over RCCL/xGMI, batch sharded over the ranks)")
    ap.add_argument("--train-model", action="store_true",
                    help="instead of the hot path: one optimiser step of the whole m7c_125m TinyLM under DistributedDataParallel "
                         "(BASELINE config 5 in full: 12 NSA blocks, cross-entropy, clip, fused AdamW)")
    ap.add_argument("--layers", type=int, default=12, help="--train-model: number of blocks")
    ap.add_argument("--vocab", type=int, default=50257, help="--train-model: vocabulary (GPT-2, as configs/m7c_125m_80g.yaml)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only with --dry)")
    ap.add_argument("--dry", action="store_true", help="CPU rehearsal of the multi-rank plumbing (no GPU, no kernels, no measurement)")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 16 if (args.seq >= 65536 and not (args.train or args.train_model)) else 8
        if args.train or args.train_model:
            args.seq = 4096 if args.seq == 65536 else args.seq  # config 5 is quoted at S=4096

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))  # nothing has touched the GPU yet
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); refusing to report a mislabelled run", file=sys.stderr)
        sys.exit(2)
    if args.dry:
        return dry_main(args, world, rank)
    if args.backend != "nccl":
        print("bench.py: measurements run over RCCL (--backend nccl); gloo is for --dry only", file=sys.stderr)
        sys.exit(2)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    device = torch.device("cuda", torch.cuda.current_device())

    import nsa_vibe_amd as nv  # fails loudly if libnsa_sel_hip.so is missing

    B, S = args.batch, args.seq
    if args.train_model:
        return train_model_mode(nv, args, dist, world, rank, device)
    if args.train:
        return train_mode(nv, args, dist, world, rank, device)
    meta, Q, Kc, K, V = make_inputs(nv, B, S, device, 1234 + rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        hot_path(nv, meta, Q, Kc, K, V, S)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hot_path(nv, meta, Q, Kc, K, V, S)
    barrier()
    elapsed = time.perf_counter() - t0
    ranks = rank_report(dist, world, elapsed, args.steps, device)
    if dist is not None:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_step = elapsed / args.steps * 1e3
    value = world * B * S / (elapsed / args.steps)

    kv_mib = 2 * B * G * S * D * 2 / 2 ** 20
    out = {
        "metric": "sel_branch_hot_path_prefill_tok_per_s", "value": value, "unit": "tok/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"m7c_125m selected-branch hot path (scores->top-n ranges->selection attention), one layer, "
                               f"S={S}, B={B} per GPU (K/V {kv_mib:.0f} MiB resident in HBM), G={G} h={H} d_k=d_v={D}, l={L_CMP} d={D_CMP} "
                               f"l'={L_SEL} n={N_SEL}",
                   "global_batch": world * B, "seq_len": S, "parallelism": f"batch x group shard over {world} GPU(s), no collective"},
        "ranks": ranks,
    }
    from nsa_vibe_amd import _lib as _nsa_lib

    out["library"] = _nsa_lib.loaded_library()  # which build of libnsa_sel_hip.so ran (NSA_HIP_LIB can point at an A/B build: never the product figure)
    refused = None
    if rank == 0:
        t_sc, t_sel, t_att, t_sa, Lsum, Lmean, n_tiles, t_scsel = stage_times(nv, meta, Q, Kc, K, V, S, max(5, args.steps // 2))
        gathered = n_tiles * BLK_KEYS * (D + D) * 2
        alg_bytes = Lsum * (D + D) * 2  # L_row * (Dk+Dv) * sizeof(bf16), K/V counted once per group
        flops = 4.0 * H * Lsum * D  # 2*h*L*Dk (QK^T) + 2*h*L*Dv (PV) per row
        tfl = flops / (t_att * 1e-3) / 1e12
        traffic = pmc_traffic(f"S{S}_B{B}")
        l2 = {"bound": "l2", "achieved": gathered / (t_att * 1e-3) / 1e9, "peak": L2_PEAK_GBPS, "unit": "GB/s",
              "frac": gathered / (t_att * 1e-3) / 1e9 / L2_PEAK_GBPS, "gathered_bytes_per_launch": gathered,
              "note": "64-key K/V blocks really brought into LDS (the 8 rows of one wave share a block any of them selected) vs the "
                      "~34.5 TB/s aggregate L2 bandwidth of the guide"}
        common = {"kernel": "sel_attn_blocks_mfma_kernel (block-sparse selection attention over the rows' ranges; from 64k keys on as two key halves "
                            "on different XCDs + the merge launch, both inside kernel_ms and traffic)", "kernel_ms": t_att,
                  "mean_selected_tokens_per_row": Lmean, "algorithmic_gather_bytes_per_launch": alg_bytes,
                  "algorithmic_gather_GBps": alg_bytes / (t_att * 1e-3) / 1e9, "traffic": traffic}
        hbm = None
        if traffic is not None:
            ach = traffic / (t_att * 1e-3) / 1e9
            hbm = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                   "frac_of_measured_stream_rate": ach / 6290.0, "traffic": traffic,
                   "note": "HBM-side bytes of the attention launches (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE per the MI355X guide, "
                           "profiles/r03/traffic_*.json, same shape) / HIP-event time"}
        # Top level = the roof SURVEY 8(d) names for the timed (prefill) form of this kernel: the in-block QK^T / PV GEMMs against the dense
        # bf16 MFMA peak (2*h*L*Dk + 2*h*L*Dv flop per row, triton_sel_kernel/__init__.py:483 counts the same L), a fraction <= 1.  The
        # memory-side views sit beside it, nested AND as scalar keys (so they survive a parser that keeps scalars only): `hbm_traffic_frac` =
        # PMC HBM bytes / time against 8 TB/s, `l2_frac` = the 64-key blocks the waves really bring into LDS / time against the guide's
        # ~34.5 TB/s aggregate.  The ALGORITHMIC gather rate of 8(d) (sum_rows L_row*256 B / time) exceeds the HBM peak in prefill because the
        # rows of a (b,g) re-read its K/V out of L2 -- it is a plain field (`algorithmic_gather_GBps`), never a fraction.  The HBM-bound
        # configuration of this path is decode: `decode_roofline`.
        out["roofline"] = dict(common, bound="mfma", achieved=tfl, peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s", frac=tfl / MFMA_BF16_PEAK_TFLOPS,
                               qk_frac=0.5 * tfl / MFMA_BF16_PEAK_TFLOPS, algorithmic_flops_per_launch=flops,
                               hbm_traffic_frac=(hbm["frac"] if hbm else None), hbm_traffic_GBps=(hbm["achieved"] if hbm else None),
                               l2_frac=l2["frac"], l2_gathered_GBps=l2["achieved"],
                               note="achieved = in-block QK^T + PV flops (4*h*L*D per row, SURVEY 8(d)) / HIP-event time of the attention launches "
                                    "vs the dense bf16 MFMA peak; qk_frac = the QK^T half (north_star's target names it).  Memory side: "
                                    "hbm_traffic (PMC bytes / time vs 8 TB/s), l2 (gathered 64-key blocks through L2 -> LDS vs ~34.5 TB/s); "
                                    "algorithmic_gather_GBps is the 8(d) byte formula / time (above the HBM peak: L2 re-use), not a fraction",
                               hbm_traffic=hbm, l2=l2)
        for key in ("frac", "hbm_traffic_frac", "l2_frac"):
            v = out["roofline"][key]
            if v is not None and not (0.0 <= v <= 1.0):  # (decided here, acted on below: the other ranks still wait at the final barrier)
                refused = f"bench.py: roofline.{key} = {v} is not a fraction of its roof; refusing to print the line"
        out["stages_ms"] = {"scores_and_select_one_launch": t_scsel, "attention": t_att, "scores_alone": t_sc, "select_alone": t_sel,
                            "select_and_attention_one_call": t_sa,
                            "note": "per-call HIP-event medians (each call timed alone, with its launch gap); ms_per_step is the back-to-back loop.  "
                                    "The step is scores + select in one launch (the selector runs in the scorer's epilogue), then the attention "
                                    "launches; scores_alone / select_alone time the two kernels as separate launches for comparison"}
        flops_sc = 2.0 * B * S * G * H * meta.S_cmp * D
        out["roofline_scores"] = {"kernel": "scores_mfma32_kernel (fused p_cmp softmax + Eq.10 + Eq.9 on 32x32x16 tiles; 16x16 forms for other group sizes)", "bound": "mfma",
                                  "achieved": flops_sc / (t_sc * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": flops_sc / (t_sc * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, "kernel_ms": t_sc,
                                  "note": "algorithmic flops 2*B*S*heads*S_cmp*Dk (one pass; the kernel sweeps K_cmp twice because the "
                                          "reference normalises over all compressed columns)"}
        del Q, Kc, K, V
        torch.cuda.empty_cache()
        if not args.no_extra and world == 1:
            extra = {}
            try:
                # BASELINE metric: decode tok/s at S in {4k, 16k, 64k}; every figure COLD (rotating cache sets, decode_bench); the top-level
                # decode_roofline is the S = 65536 configuration (north_star's length) with the highest cold fraction
                best = None
                for Bd, Sd in ((64, 4096), (256, 4096), (64, 16384), (128, 16384), (256, 16384), (64, 65536), (128, 65536), (256, 65536), (512, 65536), (1, 65536)):
                    d = decode_bench(nv, Bd, Sd, 30, device)
                    extra[f"decode_B{Bd}_S{Sd}"] = d
                    rl = decode_roofline(d, pmc_traffic(f"decode_cold_B{Bd}_S{Sd}"))
                    extra[f"decode_roofline_B{Bd}_S{Sd}"] = rl
                    if Sd == 65536 and rl["cold"] and (best is None or rl["frac"] > best["frac"]):
                        best = rl
                if best is not None:
                    out["decode_roofline"] = best
                out["decode_tok_per_s_cold"] = {k[7:]: v["tok_per_s"] for k, v in extra.items() if k.startswith("decode_B")}
                for S2, B2 in ((4096, 8), (4096, 1), (16384, 1), (65536, 1)):
                    m2, Q2, Kc2, K2, V2 = make_inputs(nv, B2, S2, device, 99)
                    ms = time_events(lambda: hot_path(nv, m2, Q2, Kc2, K2, V2, S2), 5, warm=2)
                    sc, se, at, sa, Ls, Lm, nt2, scs = stage_times(nv, m2, Q2, Kc2, K2, V2, S2, 5)
                    extra[f"prefill_S{S2}_B{B2}"] = {"ms": ms, "tok_per_s": B2 * S2 / (ms * 1e-3), "scores_ms": sc, "select_ms": se, "scores_and_select_one_launch_ms": scs,
                                                    "attention_ms": at, "select_and_attention_ms": sa, "attn_alg_GBps": Ls * 256 / (at * 1e-3) / 1e9,
                                                    "attn_gathered_GBps": nt2 * BLK_KEYS * 256 / (at * 1e-3) / 1e9,
                                                    "attn_tflops": 4.0 * H * Ls * D / (at * 1e-3) / 1e12,
                                                    "attn_mfma_frac": 4.0 * H * Ls * D / (at * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                                                    "hbm_traffic": pmc_traffic(f"S{S2}_B{B2}"),
                                                    "hbm_traffic_frac_of_peak": (pmc_traffic(f"S{S2}_B{B2}") or 0.0) / (at * 1e-3) / 1e9 / HBM_PEAK_GBPS}
                    if S2 == 4096 and B2 == 8:
                        extra[f"sel_attn_fwd_bwd_S{S2}_B{B2}"] = backward_bench(nv, m2, Q2, K2, V2, S2)
                    del m2, Q2, Kc2, K2, V2
                extra["attention_extremes"] = attention_extremes(nv, device)
                # next scope rows: the sliding/compressed branch kernel (MFMA bound), the whole layer on the native path
                for S2, B2 in ((4096, 8), (16384, 1), (65536, 1)):
                    extra[f"band_S{S2}_B{B2}"] = band_bench(nv, B2, S2, device)
                bw = extra["band_S65536_B1"]["cmp"]
                out["roofline_mfma"] = {"kernel": "band_attn_fwd_kernel<bf16,3> (compressed branch, S=65536)", "bound": "mfma",
                                        "achieved": bw["tflops"], "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                        "frac": bw["tflops"] / MFMA_BF16_PEAK_TFLOPS, "kernel_ms": bw["ms"]}
                for S2, B2 in ((4096, 8), (16384, 1), (65536, 1)):
                    extra[f"layer_S{S2}_B{B2}"] = layer_bench(nv, B2, S2, device)
                extra["layer_train_S4096_B8"] = layer_train_bench(nv, 8, 4096, device)
                # the whole m7c_125m model (BASELINE configs 2-4 name it): attention layers native, norms / MLP / head PyTorch-ROCm
                for S2, B2 in ((4096, 1), (16384, 1), (4096, 32)):
                    extra[f"model_m7c_125m_S{S2}_B{B2}"] = model_bench(B2, S2, device)
            except Exception as e:  # noqa: BLE001 -- extras must not void the headline number
                extra["error"] = repr(e)
            out["extra"] = extra
            out["cpu_baseline"] = cpu_baseline(S, B)
        if refused is None:
            print(json.dumps(out))
        else:
            print(refused, file=sys.stderr)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if refused is not None:
        sys.exit(3)


if __name__ == "__main__":
    main()
