/*
 * nsa_oracle.c -- CPU restatement of nsa-vibe's selected-branch attention hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under nsa_vibe_amd/ (the product) may import,
 * link or call this file.  Allowed users: tests/, __graft_entry__.smoke(), and the
 * `cpu_baseline` leg of bench.py -- and there only as the checker / reported CPU
 * baseline, never as the thing shipped.
 *
 * Parity status: PINNED.  oracle/make_goldens.py imports the reference
 * (/root/reference, Python) in the build container, runs the reference functions
 * cited below on seeded inputs and stores inputs+outputs under tests/golden/;
 * tests/test_oracle_golden.py checks this file against those vectors
 * (bit-exact for integer ranges and for the Eq.9/Eq.10 fp32 chain, 1e-6 for the
 * softmax scores, 1e-5 for attention outputs).
 *
 * All arithmetic is IEEE fp32 with separate multiply/add (compile with
 * -ffp-contract=off); every function cites the reference file:line it follows.
 * Paths are relative to the reference repo root.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NSA_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* A1: block starts + fractional-overlap CSR  (nsa/core/block_index.py:25-71) */
/* ------------------------------------------------------------------------- */

/* nsa/core/block_index.py:25-36 */
NSA_API int nsa_oracle_block_counts(int seq_len, int l, int d, int l_sel, int *S_cmp, int *S_sel) {
    if (d <= 0 || l <= 0 || l_sel <= 0) return -1;
    *S_cmp = (seq_len < l) ? 0 : (seq_len - l) / d + 1;
    *S_sel = (seq_len <= 0) ? 0 : (seq_len + l_sel - 1) / l_sel;
    return 0;
}

static int overlap_len(int a0, int a1, int b0, int b1) { /* block_index.py:39-40 */
    int lo = a0 > b0 ? a0 : b0;
    int hi = a1 < b1 ? a1 : b1;
    return hi > lo ? hi - lo : 0;
}

/*
 * nsa/core/block_index.py:43-71 (build_M_csl_csr).  indptr has S_cmp+1 entries,
 * indices/values must hold at least nsa_oracle_csr_nnz() entries.  Weights are
 * ov/total computed in double and rounded once to fp32, as torch.tensor(list of
 * python floats, dtype=float32) does.  Returns nnz.
 */
NSA_API int nsa_oracle_build_csr(int seq_len, int l, int d, int l_sel, int32_t *indptr,
                                 int32_t *indices, float *values) {
    int S_cmp, S_sel;
    if (nsa_oracle_block_counts(seq_len, l, d, l_sel, &S_cmp, &S_sel)) return -1;
    int nnz = 0;
    indptr[0] = 0;
    for (int i = 0; i < S_cmp; ++i) {
        int a0 = i * d, a1 = i * d + l;
        int total = 0;
        /* only selection blocks that can overlap [a0,a1) */
        int j0 = a0 / l_sel, j1 = (a1 - 1) / l_sel;
        if (j1 >= S_sel) j1 = S_sel - 1;
        for (int j = j0; j <= j1; ++j) total += overlap_len(a0, a1, j * l_sel, j * l_sel + l_sel);
        if (total > 0) {
            for (int j = j0; j <= j1; ++j) {
                int ov = overlap_len(a0, a1, j * l_sel, j * l_sel + l_sel);
                if (ov > 0) {
                    if (indices) {
                        indices[nnz] = j;
                        values[nnz] = (float)((double)ov / (double)total);
                    }
                    ++nnz;
                }
            }
        }
        indptr[i + 1] = nnz;
    }
    return nnz;
}

/* ------------------------------------------------------------------------- */
/* A2: p_cmp = softmax_c(Q K_cmp^T * scale)  (nsa/core/selection_scorer.py:42-61) */
/*   Q [B,S,G,h,Dk] contiguous; K_cmp [B,G,S_cmp,Dk]; out [B,S,G,h,S_cmp].      */
/*   Softmax over ALL S_cmp columns (no causal mask), as the reference does.     */
/* ------------------------------------------------------------------------- */
NSA_API int nsa_oracle_pcmp_all(const float *Q, const float *Kc, float *P, int B, int S, int G,
                                int h, int Dk, int S_cmp, float scale) {
    if (S_cmp == 0) return 0;
    long rows = (long)B * S * G * h;
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        long hh = r % h;
        long g = (r / h) % G;
        long s = (r / ((long)h * G)) % S;
        long b = r / ((long)h * G * S);
        (void)hh;
        (void)s;
        const float *q = Q + r * Dk;
        const float *k = Kc + ((b * G + g) * (long)S_cmp) * Dk;
        float *p = P + r * (long)S_cmp;
        float m = -INFINITY;
        for (int c = 0; c < S_cmp; ++c) {
            float acc = 0.f;
            for (int e = 0; e < Dk; ++e) acc += q[e] * k[(long)c * Dk + e];
            acc = acc * scale;
            p[c] = acc;
            if (acc > m) m = acc;
        }
        float sum = 0.f;
        for (int c = 0; c < S_cmp; ++c) {
            p[c] = expf(p[c] - m);
            sum += p[c];
        }
        for (int c = 0; c < S_cmp; ++c) p[c] = p[c] / sum;
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* A3+A4: Eq.9 COO scatter-add + Eq.10 head sum                               */
/*   (nsa/core/selection_scorer.py:89-116; nsa/core/nsa_attention.py:670,1091) */
/*   p_cmp [R,h,S_cmp_cur] -> p_slc [R,h,S_sel] (optional) -> p_grp [R,S_sel]   */
/*   COO entries in CSR order (ascending cmp row); rows >= S_cmp_cur dropped    */
/*   (selection_scorer.py:103-108).  Accumulation order = the CPU scatter_add   */
/*   order: ascending nnz; product rounded before the add; head sum ascending h.*/
/* ------------------------------------------------------------------------- */
NSA_API int nsa_oracle_map_pcmp_to_pgrp(const float *p_cmp, long R, int h, int S_cmp_cur,
                                        const int32_t *indptr, const int32_t *indices,
                                        const float *values, int S_cmp_meta, int S_sel,
                                        float *p_slc /* nullable */, float *p_grp) {
    int rmax = S_cmp_cur < S_cmp_meta ? S_cmp_cur : S_cmp_meta;
#pragma omp parallel
    {
        float *tmp = (float *)malloc(sizeof(float) * (size_t)S_sel);
#pragma omp for schedule(static)
        for (long r = 0; r < R; ++r) {
            float *grp = p_grp + r * (long)S_sel;
            for (int j = 0; j < S_sel; ++j) grp[j] = 0.f;
            for (int hh = 0; hh < h; ++hh) {
                const float *pc = p_cmp + (r * h + hh) * (long)S_cmp_cur;
                float *dst = p_slc ? p_slc + (r * h + hh) * (long)S_sel : tmp;
                for (int j = 0; j < S_sel; ++j) dst[j] = 0.f;
                for (int i = 0; i < rmax; ++i) {
                    for (int k = indptr[i]; k < indptr[i + 1]; ++k) {
                        volatile float prod = pc[i] * values[k];
                        dst[indices[k]] = dst[indices[k]] + prod;
                    }
                }
                for (int j = 0; j < S_sel; ++j) grp[j] = grp[j] + dst[j];
            }
        }
        free(tmp);
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* top-k helper: composite key (selection_scorer.py:182-187, 312-321)          */
/*   composite = fp32(masked) - fp32(idx) * 1e-8f (unfused);                    */
/*   order = composite descending, index ascending on exact ties (PRD.md:46,    */
/*   the documented intent; torch.topk leaves exact ties unspecified).          */
/* ------------------------------------------------------------------------- */
static inline float composite_key(float masked, int idx) {
    volatile float bias = (float)idx * 1e-8f;
    volatile float c = masked - bias;
    return c;
}

/* pick k best of n by (key desc, idx asc); -inf keys are never picked.  out_idx
 * receives the picks in rank order; returns how many were picked (<= k).      */
static int topk_pick(const float *key, int n, int k, int *out_idx) {
    int got = 0;
    char *used = (char *)calloc((size_t)n, 1);
    for (int r = 0; r < k; ++r) {
        int best = -1;
        float bk = -INFINITY;
        for (int i = 0; i < n; ++i) {
            if (used[i]) continue;
            if (key[i] > bk) {
                bk = key[i];
                best = i;
            }
        }
        if (best < 0) break; /* only -inf left */
        used[best] = 1;
        out_idx[got++] = best;
    }
    free(used);
    return got;
}

static int cmp_int(const void *a, const void *b) {
    int x = *(const int *)a, y = *(const int *)b;
    return (x > y) - (x < y);
}

/* ------------------------------------------------------------------------- */
/* A5: select_topn_ranges -- decode / sequential mode                          */
/*   (nsa/core/selection_scorer.py:124-249)                                     */
/*   p_grp [R,S_sel] (R = B*G rows at one token position t, or any list of rows */
/*   with per-row t in t_tokens).  Output [R,n_top,2] int32, zero padded.       */
/*   Normalisation (SURVEY 7 hard part (c)): when fewer than k_rest valid       */
/*   non-forced candidates exist the reference's topk picks -inf entries in an  */
/*   unspecified order and emits inverted ranges (start > end) for them; this   */
/*   restatement emits nothing for such picks.  Compare on {(s,e): e > s}.      */
/* ------------------------------------------------------------------------- */
NSA_API int nsa_oracle_select_topn_seq(const float *p_grp, long R, int S_sel, int l_sel,
                                       int n_top, const int32_t *t_tokens, int force_init,
                                       int force_local, int32_t *ranges) {
#pragma omp parallel
    {
        float *key = (float *)malloc(sizeof(float) * (size_t)(S_sel > 0 ? S_sel : 1));
        int *sel = (int *)malloc(sizeof(int) * (size_t)(n_top + force_local + 2 + S_sel));
#pragma omp for schedule(static)
        for (long r = 0; r < R; ++r) {
            int t = t_tokens[r];
            int32_t *out = ranges + r * (long)n_top * 2;
            memset(out, 0, sizeof(int32_t) * (size_t)n_top * 2);
            const float *p = p_grp + r * (long)S_sel;
            /* :156 valid_j <=> start_j + l_sel - 1 <= t */
            for (int j = 0; j < S_sel; ++j) {
                int valid = (j * l_sel + l_sel - 1) <= t;
                key[j] = valid ? composite_key(p[j], j) : -INFINITY;
            }
            /* :159-170 forced list, NOT deduplicated */
            int nf = 0;
            if (force_init) sel[nf++] = 0;
            int last_block = t / l_sel;
            if (last_block < 0) last_block = 0;
            for (int i = 0; i < force_local; ++i) {
                int f = last_block - i;
                if (f < 0) f = 0;
                sel[nf++] = f;
            }
            /* :172-175 forced -> -inf (scatter_ requires the index in range) */
            for (int i = 0; i < nf; ++i)
                if (sel[i] < S_sel) key[sel[i]] = -INFINITY;
            int k_rest = n_top - nf;
            if (k_rest < 0) k_rest = 0;
            int ns = nf;
            if (k_rest > 0) {
                int k_actual = k_rest < S_sel ? k_rest : S_sel; /* :186 */
                ns += topk_pick(key, S_sel, k_actual, sel + nf);
            }
            qsort(sel, (size_t)ns, sizeof(int), cmp_int); /* :211 */
            /* :221-248 unique_consecutive on block starts, merge adjacent, clamp */
            int m = 0;
            int have = 0, cur_s = 0, cur_e = 0, prev = -1;
            for (int i = 0; i < ns; ++i) {
                int blk = sel[i];
                if (have && blk == prev) continue;
                prev = blk;
                int x = blk * l_sel;
                if (!have) {
                    cur_s = x;
                    cur_e = x + l_sel;
                    have = 1;
                } else if (x == cur_e) {
                    cur_e += l_sel;
                } else {
                    if (m < n_top) {
                        int e = cur_e < t + 1 ? cur_e : t + 1;
                        out[2 * m] = cur_s;
                        out[2 * m + 1] = e;
                    }
                    ++m;
                    cur_s = x;
                    cur_e = x + l_sel;
                }
            }
            if (have && m < n_top) {
                int e = cur_e < t + 1 ? cur_e : t + 1;
                out[2 * m] = cur_s;
                out[2 * m + 1] = e;
            }
        }
        free(key);
        free(sel);
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* A6: number of forced columns the batched selector keeps                     */
/*   (selection_scorer.py:283-300): forced = sort([0, c, c-1,...]) per row then  */
/*   unique_consecutive(dim=-1) over WHOLE columns [B,S,G]: a column is dropped  */
/*   only if it equals the previous column for every row t in [0,S).            */
/*   Writes the kept column ids (into the sorted per-row forced list) to keep[]. */
/* ------------------------------------------------------------------------- */
static void forced_row_sorted(int t, int l_sel, int force_init, int force_local, int *f, int *nf) {
    int n = 0;
    if (force_init) f[n++] = 0;
    int last_block = t / l_sel;
    for (int k = 0; k < force_local; ++k) {
        int v = last_block - k;
        if (v < 0) v = 0;
        f[n++] = v;
    }
    qsort(f, (size_t)n, sizeof(int), cmp_int);
    *nf = n;
}

NSA_API int nsa_oracle_batched_forced_columns(int S, int l_sel, int force_init, int force_local,
                                              int *keep /* size force_init+force_local */) {
    int nfmax = (force_init ? 1 : 0) + (force_local > 0 ? force_local : 0);
    if (nfmax == 0) return 0;
    int nk = 0;
    int *fa = (int *)malloc(sizeof(int) * (size_t)nfmax);
    for (int c = 0; c < nfmax; ++c) {
        if (c == 0) {
            keep[nk++] = 0;
            continue;
        }
        int same = 1;
        for (int t = 0; t < S && same; ++t) {
            int nf;
            forced_row_sorted(t, l_sel, force_init, force_local, fa, &nf);
            if (fa[c] != fa[c - 1]) same = 0;
        }
        if (!same) keep[nk++] = c;
    }
    free(fa);
    return nk;
}

/* ------------------------------------------------------------------------- */
/* A6+A7: select_topn_ranges_batched + convert_indices_to_ranges_batched_v2     */
/*   (nsa/core/selection_scorer.py:255-362, 434-605)                            */
/*   p_grp_all [B,S,G,S_sel] -> ranges [B,S,G,K,2]; K returned via *K_out.      */
/*   If ranges == NULL only K is computed.                                      */
/* ------------------------------------------------------------------------- */
NSA_API int nsa_oracle_select_topn_batched(const float *p_grp_all, int B, int S, int G, int S_sel,
                                           int l_sel, int n_top, int force_init, int force_local,
                                           int32_t *ranges, int *K_out) {
    int nfmax = (force_init ? 1 : 0) + (force_local > 0 ? force_local : 0);
    int *keep = (int *)malloc(sizeof(int) * (size_t)(nfmax + 1));
    int nfc = nsa_oracle_batched_forced_columns(S, l_sel, force_init, force_local, keep);
    int k_rest = n_top - nfc;
    if (k_rest < 0) k_rest = 0;
    int k_actual = k_rest < S_sel ? k_rest : S_sel;
    int K;
    if (n_top >= S_sel) K = S_sel;                 /* :353-354 replaces selected by all_idx */
    else if (k_rest > 0) K = nfc + k_actual;       /* :339 */
    else K = nfc < n_top ? nfc : n_top;            /* :341 */
    *K_out = K;
    if (!ranges) {
        free(keep);
        return 0;
    }
    long R = (long)B * S * G;
#pragma omp parallel
    {
        float *key = (float *)malloc(sizeof(float) * (size_t)(S_sel > 0 ? S_sel : 1));
        int *sel = (int *)malloc(sizeof(int) * (size_t)(K + nfmax + 2));
        int *fa = (int *)malloc(sizeof(int) * (size_t)(nfmax + 1));
#pragma omp for schedule(static)
        for (long r = 0; r < R; ++r) {
            int t = (int)((r / G) % S);
            const float *p = p_grp_all + r * (long)S_sel;
            int32_t *out = ranges + r * (long)K * 2;
            memset(out, 0, sizeof(int32_t) * (size_t)K * 2);
            int ns = 0;
            if (n_top >= S_sel) {
                /* :349-354 all valid blocks: prefix of length num_valid(t) */
                for (int j = 0; j < S_sel; ++j)
                    if ((j + 1) * l_sel <= t + 1) sel[ns++] = j;
            } else {
                int nf;
                forced_row_sorted(t, l_sel, force_init, force_local, fa, &nf);
                /* :276-280 valid <=> block end <= t+1 */
                for (int j = 0; j < S_sel; ++j) {
                    int valid = (j + 1) * l_sel <= t + 1;
                    key[j] = valid ? composite_key(p[j], j) : -INFINITY;
                }
                int nfk = 0;
                for (int c = 0; c < nfc; ++c) {
                    int f = fa[keep[c]];
                    if (f < S_sel) key[f] = -INFINITY; /* :302-305 */
                    sel[nfk++] = f;
                }
                ns = nfk;
                if (k_rest > 0) {
                    ns += topk_pick(key, S_sel, k_actual, sel + nfk);
                } else if (ns > n_top) {
                    ns = n_top; /* :341 */
                }
                /* :344-347 keep only valid picks (forced current partial block dropped) */
                int w = 0;
                for (int i = 0; i < ns; ++i)
                    if (sel[i] >= 0 && sel[i] < S_sel && (sel[i] + 1) * l_sel <= t + 1)
                        sel[w++] = sel[i];
                ns = w;
            }
            qsort(sel, (size_t)ns, sizeof(int), cmp_int); /* :355 */
            /* v2 converter :466-474 runs where diff in {0,1}; :534-539 end = max id;
             * :569-573 clamp end to t+1; packed to the front */
            int m = 0;
            int i = 0;
            while (i < ns) {
                int first = sel[i], last = sel[i];
                int j = i + 1;
                while (j < ns && (sel[j] - sel[j - 1] == 0 || sel[j] - sel[j - 1] == 1)) {
                    last = sel[j];
                    ++j;
                }
                int s0 = first * l_sel;
                int e0 = last * l_sel + l_sel;
                if (e0 > t + 1) e0 = t + 1;
                if (m < K) {
                    out[2 * m] = s0;
                    out[2 * m + 1] = e0;
                }
                ++m;
                i = j;
            }
        }
        free(key);
        free(sel);
        free(fa);
    }
    free(keep);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* A8: grouped_selection_attention_masked -- the semantic oracle               */
/*   (nsa/core/attention_kernels.py:705-772)                                    */
/*   allowed = union of clamp([s,e), 0, S_kv); softmax over allowed keys with   */
/*   scale Dk^-1/2; rows without any allowed key -> zeros (:734-749, 769-771).  */
/*   Q [B,S,G,h,Dk], K [B,G,S_kv,Dk], V [B,G,S_kv,Dv], ranges [B,S,G,n,2].      */
/*   O [B,S,G,h,Dv]; lse (nullable) [B,S,G,h] = log sum exp of scaled logits.   */
/* ------------------------------------------------------------------------- */
NSA_API int nsa_oracle_sel_attention_masked(const float *Q, const float *K, const float *V,
                                            const int32_t *ranges, float *O, float *lse, int B,
                                            int S, int G, int h, int Dk, int Dv, int S_kv, int n,
                                            float scale) {
    long R = (long)B * S * G;
#pragma omp parallel
    {
        int *diff = (int *)malloc(sizeof(int) * (size_t)(S_kv + 2));
        int *idx = (int *)malloc(sizeof(int) * (size_t)(S_kv + 1));
        float *sc = (float *)malloc(sizeof(float) * (size_t)(S_kv + 1));
#pragma omp for schedule(dynamic, 4)
        for (long r = 0; r < R; ++r) {
            long g = r % G;
            long b = r / ((long)G * S);
            const int32_t *rg = ranges + r * (long)n * 2;
            /* difference array (:721-732) restricted to the touched span */
            int lo = S_kv, hi = 0;
            for (int i = 0; i < n; ++i) {
                long s0 = rg[2 * i], e0 = rg[2 * i + 1];
                if (s0 < 0) s0 = 0;
                if (s0 > S_kv) s0 = S_kv;
                if (e0 < 0) e0 = 0;
                if (e0 > S_kv) e0 = S_kv;
                if (e0 > s0) {
                    if (s0 < lo) lo = (int)s0;
                    if (e0 > hi) hi = (int)e0;
                }
            }
            int L = 0;
            if (hi > lo) {
                memset(diff + lo, 0, sizeof(int) * (size_t)(hi - lo + 1));
                for (int i = 0; i < n; ++i) {
                    long s0 = rg[2 * i], e0 = rg[2 * i + 1];
                    if (s0 < 0) s0 = 0;
                    if (s0 > S_kv) s0 = S_kv;
                    if (e0 < 0) e0 = 0;
                    if (e0 > S_kv) e0 = S_kv;
                    if (e0 > s0) {
                        diff[s0] += 1;
                        diff[e0] -= 1;
                    }
                }
                int run = 0;
                for (int c = lo; c < hi; ++c) {
                    run += diff[c];
                    if (run > 0) idx[L++] = c;
                }
            }
            const float *Kb = K + ((b * G + g) * (long)S_kv) * Dk;
            const float *Vb = V + ((b * G + g) * (long)S_kv) * Dv;
            for (int hh = 0; hh < h; ++hh) {
                const float *q = Q + (r * h + hh) * (long)Dk;
                float *o = O + (r * h + hh) * (long)Dv;
                for (int e = 0; e < Dv; ++e) o[e] = 0.f;
                if (L == 0) {
                    if (lse) lse[r * h + hh] = -INFINITY;
                    continue;
                }
                float m = -INFINITY;
                for (int i = 0; i < L; ++i) {
                    const float *k = Kb + (long)idx[i] * Dk;
                    float acc = 0.f;
                    for (int e = 0; e < Dk; ++e) acc += q[e] * k[e];
                    acc *= scale;
                    sc[i] = acc;
                    if (acc > m) m = acc;
                }
                double sum = 0.0;
                for (int i = 0; i < L; ++i) {
                    sc[i] = expf(sc[i] - m);
                    sum += (double)sc[i];
                }
                float inv = (float)(1.0 / sum);
                for (int i = 0; i < L; ++i) {
                    float pw = sc[i] * inv;
                    const float *v = Vb + (long)idx[i] * Dv;
                    for (int e = 0; e < Dv; ++e) o[e] += pw * v[e];
                }
                if (lse) lse[r * h + hh] = m + (float)log(sum);
            }
        }
        free(diff);
        free(idx);
        free(sc);
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* A8 backward (autograd of the masked SDPA, used to check the HIP backward).   */
/*   dO [B,S,G,h,Dv] -> dQ [B,S,G,h,Dk], dK [B,G,S_kv,Dk], dV [B,G,S_kv,Dv]      */
/*   Serial over rows (accumulates into dK/dV); small test sizes only.           */
/* ------------------------------------------------------------------------- */
NSA_API int nsa_oracle_sel_attention_masked_bwd(const float *Q, const float *K, const float *V,
                                                const int32_t *ranges, const float *dO, float *dQ,
                                                float *dK, float *dV, int B, int S, int G, int h,
                                                int Dk, int Dv, int S_kv, int n, float scale) {
    long R = (long)B * S * G;
    memset(dQ, 0, sizeof(float) * (size_t)(R * h * Dk));
    memset(dK, 0, sizeof(float) * (size_t)((long)B * G * S_kv * Dk));
    memset(dV, 0, sizeof(float) * (size_t)((long)B * G * S_kv * Dv));
    char *allowed = (char *)malloc((size_t)S_kv + 1);
    double *p = (double *)malloc(sizeof(double) * (size_t)(S_kv + 1));
    double *dp = (double *)malloc(sizeof(double) * (size_t)(S_kv + 1));
    for (long r = 0; r < R; ++r) {
        long g = r % G;
        long b = r / ((long)G * S);
        const int32_t *rg = ranges + r * (long)n * 2;
        memset(allowed, 0, (size_t)S_kv + 1);
        int any = 0;
        for (int i = 0; i < n; ++i) {
            long s0 = rg[2 * i], e0 = rg[2 * i + 1];
            if (s0 < 0) s0 = 0;
            if (s0 > S_kv) s0 = S_kv;
            if (e0 < 0) e0 = 0;
            if (e0 > S_kv) e0 = S_kv;
            for (long c = s0; c < e0; ++c) {
                allowed[c] = 1;
                any = 1;
            }
        }
        if (!any) continue;
        const float *Kb = K + ((b * G + g) * (long)S_kv) * Dk;
        const float *Vb = V + ((b * G + g) * (long)S_kv) * Dv;
        float *dKb = dK + ((b * G + g) * (long)S_kv) * Dk;
        float *dVb = dV + ((b * G + g) * (long)S_kv) * Dv;
        for (int hh = 0; hh < h; ++hh) {
            const float *q = Q + (r * h + hh) * (long)Dk;
            const float *go = dO + (r * h + hh) * (long)Dv;
            float *gq = dQ + (r * h + hh) * (long)Dk;
            double m = -INFINITY, sum = 0.0;
            for (int c = 0; c < S_kv; ++c) {
                if (!allowed[c]) continue;
                double acc = 0.0;
                for (int e = 0; e < Dk; ++e) acc += (double)q[e] * Kb[(long)c * Dk + e];
                p[c] = acc * scale;
                if (p[c] > m) m = p[c];
            }
            for (int c = 0; c < S_kv; ++c)
                if (allowed[c]) {
                    p[c] = exp(p[c] - m);
                    sum += p[c];
                }
            double delta = 0.0;
            for (int c = 0; c < S_kv; ++c)
                if (allowed[c]) {
                    p[c] /= sum;
                    double a = 0.0;
                    for (int e = 0; e < Dv; ++e) a += (double)go[e] * Vb[(long)c * Dv + e];
                    dp[c] = a;
                    delta += p[c] * a;
                }
            for (int c = 0; c < S_kv; ++c)
                if (allowed[c]) {
                    double ds = p[c] * (dp[c] - delta) * scale;
                    for (int e = 0; e < Dk; ++e) {
                        gq[e] += (float)(ds * Kb[(long)c * Dk + e]);
                        dKb[(long)c * Dk + e] += (float)(ds * q[e]);
                    }
                    for (int e = 0; e < Dv; ++e) dVb[(long)c * Dv + e] += (float)(p[c] * go[e]);
                }
        }
    }
    free(allowed);
    free(p);
    free(dp);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* A7 alone: convert_indices_to_ranges_batched_v2                              */
/*   (nsa/core/selection_scorer.py:434-605).  indices [R,K] sorted ascending,   */
/*   -1 padded; per-row clamp t_rows[r]+1; out [R,K,2] zero padded.  Ids outside */
/*   [0,S_sel) make the run invalid -> [0,0] (:554-565).                         */
/* ------------------------------------------------------------------------- */
NSA_API int nsa_oracle_indices_to_ranges_v2(const int32_t *indices, long R, int K, int S_sel,
                                            int l_sel, const int32_t *t_rows, int32_t *ranges) {
#pragma omp parallel for schedule(static)
    for (long r = 0; r < R; ++r) {
        const int32_t *x = indices + r * (long)K;
        int32_t *out = ranges + r * (long)K * 2;
        memset(out, 0, sizeof(int32_t) * (size_t)K * 2);
        int t = t_rows[r];
        int m = 0;
        int i = 0;
        while (i < K) {
            if (x[i] < 0) {
                ++i;
                continue;
            }
            int first = x[i], last = x[i];
            int j = i + 1;
            /* :470-474 a run continues while the previous element is valid and diff in {0,1} */
            while (j < K && x[j] >= 0 && (x[j] - x[j - 1] == 0 || x[j] - x[j - 1] == 1)) {
                if (x[j] > last) last = x[j];
                ++j;
            }
            if (first < S_sel && last >= 0 && last < S_sel) {
                int s0 = first * l_sel;
                int e0 = last * l_sel + l_sel;
                if (e0 > t + 1) e0 = t + 1;
                out[2 * m] = s0;
                out[2 * m + 1] = e0;
            }
            ++m;
            i = j;
        }
    }
    return 0;
}

NSA_API int nsa_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

NSA_API void nsa_oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
