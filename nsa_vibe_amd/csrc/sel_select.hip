// Deterministic top-n + forced blocks + range merge: one wave64 per (b,t,g) row.
//
// Reference: select_topn_ranges (nsa/core/selection_scorer.py:124-249, decode / sequential prefill)
// and select_topn_ranges_batched + convert_indices_to_ranges_batched_v2 (:255-362, :434-605).
//
// Per row the selected set is kept as a bitmap over selection blocks (bit per block, lane l owns
// blocks l, l+64, l+128, ...), which makes the reference's "concat forced + top-k, sort, drop
// duplicates, merge adjacent" a run extraction on the bitmap:
//   * candidate key  = fp32(p) - fp32(idx) * 1e-8f, multiply and subtract rounded separately
//     (selection_scorer.py:182-184; __fmul_rn/__fsub_rn so hipcc cannot contract to an FMA);
//   * the k picks are a threshold (radix) select: 32 ballot+popcount rounds find the k-th largest key,
//     ties at the threshold go to the lowest indices = order (key desc, idx asc); -inf keys are never
//     picked (the reference then picks arbitrary -inf entries, which are either duplicates of forced
//     blocks or future blocks it drops / emits as garbage);
//   * runs of the bitmap are emitted in ascending order as [start*l', min((end+1)*l', t+1)).
#include "nsa_common.hpp"

#include "sel_select_row.hpp"

namespace nsa {

template <int CAND>
__global__ __launch_bounds__(256) void select_topn_kernel(SelectParams P) {
    __shared__ int scr[4][128];  // run extraction scratch, one slice per wave
    const int64_t row = (int64_t)blockIdx.x * 4 + uniform((int)(threadIdx.x >> 6));
    if (row >= P.R) return;
    // one row per wave: the row and its token are scalars (and 32-bit arithmetic when the row count allows: a 64-bit division is ~100 instructions)
    int t;
    if (P.t_rows) t = P.t_rows[row];
    else if (P.R < ((int64_t)1 << 31)) t = P.t0 + (int)(((unsigned)row / (unsigned)P.G) % (unsigned)P.S);
    else t = P.t0 + (int)((row / P.G) % P.S);
    t = uniform(t);
    select_topn_row_auto<CAND>(P, P.p_grp + row * (int64_t)P.S_sel, t, P.out + row * (int64_t)P.W * 2, scr[threadIdx.x >> 6]);
}

// ---- v2 converter alone: one thread per row -------------------------------------------------
struct ConvParams {
    const int32_t *idx;
    int32_t *out;
    int64_t R;
    int S, G, t0, K, S_sel, l_sel;
};

__global__ __launch_bounds__(256) void indices_to_ranges_kernel(ConvParams P) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= P.R) return;
    const int t = P.t0 + (int)((row / P.G) % P.S);
    const int32_t *x = P.idx + row * (int64_t)P.K;
    int32_t *out = P.out + row * (int64_t)P.K * 2;
    for (int i = 0; i < 2 * P.K; ++i) out[i] = 0;
    int m = 0, i = 0;
    while (i < P.K) {
        if (x[i] < 0) {
            ++i;
            continue;
        }
        const int first = x[i];
        int last = x[i];
        int j = i + 1;
        while (j < P.K && x[j] >= 0 && (x[j] - x[j - 1] == 0 || x[j] - x[j - 1] == 1)) {
            last = max(last, x[j]);
            ++j;
        }
        if (first < P.S_sel && last < P.S_sel) {
            out[2 * m] = first * P.l_sel;
            out[2 * m + 1] = min(last * P.l_sel + P.l_sel, t + 1);
        }
        ++m;
        i = j;
    }
}

// ---- host ---------------------------------------------------------------------------------
// forced-column rule of the batched selector (selection_scorer.py:283-300), closed form:
// sorted per-row forced list = [0 (init)] + [max(c - a, 0) for a = fl-1 .. 0], c = t // l';
// torch.unique_consecutive(dim=-1) drops column i iff it equals column i-1 for EVERY t in [0,S),
// i.e. iff c_max = (S-1)//l' <= a_i.
unsigned batched_keepmask(int S, int l_sel, int force_init, int force_local, int *n_kept) {
    const int nf = (force_init ? 1 : 0) + (force_local > 0 ? force_local : 0);
    const int cmax = S > 0 ? (S - 1) / l_sel : 0;
    unsigned mask = 0;
    int kept = 0;
    for (int i = 0; i < nf && i < 32; ++i) {
        bool keep;
        if (i == 0) keep = true;
        else {
            const int a = nf - 1 - i;  // this column is max(c - a, 0)
            keep = cmax > a;
        }
        if (keep) {
            mask |= 1u << i;
            ++kept;
        }
    }
    if (n_kept) *n_kept = kept;
    return mask;
}

int batched_width(int S, int S_sel, int l_sel, int n_top, int force_init, int force_local) {
    int nfc;
    batched_keepmask(S, l_sel, force_init, force_local, &nfc);
    const int k_rest = n_top - nfc > 0 ? n_top - nfc : 0;
    const int k_actual = k_rest < S_sel ? k_rest : S_sel;
    if (n_top >= S_sel) return S_sel;
    if (k_rest > 0) return nfc + k_actual;
    return nfc < n_top ? nfc : n_top;
}

int select_params_sequential(SelectParams *P, int S_sel, int l_sel, int n_top, int force_init, int force_local, int W) {
    NSA_CHECK_ARG(S_sel >= 1 && S_sel <= 64 * 64 && l_sel >= 1 && n_top >= 0 && force_local >= 0 && force_local <= 30 && W == n_top,
                  "select (sequential): bad sizes");
    P->S_sel = S_sel; P->l_sel = l_sel; P->n_top = n_top; P->force_init = force_init ? 1 : 0; P->force_local = force_local;
    P->mode = NSA_SEL_SEQUENTIAL; P->W = W;
    P->l_sel_shift = (l_sel & (l_sel - 1)) == 0 ? __builtin_ctz((unsigned)l_sel) : -1;
    const int nf_all = P->force_init + force_local;
    const int k_rest = n_top - nf_all > 0 ? n_top - nf_all : 0;
    P->k_actual = k_rest < S_sel ? k_rest : S_sel;
    P->n_forced = nf_all;
    P->keepmask = 0xffffffffu;
    P->all_valid = 0;
    P->forced_plain = 1;
    return NSA_OK;
}

int select_params_fill(SelectParams *Pp, int S_sel, int l_sel, int n_top, int force_init, int force_local, int mode, int S_total, int W) {
    SelectParams &P = *Pp;
    NSA_CHECK_ARG(S_sel >= 1 && S_sel <= 64 * 64 && l_sel >= 1 && n_top >= 0 && force_local >= 0 && force_local <= 30, "select: bad sizes");
    P.S_sel = S_sel; P.l_sel = l_sel; P.n_top = n_top; P.force_init = force_init ? 1 : 0; P.force_local = force_local; P.mode = mode; P.W = W;
    P.l_sel_shift = (l_sel & (l_sel - 1)) == 0 ? __builtin_ctz((unsigned)l_sel) : -1;
    const int nf_all = P.force_init + force_local;
    if (mode == NSA_SEL_SEQUENTIAL) {
        NSA_CHECK_ARG(W == n_top, "select (sequential): out_width must be n_top");
        const int k_rest = n_top - nf_all > 0 ? n_top - nf_all : 0;
        P.k_actual = k_rest < S_sel ? k_rest : S_sel;
        P.n_forced = nf_all;
        P.keepmask = 0xffffffffu;
        P.all_valid = 0;
        P.forced_plain = 1;
    } else if (mode == NSA_SEL_BATCHED) {
        int nfc;
        P.keepmask = batched_keepmask(S_total, l_sel, force_init, force_local, &nfc);
        NSA_CHECK_ARG(W == batched_width(S_total, S_sel, l_sel, n_top, force_init, force_local),
                      "select (batched): out_width %d != nsa_batched_ranges_width()", W);
        const int k_rest = n_top - nfc > 0 ? n_top - nfc : 0;
        P.k_actual = k_rest < S_sel ? k_rest : S_sel;
        P.n_forced = k_rest > 0 ? nfc : (nfc < n_top ? nfc : n_top);
        P.all_valid = n_top >= S_sel ? 1 : 0;
        P.forced_plain = (nfc == nf_all && P.n_forced == nf_all) ? 1 : 0;
    } else {
        NSA_CHECK_ARG(false, "select: unknown mode %d", mode);
    }
    return NSA_OK;
}

int launch_select_topn(const float *p_grp, int64_t R, int S, int G, int t0, const int32_t *t_rows, int S_sel,
                       int l_sel, int n_top, int force_init, int force_local, int mode, int S_total,
                       int32_t *out, int W, hipStream_t st) {
    NSA_CHECK_ARG(R >= 0 && S >= 1 && G >= 1 && S_sel >= 1 && l_sel >= 1 && n_top >= 0, "select: bad sizes");
    NSA_CHECK_ARG(S_sel <= 64 * 64, "select: S_sel=%d exceeds 4096 selection blocks", S_sel);
    NSA_CHECK_ARG(force_local >= 0 && force_local <= 30, "select: force_local out of range");
    SelectParams P{};
    P.p_grp = p_grp; P.t_rows = t_rows; P.out = out; P.R = R; P.S = S; P.G = G; P.t0 = t0; P.S_sel = S_sel;
    P.l_sel = l_sel; P.n_top = n_top; P.force_init = force_init ? 1 : 0; P.force_local = force_local; P.mode = mode;
    P.W = W;
    if (int rc = select_params_fill(&P, S_sel, l_sel, n_top, force_init, force_local, mode, S_total, W)) return rc;
    if (R == 0 || W == 0) return NSA_OK;
    const unsigned grid = (unsigned)((R + 3) / 4);
    const int cand = (S_sel + 63) / 64;
#define NSA_SEL_LAUNCH(C) hipLaunchKernelGGL(select_topn_kernel<C>, dim3(grid), dim3(256), 0, st, P)
    if (cand <= 1) NSA_SEL_LAUNCH(1);
    else if (cand <= 2) NSA_SEL_LAUNCH(2);
    else if (cand <= 4) NSA_SEL_LAUNCH(4);
    else if (cand <= 8) NSA_SEL_LAUNCH(8);
    else if (cand <= 16) NSA_SEL_LAUNCH(16);
    else if (cand <= 32) NSA_SEL_LAUNCH(32);
    else NSA_SEL_LAUNCH(64);
#undef NSA_SEL_LAUNCH
    NSA_LAUNCH_CHECK("select_topn");
    return NSA_OK;
}

int launch_indices_to_ranges(const int32_t *idx, int64_t R, int S, int G, int t0, int K, int S_sel, int l_sel,
                             int32_t *out, hipStream_t st) {
    if (R == 0 || K == 0) return NSA_OK;
    ConvParams P{idx, out, R, S, G, t0, K, S_sel, l_sel};
    hipLaunchKernelGGL(indices_to_ranges_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, P);
    NSA_LAUNCH_CHECK("indices_to_ranges");
    return NSA_OK;
}

}  // namespace nsa
