// Kernel argument blocks for the selection-attention kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nsa {

struct SelAttnParams {
    const void *Q;          // [R,h,Dk]   (R = B*S*G rows, row = (b*S+s)*G+g)
    const void *K;          // [B,G,S_kv,Dk] with element strides ksb/ksg/kss
    const void *V;          // [B,G,S_kv,Dv] with element strides vsb/vsg/vss
    const int32_t *ranges;  // [R,n,2]
    void *O;                // [R,h,Dv]
    float *lse;             // [R,h] or null
    int64_t R;
    int S, G, h, Dk, Dv, S_kv, n;
    int64_t ksb, ksg, kss, vsb, vsg, vss;
    float scale;
    // split-KV (few rows): partial results, see sel_attn_mfma.hip
    float *part;  // workspace or null
    int nsplit;
    int map_mode;  // 0 rows linear in blockIdx; 1 workgroup = 4 tokens of one (b,g); 2 = 1 + XCD-aware order
    int defer_combine;  // split-KV: leave the partial records for the caller's own combine pass
    int fuse_select;    // the kernel first selects the row's ranges from its group scores (select = const SelectParams *, host)
    const void *select;
    int tpw, nw, wave_lds;  // query-tile form (sel_attn_rows_mfma.hip): rows per wave, 32-tile bitmap words per row, LDS bytes per wave
    void *ks_ws;            // key-split form of the block kernel (sel_attn_blocks_mfma.hip): caller's workspace (16-byte aligned) or null
    size_t ks_bytes;
    // key-split form, filled by its launcher: rows [0, ks_r1) of a (b,g) are walked unsplit, [ks_r1, ks_r2) in two key classes, [ks_r2, S) in
    // four (multiples of the 32 rows of a workgroup); workgroups per XCD of one class of each zone
    int ks_r1, ks_r2, ks_w1, ks_w2, ks_wa, ks_wb, ks_wc;  // (ks_w1 / ks_w2: first workgroup of zones 1 / 2; ks_r = min(ks_w * rows per workgroup, S))
};

struct SelAttnBwdParams {
    const void *Q, *K, *V;
    const int32_t *ranges;
    const void *O;
    const float *lse;
    const void *dO;
    void *dQ;
    float *dK, *dV;
    int64_t R;
    int S, G, h, Dk, Dv, S_kv, n;
    int64_t ksb, ksg, kss, vsb, vsg, vss;
    float scale;
    int skip_delta_dq;  // MFMA route: delta is already in the workspace and dQ is produced elsewhere (band backward)
};

// Band attention (sliding-window and compressed branches): query row t (position t0 + t) attends keys [max(0, hi - w), hi),
//   hi(t) = (t0 + t + 1 >= a) ? min(S_kv, (t0 + t + 1 - a) / dd + c) : 0.
// sliding window: a = 0, dd = 1, c = 0 (hi = t + 1), w = window; compressed: a = l, dd = d, c = 1, w = "infinite".
struct BandAttnParams {
    const void *Q;  // [B,S,G,h,D]
    const void *K;  // [B,G,S_kv,D] with element strides ksb/ksg/kss
    const void *V;
    void *O;        // [B,S,G,h,D]
    float *lse;     // [B,S,G,h] or null
    int B, S, G, h, Dk, Dv, S_kv;
    int64_t ksb, ksg, kss, vsb, vsg, vss;
    float scale;
    int t0, a, dd, c, w;
    float *part;  // split-KV partial records (few rows) or null
    int nsplit;
    int map_mode;  // 0 linear, 1 workgroup = 4 token groups of one (b,g), 2 = 1 + XCD-aware order
    int tpw;       // tokens per wave
    int defer_combine;  // split-KV: leave the partial records for the caller's own combine pass
};
__host__ __device__ inline int band_hi(int t0, int a, int dd, int c, int S_kv, int t) {
    const int e = t0 + t + 1 - a;
    if (e < 0) return 0;
    const int hi = e / dd + c;
    return hi < S_kv ? hi : S_kv;
}
bool band_attn_mfma_supported(int dtype, int h, int Dk, int Dv);
size_t band_attn_workspace(int B, int S, int G, int h, int Dk, int Dv, int dtype, int *nsplit_out);
int launch_band_attn_fwd_mfma(const BandAttnParams &P, int dtype, hipStream_t st);
int launch_band_attn_fwd_dual(const BandAttnParams &P0, const BandAttnParams &P1, int dtype, hipStream_t st);
// the sliding (w) and compressed (c) branch of a decode step as extra workgroups of ANOTHER launch (the one-launch decode step of the
// selected branch): workgroups [n_sel, n_sel + n_w) take w, those behind them c; every wave of such a workgroup is one (row, split) unit
// mg.on: the splits of a (row, branch) unit are the consecutive waves of ONE workgroup (nsplit divides the workgroup's waves) and are merged
// through LDS by those waves -- the branch output O is final in the activation dtype (the arithmetic of the decode finish kernel's merge, bit
// for bit), no partial record goes to memory; the sliding branch's unit also evaluates the row's gate probabilities into gates [R,3]
struct BandMergeArgs {
    int on;
    int Hd;
    float tau;
    const void *gw1, *gb1, *gw2, *gb2;  // gate MLP (fc1 weight / bias, fc2 weight / bias)
    float *gates;
};
struct DecBandPair {
    BandAttnParams w, c;
    BandMergeArgs mg;
    unsigned n_sel;  // workgroups of the host kernel's own work (0xffffffff: the launch carries no band work)
    unsigned n_w;    // workgroups of the sliding branch
};
// fills tpw of both argument blocks and returns the (row, split) units = waves of each; false: not both in split form with deferred combine
bool band_dual_plan(BandAttnParams *P0, BandAttnParams *P1, int dtype, int64_t waves[2]);
int launch_band_attn_bwd_dq(const BandAttnParams &P, const void *dO, const float *lse, const float *delta, void *dQ, int dtype,
                            hipStream_t st);
int launch_bwd_delta(const void *O, const void *dO, float *delta, int64_t n_rows, int Dv, int dtype, hipStream_t st);
int launch_band_ranges(int32_t *ranges, int B, int S, int G, int S_kv, int t0, int a, int dd, int c, int w, hipStream_t st);
int launch_band_attn_fwd_generic(const BandAttnParams &P, int dtype, hipStream_t st);

int launch_sel_attn_fwd_generic(const SelAttnParams &P, int dtype, hipStream_t st);
int launch_sel_attn_bwd_generic(const SelAttnBwdParams &P, int dtype, hipStream_t st);
// decode form (S = 1): one 1024-thread workgroup per row, partials merged through LDS (sel_attn_decode.hpp)
bool sel_attn_decode_wg_supported(int dtype, int h, int Dk, int Dv, int n, int64_t ksb, int64_t ksg, int64_t kss, int64_t vsb, int64_t vsg,
                                  int64_t vss, const void *Q, const void *K, const void *V);
int launch_sel_attn_decode_wg(const void *Q, const void *K, const void *V, const int32_t *ranges, void *O, int64_t R, int G, int h, int S_kv, int n,
                              int64_t ksb, int64_t ksg, int64_t kss, int64_t vsb, int64_t vsg, int64_t vss, int dtype, float scale, hipStream_t st);
int launch_sel_head_causal(const SelAttnParams &P, int dtype, hipStream_t st);  // parity mode of _sdpa_over_ranges
// returns NSA_ERR_INVALID (without setting an error) when the shape is not covered
bool sel_attn_mfma_supported(int dtype, int h, int Dk, int Dv);
int launch_sel_attn_fwd_mfma(const SelAttnParams &P, int dtype, hipStream_t st);
size_t sel_attn_mfma_workspace(int64_t R, int h, int Dv, int *nsplit_out);
// query-tile form: rows per wave for this shape, 0 = not covered (use the one-row-per-wave kernel)
int sel_attn_rows_tpw(int dtype, int h, int Dk, int Dv, int S, int S_kv, int n, int64_t R, int *nt);
// key-split form of the block kernel: bytes of partial records it needs, 0 = not wanted for this shape (rule / tuning in sel_attn_blocks_mfma.hip)
size_t sel_attn_ksplit_workspace(int dtype, int h, int Dk, int Dv, int S, int S_kv, int n, int64_t R);
int launch_sel_attn_rows_mfma(const SelAttnParams &P, int dtype, int tpw, int nt, hipStream_t st);
// block form (64-key blocks, NT column tiles of 16/h rows per wave): column tiles per wave for this shape, 0 = not covered
int sel_attn_blocks_nt(int dtype, int h, int Dk, int Dv, int S, int S_kv, int n, int64_t R, int64_t kss, int64_t vss);
int launch_sel_attn_blocks_mfma(const SelAttnParams &P, int dtype, int nt, hipStream_t st);
bool sel_attn_bwd_mfma_supported(int dtype, int h, int Dk, int Dv);
size_t sel_attn_bwd_mfma_workspace(int64_t R, int h, int S, int64_t nbg, int S_kv);
int launch_sel_attn_bwd_mfma(const SelAttnBwdParams &P, int dtype, float *delta_ws, hipStream_t st);

}  // namespace nsa
