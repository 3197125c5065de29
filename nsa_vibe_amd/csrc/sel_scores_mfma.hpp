// Parameters of the fused MFMA selection scorers (sel_scores_mfma.hip: 16x16x32 tiles, every group size; sel_scores_mfma32.hip: 32x32x16
// tiles, h = 6 and D = 64).
#pragma once
#include "nsa_common.hpp"

namespace nsa {

struct ScoresMfmaParams {
    const void *Q;   // [B,S,G,h,D]
    const void *Kc;  // [B,G,S_cmp,D] strided
    float *p_grp;    // [B,S,G,S_sel]
    int B, S, G, h, S_cmp, S_sel;
    int64_t csb, csg, css;
    float scale;
    int causal_skip;
    int d_stride;  // the compression stride d (tokens); l' = 4d
    int big_out;   // S G S_sel >= 2^31 elements per sequence: 64-bit output offsets
};

// h = 6, D = 64, 32-bit output offsets: the 32x32x16 form (sel_scores_mfma32.hip)
bool scores_mfma32_supported(const ScoresMfmaParams &P, int Dk);
// sel != null: the workgroup also selects the ranges of its 64 query rows, right behind its second sweep (the scores it reads back are
// the ones it has just written: L2 hits), so the step needs no select launch; needs S_sel <= 1024 and at most 64 ranges per row
struct SelectParams;
bool scores_mfma32_select_supported(const ScoresMfmaParams &P, int Dk, const SelectParams &SP);
int launch_scores_mfma32(const ScoresMfmaParams &P, int dtype, hipStream_t st, const SelectParams *sel = nullptr);

}  // namespace nsa
