// W_hh of one GRU direction as the resident MFMA fragments of the two cluster recurrences (f16x2 planes, f16_split.h), written once
// per weights version (inference) or once per step (training: train_prep_kernel), so that a recurrence's prologue is coalesced
// 16-byte loads instead of row-strided fp32 loads + the split -- 128 workgroups gathering 4-byte elements of a cold W_hh cost the
// matrix-core BPTT ~25-45 us per launch (profiles/r04/ab_bptt.txt).
#pragma once
#include "bf16x6_kernels.h"     // (brings f16_split.h in with the vector types it needs)

constexpr int GQ_FRAG_THREADS = 4 * 4 * 3 * 8 * 64;          // forward: [quarter][wave][gate][k-step] x 64 lanes
constexpr int BQ_FRAG_THREADS = 4 * 4 * 4 * 6 * 64;          // backward: [quarter][wave][destination quarter][row step] x 64 lanes
constexpr size_t GRU_FRAG_BYTES = (size_t)768 * 256 * 4;     // either form: 2 planes x 2 bytes per weight

// forward (gru_quad_kernel.h): [quarter][wave][gate][k-step][plane (hi, lo')][lane] uint4; A row = unit, 8 consecutive k per lane
__device__ __forceinline__ void prep_whh_quad_elem(const float* __restrict__ whh, uint4* __restrict__ frag, int idx) {
    if (idx >= GQ_FRAG_THREADS) return;                      // ((((q*4 + wv)*3 + g)*8 + s)*64 + lane)
    const int lane = idx & 63, s = (idx >> 6) & 7, g = (idx >> 9) % 3, qw = idx / (64 * 8 * 3), wv = qw & 3, q = qw >> 2;
    const float* wrow = whh + (size_t)(g * 256 + q * 64 + wv * 16 + (lane & 15)) * 256 + s * 32 + (lane >> 4) * 8;
    uint2 h0, l0, h1, l1;
    split2h_quad(*reinterpret_cast<const float4*>(wrow), h0, l0);
    split2h_quad(*reinterpret_cast<const float4*>(wrow + 4), h1, l1);
    uint4* o = frag + ((size_t)(((qw * 3 + g) * 8 + s) * 2) * 64 + lane);
    o[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
    o[64] = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

// backward (gru_bwd_quad_kernel.h): [quarter][wave][destination quarter][row step][plane][lane] uint4; A row m = k index
// 64 d + 16 wv + m, columns = 8 consecutive OWN gate rows kk = 32 s + 8 kg + i of quarter q (gate kk >> 6, unit 64 q + (kk & 63))
__device__ __forceinline__ void prep_whh_bwd_quad_elem(const float* __restrict__ whh, uint4* __restrict__ frag, int idx) {
    if (idx >= BQ_FRAG_THREADS) return;                      // ((((q*4 + wv)*4 + d)*6 + s)*64 + lane)
    const int lane = idx & 63, s = (idx >> 6) % 6, qwd = idx / (64 * 6), d = qwd & 3, wv = (qwd >> 2) & 3, q = qwd >> 4;
    const int kk0 = 32 * s + 8 * (lane >> 4);
    const float* wcol = whh + (size_t)((kk0 >> 6) * 256 + q * 64 + (kk0 & 63)) * 256 + 64 * d + 16 * wv + (lane & 15);
    unsigned hh[4], ll[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split2h_pair(wcol[(size_t)(2 * i) * 256], wcol[(size_t)(2 * i + 1) * 256], hh[i], ll[i]);
    uint4* o = frag + ((size_t)((qwd * 6 + s) * 2) * 64 + lane);
    o[0] = make_uint4(hh[0], hh[1], hh[2], hh[3]);
    o[64] = make_uint4(ll[0], ll[1], ll[2], ll[3]);
}
