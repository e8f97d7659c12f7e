// Device kernels of the CNN + BiGRU + attention model (inference forms), gfx950.
// Included by model_infer.hip (and later by the training translation unit).
//
// Activation layout is channels-last ("NHWC") so that (a) conv outputs are stored as 128-byte
// runs of 32 consecutive channels per pixel and (b) the implicit-GEMM convolutions read their
// pixel-major A operand from LDS with one ds_read_b128 per four MFMA steps.
// All contractions use v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate, an exact fmaf chain:
// cdna guide section 3 "FP32-input MFMA") because the parity bar is identical argmax vs the fp32
// CPU path; lane maps: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D col=l&31,
// row=(r&3)+8*(r>>2)+4*(l>>5).
#pragma once
#include "sir_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float mk_f32x2 __attribute__((ext_vector_type(2)));

#define SIR_BN_EPS 1e-5f

// ------------------------------------------------------------------------------------------
// weight preparation (tiny, runs at the head of every forward so it always sees current weights)
// ------------------------------------------------------------------------------------------

// eval-mode BatchNorm folded to y = x*scale + shift (models/models.py:50-52, running stats)
static __global__ void prep_bn_kernel(const float* __restrict__ g, const float* __restrict__ b, const float* __restrict__ mean,
                               const float* __restrict__ var, float* __restrict__ scale, float* __restrict__ shift, int c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    const float s = g[i] / sqrtf(var[i] + SIR_BN_EPS);
    scale[i] = s;
    shift[i] = b[i] - mean[i] * s;
}

// ------------------------------------------------------------------------------------------
// conv1 (1 -> 32 channels) + BN + ReLU + 2x2 max-pool, direct form (K = 9: memory-bound)
//   x [B][H=64][W] -> out NHWC [B][H/2][W/2][32];  lane&31 = channel, half-waves walk pixels
// ------------------------------------------------------------------------------------------
constexpr int C1_PROWS = 4, C1_PCOLS = 32;          // pooled pixels per block: 4 x 32
constexpr int C1_TR = 2 * C1_PROWS + 2, C1_TC = 2 * C1_PCOLS + 2;

// The same block on the f32 matrix pipe: the direct form above is bound by VALU issue (36 FMAs per output), and
// v_mfma_f32_32x32x2_f32 does 2048 of them per instruction at the packed-f32 vector rate while the VALU is free for the
// BN / ReLU / pooling epilogue.  GEMM view: D[channel i][pixel j] = sum_tap w[i][tap] x_tap[j], K = 9 taps padded to 10
// (five K = 2 steps; exact f32, an fmaf chain over the taps in order -- the same arithmetic as the direct form).
//   A operand (lane l): w[l & 31][2 s + (l >> 5)], five registers for the whole block;
//   B operand (lane l): tile[(row + ky) * C1_TC + col0 + (l & 31) + kx] for tap = 2 s + (l >> 5): one ds_read_b32;
//   D: lane l holds pixel l & 31, channels (r & 3) + 8 (r >> 2) + 4 (l >> 5), r = 0..15.
// Wave w = pooled row w of the 4 x 32 pooled-pixel block; per half (32 conv columns) two accumulators (the two conv rows
// of the pooled row); 2x2 max = max of the two accumulators and of lanes j, j ^ 1.  Even lanes then store channel groups
// 0, 1 and odd lanes groups 2, 3 of pooled pixel j / 2 (two float4 stores per lane).
static __global__ __launch_bounds__(256, 4) void conv1_mfma_bn_relu_pool_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ out, int H, int W, int Hp, int Wp) {
    // One block walks ALL row tiles of its column strip (grid = (column strips, 1, B)): weights / scale / shift are
    // loaded once, and the next tile's pixels are fetched into registers while the matrix pipe works on the current one
    // (one short block per tile spent most of its life waiting for its own loads: 50 us for 17 us of MFMA work).
    __shared__ float tiles[2][C1_TR * C1_TC];
    const int b = blockIdx.z, px0 = blockIdx.x * C1_PCOLS;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 31, kh = lane >> 5;
    const float* xb = x + (size_t)b * H * W;
    constexpr int NPRE = (C1_TR * C1_TC + 255) / 256;
    float pre[NPRE];
    auto fetch = [&](int py0) {
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
            const int i = tid + 256 * q, ty = i / C1_TC, tx = i - ty * C1_TC;
            const int gy = 2 * py0 - 1 + ty, gx = 2 * px0 - 1 + tx;
            pre[q] = (i < C1_TR * C1_TC && gy >= 0 && gy < H && gx >= 0 && gx < W) ? xb[(size_t)gy * W + gx] : 0.0f;
        }
    };
    fetch(0);
    float wa[5];
    int boff[5];                                            // tile offset of this lane's tap in step s
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int tap = 2 * s + kh;
        wa[s] = tap < 9 ? w[j * 9 + tap] : 0.0f;            // j doubles as the channel index i of the A operand
        const int tc = tap < 9 ? tap : 8;
        boff[s] = (tc / 3) * C1_TC + (tc % 3) + j;
    }
    float sc[16], sh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ch = (r & 3) + 8 * (r >> 2) + 4 * kh;
        sc[r] = scale[ch]; sh[r] = shift[ch];
    }
    const bool odd = j & 1;                                  // even lanes store channel groups 0, 1; odd lanes groups 2, 3
    int it = 0;
    for (int py0 = 0; py0 < Hp; py0 += C1_PROWS, ++it) {
        float* tile = tiles[it & 1];
#pragma unroll
        for (int q = 0; q < NPRE; ++q)
            if (tid + 256 * q < C1_TR * C1_TC) tile[tid + 256 * q] = pre[q];
        __syncthreads();                                     // (the other buffer was last read before the previous barrier)
        if (py0 + C1_PROWS < Hp) fetch(py0 + C1_PROWS);
        const int py = py0 + wv;
        if (py >= Hp) continue;
        const float* trow = tile + (2 * wv) * C1_TC;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (px0 + 16 * h >= Wp) break;                   // the last strip of a 100-column image has 4 valid columns
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const float b0 = trow[boff[s] + 32 * h], b1 = trow[boff[s] + 32 * h + C1_TC];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s], b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s], b1, acc1, 0, 0, 0);
            }
            float m[16];
            // BN of both conv rows as packed-f32 pairs (v_pk_fma_f32 / v_pk_max_f32: the same roundings, half the instructions -- the f32
            // MFMAs above run on the SAME ALUs as the vector f32 ops, so every instruction of this epilogue adds to the kernel's time)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const mk_f32x2 s2 = {sc[r], sc[r + 1]}, h2 = {sh[r], sh[r + 1]};
                const mk_f32x2 y0 = __builtin_elementwise_fma((mk_f32x2){acc0[r], acc0[r + 1]}, s2, h2);
                const mk_f32x2 y1 = __builtin_elementwise_fma((mk_f32x2){acc1[r], acc1[r + 1]}, s2, h2);
                // (scalar from here on: with the maximum taken on the pair, hipcc 7.2 expands the two DPP moves of its halves to TWO moves of
                // the SAME half -- element r + 1 then pooled against element r's neighbour)
                const float v0 = fmaxf(y0.x, y1.x), v1 = fmaxf(y0.y, y1.y);
                // neighbouring pixel = lane ^ 1: quad_perm(1, 0, 3, 2) on the DPP path (no LDS round trip)
                const float nb0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v0), 0xB1, 0xF, 0xF, true));
                const float nb1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v1), 0xB1, 0xF, 0xF, true));
                m[r] = fmaxf(fmaxf(v0, nb0), 0.0f);
                m[r + 1] = fmaxf(fmaxf(v1, nb1), 0.0f);
            }
            const int px = px0 + 16 * h + (j >> 1);
            const float4 lo = odd ? make_float4(m[8], m[9], m[10], m[11]) : make_float4(m[0], m[1], m[2], m[3]);
            const float4 hi = odd ? make_float4(m[12], m[13], m[14], m[15]) : make_float4(m[4], m[5], m[6], m[7]);
            if (px < Wp) {
                float* o = out + (((size_t)b * Hp + py) * Wp + px) * 32 + 4 * kh + (odd ? 16 : 0);
                *reinterpret_cast<float4*>(o) = lo;
                *reinterpret_cast<float4*>(o + 8) = hi;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// C[m][z*N + n] = sum_k A[m][k] * Bz[n][k] + biasz[n]    (fp32 MFMA, "NT": both operands k-contiguous)
//   used for the GRU input projections (z = direction) and the classifier head.
//   128 x 64 block tile, BK = 32, 4 waves as 2(M) x 2(N), wave tile 64 x 32; LDS rows padded to 36
//   floats so the ds_read_b128 operand reads are bank-conflict-free; global->register prefetch of
//   the next k-tile overlaps the MFMAs of the current one.
// ------------------------------------------------------------------------------------------
constexpr int GB_M = 128, GB_N = 64, GB_K = 32, GB_S = GB_K + 4;     // general (backward) GEMM tile

// ------------------------------------------------------------------------------------------
// GRU recurrence of one layer, both directions (torch.nn.GRU cell, gate order r,z,n, h0 = 0):
//   gi  [B*S][1536]   = x W_ih^T + b_ih for both directions (from the GEMM above)
//   wt  [2][64][768][4] transposed W_hh (prep_whh_kernel), bhh [2][768]
//   y   [B][S][512]   direction d writes columns d*256 .. d*256+255
// One workgroup = GRU_BW utterances of one direction for all S steps: no inter-workgroup traffic.
// 512 threads = 256 hidden units x 2 K-halves; a thread owns the 3 gate rows of its unit over
// its 128-wide K half.  W_hh is 786 KB per direction -- more than a CU can hold -- and
// re-streaming all of it from L2 every step is bound by the per-CU load path (round-1 profile:
// 7.5 us/step).  So 34 % of each thread's weights stay in REGISTERS and 12.5 % in LDS for all S
// steps (loaded once), and only the remaining 53 % is streamed per step as coalesced 16-byte
// loads, which the FMA work of the step now covers.  h is broadcast from LDS; the two K-halves are
// summed through LDS, then thread (u, b) applies the gate math for two utterances.
// SAVE: also store (r, z, n, W_hn h + b_hn) per step for back-propagation through time,
//   gates [B][S][2 dirs][4][256]
// ------------------------------------------------------------------------------------------
constexpr int GRU_H = 256, GRU_BW = 2;              // utterances per workgroup: 2 -> B/2 x 2 = 256 workgroups at B=256
constexpr int GRU_NPART = 4;                        // K split: threads = 256 units x NPART
constexpr int GRU_KPER4 = 64 / GRU_NPART;           // k/4 groups per part
constexpr int GRU_KREG4 = 3;                        // k/4 groups per part held in registers
constexpr int GRU_KLDS4 = 2;                        // ... held in LDS
constexpr int GRU_KSTR4 = GRU_KPER4 - GRU_KREG4 - GRU_KLDS4;   // ... streamed from L2 each step
constexpr int GRU_NQ = (GRU_BW + GRU_NPART - 1) / GRU_NPART;   // utterances finished per thread (bl = part + NPART*q)
constexpr int GRU_THREADS = 256 * GRU_NPART;
constexpr size_t GRU_LDS_BYTES =
    (size_t)(GRU_NPART * GRU_KLDS4 * 3 * 256 * 4 + GRU_BW * GRU_H + GRU_NPART * GRU_BW * 3 * GRU_H) * 4;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int NB>
__device__ __forceinline__ void gru_fma4(float (&acc)[NB], const float4 w, const float4 (&h4)[NB]) {
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
        acc[bb] = fmaf(w.x, h4[bb].x, acc[bb]); acc[bb] = fmaf(w.y, h4[bb].y, acc[bb]);
        acc[bb] = fmaf(w.z, h4[bb].z, acc[bb]); acc[bb] = fmaf(w.w, h4[bb].w, acc[bb]);
    }
}

// ------------------------------------------------------------------------------------------
// attention pooling (models/models.py:63-64): scores = y a + b; softmax over time; ctx = sum_t w_t y_t
// one workgroup per utterance, wave shuffles for the 512-wide dots
// ------------------------------------------------------------------------------------------
constexpr int ATT_MAX_S = 256;

static __global__ __launch_bounds__(256) void attention_pool_kernel(const float* __restrict__ y, const float* __restrict__ aw,
                                                             const float* __restrict__ ab, float* __restrict__ ctx,
                                                             int S, const float* __restrict__ fcw,
                                                             const float* __restrict__ fcb, int C,
                                                             float* __restrict__ logits, long long* __restrict__ amax) {
    // fused tail (models.py:63-67 + evaluate.py:83): attention pooling, the 512 -> C classifier and the
    // arg-max of one utterance per workgroup.  The head is 31.7 kFLOP per utterance; as a separate
    // 128x64-tile MFMA GEMM it occupied 2 workgroups and cost 34 us per batch, fused here it is free.
    __shared__ float sc[ATT_MAX_S];
    __shared__ float cs[512];
    __shared__ float lg[64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* yb = y + (size_t)b * S * 512;
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = aw[lane + 64 * i];
    // four time steps per wave and round with all 32 loads issued before the first reduction (one step at a time, the
    // seven dependent load -> reduce rounds of a wave were half of this kernel's time)
    for (int t0 = 4 * wv; t0 < S; t0 += 16) {
        float v[4][8];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[k][i] = (t0 + k < S) ? yb[(size_t)(t0 + k) * 512 + lane + 64 * i] : 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float d = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) d = fmaf(v[k][i], a[i], d);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
            if (lane == 0 && t0 + k < S) sc[t0 + k] = d + ab[0];
        }
    }
    __syncthreads();
    // softmax weights once per time step (thread t), not once per (channel, time step): the exponentials were 5000
    // instructions per thread and most of this kernel's 21 us
    __shared__ float ex[ATT_MAX_S];
    float mx = -INFINITY;
    for (int t = 0; t < S; ++t) mx = fmaxf(mx, sc[t]);
    if (tid < S) ex[tid] = expf(sc[tid] - mx);
    __syncthreads();
    float den = 0.0f;
    for (int t = 0; t < S; ++t) den += ex[t];
    __syncthreads();                                          // every thread has read ex[] before it is overwritten
    if (tid < S) ex[tid] = ex[tid] / den;
    __syncthreads();
    for (int c = tid; c < 512; c += 256) {
        float acc = 0.0f;
        for (int t = 0; t < S; ++t) acc = fmaf(ex[t], yb[(size_t)t * 512 + c], acc);
        ctx[(size_t)b * 512 + c] = acc;
        cs[c] = acc;
    }
    if (!logits) return;
    __syncthreads();
    // classifier: thread = (class j = tid / 8 (+ 32), eighth of the 512 inputs); sixteen independent float4 loads per thread,
    // then a reduction over the eight lanes of a class (C <= 64)
    for (int j = tid >> 3; j < C; j += 32) {
        const int part = tid & 7;
        const float4* wr = reinterpret_cast<const float4*>(fcw + (size_t)j * 512 + part * 64);
        const float4* cr = reinterpret_cast<const float4*>(cs + part * 64);
        float d = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float4 wv4 = wr[i], c4 = cr[i];
            d = fmaf(c4.x, wv4.x, d); d = fmaf(c4.y, wv4.y, d); d = fmaf(c4.z, wv4.z, d); d = fmaf(c4.w, wv4.w, d);
        }
        d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4);
        if (part == 0) {
            d += fcb[j];
            lg[j] = d;
            logits[(size_t)b * C + j] = d;
        }
    }
    if (!amax) return;
    __syncthreads();
    if (tid == 0) {
        float best = lg[0];
        int bi = 0;
        for (int c = 1; c < C; ++c)
            if (lg[c] > best) { best = lg[c]; bi = c; }      // first maximum, as torch.argmax
        amax[b] = bi;
    }
}

