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

#define SIR_BN_EPS 1e-5f

// ------------------------------------------------------------------------------------------
// weight preparation (tiny, runs at the head of every forward so it always sees current weights)
// ------------------------------------------------------------------------------------------

// conv weight [COUT][CIN][3][3] -> wp[g][co][8], g = (ci/8)*9 + tap, e = ci%8: one wave-load of
// the B operand (32 output channels x 8 input channels of one tap) is 1 KiB contiguous.
static __global__ void prep_conv_w_kernel(const float* __restrict__ w, float* __restrict__ wp, int cin, int cout) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = cin * 9 * cout;
    if (idx >= total) return;
    const int e = idx & 7, co = (idx >> 3) % cout, g = (idx >> 3) / cout;
    const int ci = (g / 9) * 8 + e, tap = g % 9;
    wp[idx] = w[((size_t)co * cin + ci) * 9 + tap];
}

// eval-mode BatchNorm folded to y = x*scale + shift (models/models.py:50-52, running stats)
static __global__ void prep_bn_kernel(const float* __restrict__ g, const float* __restrict__ b, const float* __restrict__ mean,
                               const float* __restrict__ var, float* __restrict__ scale, float* __restrict__ shift, int c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    const float s = g[i] / sqrtf(var[i] + SIR_BN_EPS);
    scale[i] = s;
    shift[i] = b[i] - mean[i] * s;
}

// W_hh [768][256] (k contiguous) -> wt[k/4][768][4]: lane = gate row, one 16-byte load carries 4 k
static __global__ void prep_whh_kernel(const float* __restrict__ w, float* __restrict__ wt) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // over 768*256
    if (idx >= 768 * 256) return;
    const int e = idx & 3, row = (idx >> 2) % 768, k4 = (idx >> 2) / 768;
    wt[idx] = w[(size_t)row * 256 + k4 * 4 + e];
}

// ------------------------------------------------------------------------------------------
// conv1 (1 -> 32 channels) + BN + ReLU + 2x2 max-pool, direct form (K = 9: memory-bound)
//   x [B][H=64][W] -> out NHWC [B][H/2][W/2][32];  lane&31 = channel, half-waves walk pixels
// ------------------------------------------------------------------------------------------
constexpr int C1_PROWS = 4, C1_PCOLS = 32;          // pooled pixels per block: 4 x 32
constexpr int C1_TR = 2 * C1_PROWS + 2, C1_TC = 2 * C1_PCOLS + 2;

static __global__ __launch_bounds__(256) void conv1_bn_relu_pool_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ out, int H, int W, int Hp, int Wp) {
    __shared__ float tile[C1_TR * C1_TC];
    const int b = blockIdx.z, py0 = blockIdx.y * C1_PROWS, px0 = blockIdx.x * C1_PCOLS;
    const int tid = threadIdx.x, c = tid & 31, slot = tid >> 5;
    const float* xb = x + (size_t)b * H * W;
    for (int i = tid; i < C1_TR * C1_TC; i += 256) {
        const int ty = i / C1_TC, tx = i - ty * C1_TC;
        const int gy = 2 * py0 - 1 + ty, gx = 2 * px0 - 1 + tx;
        tile[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? xb[(size_t)gy * W + gx] : 0.0f;
    }
    float wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wk[i] = w[c * 9 + i];
    const float s = scale[c], t = shift[c];
    __syncthreads();
    for (int i = 0; i < (C1_PROWS * C1_PCOLS) / 8; ++i) {
        const int pp = slot + 8 * i, pyl = pp / C1_PCOLS, pxl = pp % C1_PCOLS;
        const int py = py0 + pyl, px = px0 + pxl;
        if (py >= Hp || px >= Wp) continue;
        float in[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) in[r][q] = tile[(2 * pyl + r) * C1_TC + 2 * pxl + q];
        float best = 0.0f;                               // ReLU floor
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float a = 0.0f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) a = fmaf(in[dy + ky][dx + kx], wk[ky * 3 + kx], a);
                best = fmaxf(best, fmaf(a, s, t));
            }
        out[(((size_t)b * Hp + py) * Wp + px) * 32 + c] = best;
    }
}

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
static __global__ __launch_bounds__(256) void conv1_mfma_bn_relu_pool_kernel(
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
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = fmaxf(fmaf(acc0[r], sc[r], sh[r]), fmaf(acc1[r], sc[r], sh[r]));
                // neighbouring pixel = lane ^ 1: quad_perm(1, 0, 3, 2) on the DPP path (no LDS round trip)
                const float nb = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
                m[r] = fmaxf(fmaxf(v, nb), 0.0f);
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
// conv 3x3 (CIN -> COUT) + BN + ReLU + 2x2 max-pool as an implicit GEMM on fp32 MFMA.
//   M = pixels (A operand, from an NHWC input tile with halo staged in LDS),
//   N = output channels (B operand, streamed per wave from the prepared weights in L2),
//   K = (tap, ci).  One MFMA row tile = an 8-row x 4-col pixel patch whose bit layout
//   m = x0 | y0<<1 | x1<<2 | y1<<3 | y2<<4 puts every 2x2 pool window in registers 4q..4q+3 of
//   one lane, so pooling is four v_max per pooled pixel with no cross-lane traffic, and the
//   pooled store is 32 consecutive channels (128 B) per half-wave.
//   Workgroup = 4 waves x 2 patches = PR x PC patches; all COUT channels per wave (NT tiles).
// OUT_MODE 0: NHWC [B][Hp][Wp][COUT];  1: GRU input [B][Wp][COUT*Hp] with feature = co*Hp + py
//   (the permute(0,3,1,2).view of models/models.py:55-57, folded into the store).
// ------------------------------------------------------------------------------------------
// Measured (round 1, same box A/B): hipcc shrinks this kernel to ~96 VGPRs by sinking each B-operand
// load next to its use (load, s_waitcnt, 4 MFMAs, ...).  Forcing the source-level prefetch to stay
// (sched_barrier + amdgpu_waves_per_eu) costs 40+ VGPRs and LOST 15-20 %: with 64-cycle fp32 MFMAs,
// 5 resident waves per SIMD hide the load latency better than a deeper per-wave pipeline at 3.
template <int CIN, int COUT, int PR, int PC, int OUT_MODE, int MT = 2, int CK = 32>
__global__ __launch_bounds__(256, (MT == 1 ? 3 : 2)) void conv3x3_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ out, int H, int W, int Hp, int Wp,
    float2* __restrict__ stats = nullptr) {
    constexpr int NT = COUT / 32, PS = CK + 4;              // MT patches per wave, NT channel tiles, CK channels per LDS chunk
    constexpr int C4 = CK / 4, NIT = (CK / 8) * 9;          // float4 per pixel, (ci-group, tap) iterations per chunk
    constexpr int TR = 8 * PR, TC = 4 * PC, TROWS = TR + 2, TCOLS = TC + 2;
    static_assert(PR * PC == 4 * MT && CIN % CK == 0 && (MT == 1 || PR % 2 == 0), "tile shape");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.z, ty0 = blockIdx.y * TR, tx0 = blockIdx.x * TC;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m = lane & 31, kh = lane >> 5;
    const int pxl = (m & 1) + 2 * ((m >> 2) & 1);
    const int pyl = ((m >> 1) & 1) + 2 * ((m >> 3) & 1) + 4 * ((m >> 4) & 1);
    int aoff[MT], pr_[MT], pc_[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int pi = MT * wv + mt;             // column-major patch order: a wave's two patches share a
        pr_[mt] = pi % PR;                       // patch column, so a partially covered last tile leaves
        pc_[mt] = pi / PR;                       // whole waves idle instead of half-used waves
        aoff[mt] = ((8 * pr_[mt] + pyl) * TCOLS + 4 * pc_[mt] + pxl) * PS + kh * 4;
    }
    // a wave whose patch column starts at or beyond W contributes nothing: it skips its MFMA loop
    // (wave-uniform, made provably so with readfirstlane so the branch is scalar)
    bool pvalid[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) pvalid[mt] = __builtin_amdgcn_readfirstlane((tx0 + 4 * pc_[mt] < W) ? 1 : 0) != 0;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    const float4* wp4 = reinterpret_cast<const float4*>(wp);     // float4 index = (g*COUT + co)*2 + kh
    const float* xb = x + (size_t)b * H * W * CIN;

    for (int cc = 0; cc < CIN / CK; ++cc) {
        if (cc) __syncthreads();
        for (int idx = tid; idx < TROWS * TCOLS * C4; idx += 256) {
            const int pix = idx / C4, part = idx % C4;
            const int tyy = pix / TCOLS, txx = pix - tyy * TCOLS;
            const int gy = ty0 - 1 + tyy, gx = tx0 - 1 + txx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const float4*>(xb + ((size_t)gy * W + gx) * CIN + cc * CK + part * 4);
            *reinterpret_cast<float4*>(lds + pix * PS + part * 4) = v;
        }
        __syncthreads();
        if (!pvalid[0]) continue;                 // wave-uniform: this wave's patches lie beyond W
        float4 bcur[NT], bnxt[NT];
        const int g0 = cc * NIT;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bcur[nt] = wp4[((size_t)g0 * COUT + nt * 32 + m) * 2 + kh];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int cgl = it / 9, tap = it % 9, ky = tap / 3, kx = tap % 3;
            if (it + 1 < NIT) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bnxt[nt] = wp4[((size_t)(g0 + it + 1) * COUT + nt * 32 + m) * 2 + kh];
            }
            float4 a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                a[mt] = *reinterpret_cast<const float4*>(lds + aoff[mt] + (ky * TCOLS + kx) * PS + cgl * 8);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].x, bcur[nt].x, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].y, bcur[nt].y, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].z, bcur[nt].z, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].w, bcur[nt].w, acc[mt][nt], 0, 0, 0);
                }
            }
            if (it + 1 < NIT) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bcur[nt] = bnxt[nt];
            }
        }
    }

    if (OUT_MODE == 2) {
        // raw epilogue (training forward / data gradient): store the un-normalised conv output at
        // full resolution, NHWC, and optionally the per-channel (sum, sum of squares) of this
        // workgroup's valid pixels for the batch-statistics BatchNorm (deterministic partials).
        float ssum[NT], ssq[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { ssum[nt] = 0.0f; ssq[nt] = 0.0f; }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int xl = (r & 1) + 2 * kh, yl = ((r >> 1) & 1) + 2 * ((r >> 2) & 1) + 4 * ((r >> 3) & 1);
                const int gy = ty0 + 8 * pr_[mt] + yl, gx = tx0 + 4 * pc_[mt] + xl;
                if (gy < H && gx < W) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float v = acc[mt][nt][r];
                        out[(((size_t)b * H + gy) * W + gx) * COUT + nt * 32 + m] = v;
                        ssum[nt] += v;
                        ssq[nt] = fmaf(v, v, ssq[nt]);
                    }
                }
            }
        if (stats) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                ssum[nt] += __shfl_xor(ssum[nt], 32);
                ssq[nt] += __shfl_xor(ssq[nt], 32);
            }
            __syncthreads();                            // every wave is done reading the input tile
            if (kh == 0) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    lds[(wv * COUT + nt * 32 + m) * 2] = ssum[nt];
                    lds[(wv * COUT + nt * 32 + m) * 2 + 1] = ssq[nt];
                }
            }
            __syncthreads();
            const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            for (int c = tid; c < COUT; c += 256) {
                float s = 0.0f, q = 0.0f;
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) { s += lds[(w4 * COUT + c) * 2]; q += lds[(w4 * COUT + c) * 2 + 1]; }
                stats[blk * COUT + c] = make_float2(s, q);
            }
        }
        return;
    }

    // epilogue: BN (folded) -> ReLU -> 2x2 max over registers 4q..4q+3 -> store
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int PX = (tx0 + 4 * pc_[mt]) / 2 + kh;
        const int PYb = (ty0 + 8 * pr_[mt]) / 2;
        if (PX >= Wp) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = nt * 32 + m;
            const float s = scale[co], t = shift[co];
            float pooled[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) v = fmaxf(v, fmaf(acc[mt][nt][4 * q + r], s, t));
                pooled[q] = v;
            }
            if (OUT_MODE == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (PYb + q < Hp) out[(((size_t)b * Hp + PYb + q) * Wp + PX) * COUT + co] = pooled[q];
            } else {
                float* o = out + ((size_t)b * Wp + PX) * (COUT * Hp) + (size_t)co * Hp + PYb;
                if ((Hp & 3) == 0) {
                    *reinterpret_cast<float4*>(o) = make_float4(pooled[0], pooled[1], pooled[2], pooled[3]);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (PYb + q < Hp) o[q] = pooled[q];
                }
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
// forward NT GEMM, templated on the K-tile depth GF_K (32 measured faster than 64: more workgroups
// per CU beat fewer barriers)
template <int GF_K, bool HOIST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void gemm_nt_bias_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ B0, const float* __restrict__ B1, int ldb,
    const float* __restrict__ bias0, const float* __restrict__ bias1, float* __restrict__ C, int ldc,
    int M, int N, int K) {
    constexpr int GF_S = GF_K + 4;
    __shared__ __attribute__((aligned(16))) float As[GB_M * GF_S];
    __shared__ __attribute__((aligned(16))) float Bs[GB_N * GF_S];
    const int z = blockIdx.z;
    const float* __restrict__ B = z ? B1 : B0;
    const float* __restrict__ bias = z ? bias1 : bias0;
    const int m0 = blockIdx.y * GB_M, n0 = blockIdx.x * GB_N;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1, m = lane & 31, kh = lane >> 5;
    constexpr int C4 = GF_K / 4;                      // float4 per tile row
    constexpr int NA = GB_M * C4 / 256, NB = GB_N * C4 / 256;

    float4 ra[NA], rb[NB];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + 256 * i, row = idx / C4, c4 = idx % C4;
            ra[i] = (m0 + row < M) ? *reinterpret_cast<const float4*>(A + (size_t)(m0 + row) * lda + kt * GF_K + c4 * 4)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + 256 * i, row = idx / C4, c4 = idx % C4;
            rb[i] = (n0 + row < N) ? *reinterpret_cast<const float4*>(B + (size_t)(n0 + row) * ldb + kt * GF_K + c4 * 4)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + 256 * i, row = idx / C4, c4 = idx % C4;
            *reinterpret_cast<float4*>(As + row * GF_S + c4 * 4) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + 256 * i, row = idx / C4, c4 = idx % C4;
            *reinterpret_cast<float4*>(Bs + row * GF_S + c4 * 4) = rb[i];
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;

    const int nk = K / GF_K;
    load_tile(0);
    store_tile();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tile(kt + 1);
        float4 fa0[GF_K / 8], fa1[GF_K / 8], fb[GF_K / 8];
        if (HOIST) {
            // all operand fragments of the k-tile first (one exposed LDS latency per tile instead of one
            // per 16 MFMAs), pinned above the MFMAs
#pragma unroll
            for (int kk = 0; kk < GF_K / 8; ++kk) {
                fa0[kk] = *reinterpret_cast<const float4*>(As + (wm * 64 + m) * GF_S + kk * 8 + kh * 4);
                fa1[kk] = *reinterpret_cast<const float4*>(As + (wm * 64 + 32 + m) * GF_S + kk * 8 + kh * 4);
                fb[kk] = *reinterpret_cast<const float4*>(Bs + (wn * 32 + m) * GF_S + kk * 8 + kh * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kk = 0; kk < GF_K / 8; ++kk) {
            const float4 a0 = HOIST ? fa0[kk] : *reinterpret_cast<const float4*>(As + (wm * 64 + m) * GF_S + kk * 8 + kh * 4);
            const float4 a1 = HOIST ? fa1[kk] : *reinterpret_cast<const float4*>(As + (wm * 64 + 32 + m) * GF_S + kk * 8 + kh * 4);
            const float4 bq = HOIST ? fb[kk] : *reinterpret_cast<const float4*>(Bs + (wn * 32 + m) * GF_S + kk * 8 + kh * 4);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bq.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bq.x, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bq.y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bq.y, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bq.z, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, bq.z, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bq.w, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, bq.w, acc[1], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nk) {
            store_tile();
            __syncthreads();
        }
    }
    const int n = n0 + wn * 32 + m;
    if (n < N) {
        const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < M) C[(size_t)row * ldc + (size_t)z * N + n] = acc[mt][r] + bv;
            }
    }
}

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

template <bool SAVE>
__global__ __launch_bounds__(GRU_THREADS) void gru_recurrence_kernel(
    const float* __restrict__ gi, const float* __restrict__ wt, const float* __restrict__ bhh0,
    const float* __restrict__ bhh1, float* __restrict__ y, int B, int S, float* __restrict__ gates) {
    extern __shared__ __attribute__((aligned(16))) float glds[];
    float4* wl4 = reinterpret_cast<float4*>(glds);                   // [NPART][KLDS4][3][256] float4
    float* hs = glds + GRU_NPART * GRU_KLDS4 * 3 * 256 * 4;          // h[b][k]
    float* ps = hs + GRU_BW * GRU_H;                                 // partial[part][b][gate*256+u]
    const int dir = blockIdx.y, b0 = blockIdx.x * GRU_BW;
    const int tid = threadIdx.x, u = tid & 255, part = tid >> 8;
    const float4* w4 = reinterpret_cast<const float4*>(wt) + (size_t)dir * 64 * 768;
    const float* bhh = dir ? bhh1 : bhh0;
    const int kb4 = part * GRU_KPER4;                                // first k/4 group of this part
    // weights that stay on chip for the whole sequence
    float4 wr[GRU_KREG4][3];
#pragma unroll
    for (int i = 0; i < GRU_KREG4; ++i)
#pragma unroll
        for (int g = 0; g < 3; ++g) wr[i][g] = w4[(size_t)(kb4 + i) * 768 + g * 256 + u];
#pragma unroll
    for (int i = 0; i < GRU_KLDS4; ++i)
#pragma unroll
        for (int g = 0; g < 3; ++g)
            wl4[((part * GRU_KLDS4 + i) * 3 + g) * 256 + u] = w4[(size_t)(kb4 + GRU_KREG4 + i) * 768 + g * 256 + u];
    const float bh_r = bhh[u], bh_z = bhh[256 + u], bh_n = bhh[512 + u];
    for (int i = tid; i < GRU_BW * GRU_H; i += GRU_THREADS) hs[i] = 0.0f;
    float hprev[GRU_NQ];
#pragma unroll
    for (int q = 0; q < GRU_NQ; ++q) hprev[q] = 0.0f;
    __syncthreads();
    for (int step = 0; step < S; ++step) {
        const int t = dir ? (S - 1 - step) : step;
        float gr[GRU_NQ], gz[GRU_NQ], gn[GRU_NQ];
#pragma unroll
        for (int q = 0; q < GRU_NQ; ++q) {
            const int blq = part + GRU_NPART * q, bq = b0 + blq;
            gr[q] = gz[q] = gn[q] = 0.0f;
            if (blq < GRU_BW && bq < B) {
                const float* g = gi + ((size_t)bq * S + t) * 1536 + dir * 768;
                gr[q] = g[u]; gz[q] = g[256 + u]; gn[q] = g[512 + u];
            }
        }
        float acc[3][GRU_BW];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int bb = 0; bb < GRU_BW; ++bb) acc[g][bb] = 0.0f;
        // streamed part first (its loads fly while the resident parts compute)
#pragma unroll 2
        for (int i = 0; i < GRU_KSTR4; ++i) {
            const int k4 = kb4 + GRU_KREG4 + GRU_KLDS4 + i;
            const float4 w0 = w4[(size_t)k4 * 768 + u], w1 = w4[(size_t)k4 * 768 + 256 + u], w2 = w4[(size_t)k4 * 768 + 512 + u];
            float4 h4[GRU_BW];
#pragma unroll
            for (int bb = 0; bb < GRU_BW; ++bb) h4[bb] = *reinterpret_cast<const float4*>(hs + bb * GRU_H + k4 * 4);
            gru_fma4(acc[0], w0, h4); gru_fma4(acc[1], w1, h4); gru_fma4(acc[2], w2, h4);
        }
#pragma unroll
        for (int i = 0; i < GRU_KREG4; ++i) {
            float4 h4[GRU_BW];
#pragma unroll
            for (int bb = 0; bb < GRU_BW; ++bb) h4[bb] = *reinterpret_cast<const float4*>(hs + bb * GRU_H + (kb4 + i) * 4);
            gru_fma4(acc[0], wr[i][0], h4); gru_fma4(acc[1], wr[i][1], h4); gru_fma4(acc[2], wr[i][2], h4);
        }
#pragma unroll
        for (int i = 0; i < GRU_KLDS4; ++i) {
            float4 h4[GRU_BW];
#pragma unroll
            for (int bb = 0; bb < GRU_BW; ++bb)
                h4[bb] = *reinterpret_cast<const float4*>(hs + bb * GRU_H + (kb4 + GRU_KREG4 + i) * 4);
#pragma unroll
            for (int g = 0; g < 3; ++g) gru_fma4(acc[g], wl4[((part * GRU_KLDS4 + i) * 3 + g) * 256 + u], h4);
        }
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int bb = 0; bb < GRU_BW; ++bb) ps[((part * GRU_BW + bb) * 3 + g) * GRU_H + u] = acc[g][bb];
        __syncthreads();
        float hnew[GRU_NQ];
#pragma unroll
        for (int q = 0; q < GRU_NQ; ++q) {
            const int bl = part + GRU_NPART * q;
            hnew[q] = 0.0f;
            if (bl >= GRU_BW) continue;
            float hr = bh_r, hz = bh_z, hn = bh_n;
#pragma unroll
            for (int pp = 0; pp < GRU_NPART; ++pp) {
                hr += ps[((pp * GRU_BW + bl) * 3 + 0) * GRU_H + u];
                hz += ps[((pp * GRU_BW + bl) * 3 + 1) * GRU_H + u];
                hn += ps[((pp * GRU_BW + bl) * 3 + 2) * GRU_H + u];
            }
            const float r = sigmoidf_(gr[q] + hr);
            const float zg = sigmoidf_(gz[q] + hz);
            const float nn = tanhf(gn[q] + r * hn);
            hnew[q] = (1.0f - zg) * nn + zg * hprev[q];
            hprev[q] = hnew[q];
            if (b0 + bl < B) {
                y[((size_t)(b0 + bl) * S + t) * 512 + dir * 256 + u] = hnew[q];
                if (SAVE) {
                    float* gs = gates + (((size_t)(b0 + bl) * S + t) * 2 + dir) * 1024;
                    gs[u] = r; gs[256 + u] = zg; gs[512 + u] = nn; gs[768 + u] = hn;
                }
            }
        }
        __syncthreads();                       // every partial consumed, every old h read
#pragma unroll
        for (int q = 0; q < GRU_NQ; ++q)
            if (part + GRU_NPART * q < GRU_BW) hs[(part + GRU_NPART * q) * GRU_H + u] = hnew[q];
        __syncthreads();
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

// first index of the row maximum (torch.argmax / torch.max semantics on ties)
static __global__ void argmax_rows_kernel(const float* __restrict__ logits, long long* __restrict__ idx, int B, int C) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* r = logits + (size_t)b * C;
    float best = r[0];
    int bi = 0;
    for (int c = 1; c < C; ++c)
        if (r[c] > best) { best = r[c]; bi = c; }
    idx[b] = bi;
}
