// Superseded kernel generations, kept OUT of libsir_hip.so: the A/B harnesses (devtools/kernel_ab/bench_gemm.hip, devtools/kernel_ab/bench_conv.hip) include
// this header to time them against the product kernels and to compare results bitwise.  Nothing under csrc/ references them.
//   fp32-MFMA convolutions / GEMMs / weight gradients (v_mfma_f32_32x32x2_f32), the first bf16x6 GEMM and convolution, the
//   direct VALU conv1, the streaming and the paired fp32-FMA GRU recurrences, the L2-streaming BPTT kernel, the BatchNorm
//   backward reduction over the raw conv outputs, conv1 statistics by recomputation.
#pragma once
#include "../csrc/bf16x6_kernels.h"
#include "../csrc/train_kernels.h"
#include "../csrc/gru_pair_kernel.h"

// LDS image of the convolution input tile: 48-byte pixels (16 channels x 3 planes kept in separate plane
// blocks), rows padded to a stride of 4 (mod 8) sixteen-byte slots.  ds_read_b128 serves the lanes in the
// groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): with the 8x4 patch bit layout of a fragment those 16
// lanes then hit 16 different slots of the 256-byte bank line (the natural strides 30 and 54 give 2-way
// conflicts on two lane pairs per group: SQ_LDS_BANK_CONFLICT = 50 % of LDS cycles).
// Padding costs LDS: the 32x8-pixel tiles (PC = 2) are left unpadded (48,960 B) so that THREE workgroups fit a CU --
// measured 175 -> 154 us for conv2 and 229 -> 195 us for its data gradient, against ~3 % from the conflict-free reads.
constexpr int conv_bf16x6_row_bytes(int PC, bool pad = true) {
    const int slots = (4 * PC + 2) * 3;
    return (pad ? slots + ((4 - slots % 8) + 8) % 8 : slots) * 16;
}
constexpr size_t conv_bf16x6_lds_bytes(int PR, int PC, bool pad = true) { return (size_t)3 * (8 * PR + 2) * conv_bf16x6_row_bytes(PC, pad); }


// ---- from csrc/model_kernels.h ------------------------------------------------------------
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

// ---- from csrc/model_kernels.h ------------------------------------------------------------
// W_hh [768][256] (k contiguous) -> wt[k/4][768][4]: lane = gate row, one 16-byte load carries 4 k
static __global__ void prep_whh_kernel(const float* __restrict__ w, float* __restrict__ wt) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // over 768*256
    if (idx >= 768 * 256) return;
    const int e = idx & 3, row = (idx >> 2) % 768, k4 = (idx >> 2) / 768;
    wt[idx] = w[(size_t)row * 256 + k4 * 4 + e];
}

// ---- from csrc/model_kernels.h ------------------------------------------------------------
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

// ---- from csrc/model_kernels.h ------------------------------------------------------------
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

// ---- from csrc/model_kernels.h ------------------------------------------------------------
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

// ---- from csrc/model_kernels.h ------------------------------------------------------------
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

// ---- from csrc/model_kernels.h ------------------------------------------------------------
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

// ---- from csrc/bf16x6_kernels.h ------------------------------------------------------------
static __global__ __launch_bounds__(256) void gemm_nt_bf16x6_kernel(
    const unsigned short* __restrict__ Ap, const unsigned short* __restrict__ Bp0, const unsigned short* __restrict__ Bp1,
    const float* __restrict__ bias0, const float* __restrict__ bias1, float* __restrict__ C, int ldc, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) unsigned char As[3 * GB_M * XB_ROW];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[3 * GB_N * XB_ROW];
    const int z = blockIdx.z;
    const unsigned short* __restrict__ Bp = z ? Bp1 : Bp0;
    const float* __restrict__ bias = z ? bias1 : bias0;
    const int m0 = blockIdx.y * GB_M, n0 = blockIdx.x * GB_N;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1, m = lane & 31, h = lane >> 5;
    const size_t planeA = (size_t)M * K, planeB = (size_t)N * K;

    uint4 ra[6], rb[3];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {                       // 3 planes x 128 rows x 4 chunks of 16 B
            const int idx = tid + 256 * i, p = idx >> 9, row = (idx >> 2) & 127, c = idx & 3;
            ra[i] = (m0 + row < M) ? *reinterpret_cast<const uint4*>(Ap + p * planeA + (size_t)(m0 + row) * K + kt * 32 + c * 8)
                                   : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {                       // 3 planes x 64 rows x 4 chunks
            const int idx = tid + 256 * i, p = idx >> 8, row = (idx >> 2) & 63, c = idx & 3;
            rb[i] = (n0 + row < N) ? *reinterpret_cast<const uint4*>(Bp + p * planeB + (size_t)(n0 + row) * K + kt * 32 + c * 8)
                                   : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int idx = tid + 256 * i, p = idx >> 9, row = (idx >> 2) & 127, c = idx & 3;
            *reinterpret_cast<uint4*>(As + (p * GB_M + row) * XB_ROW + c * 16) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = tid + 256 * i, p = idx >> 8, row = (idx >> 2) & 63, c = idx & 3;
            *reinterpret_cast<uint4*>(Bs + (p * GB_N + row) * XB_ROW + c * 16) = rb[i];
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;

    const int nk = K / 32;
    load_tile(0);
    store_tile();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a0[3], a1[3], b[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a0[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(As + (p * GB_M + wm * 64 + m) * XB_ROW + ks * 32 + h * 16));
                a1[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(As + (p * GB_M + wm * 64 + 32 + m) * XB_ROW + ks * 32 + h * 16));
                b[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Bs + (p * GB_N + wn * 32 + m) * XB_ROW + ks * 32 + h * 16));
            }
            // small terms first: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi   (planes: 0 = hi, 1 = mid, 2 = lo)
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[2], b[0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[2], b[0], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], b[2], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b[2], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[1], b[1], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[1], b[1], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[1], b[0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[1], b[0], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], b[1], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b[1], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], b[0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b[0], acc[1], 0, 0, 0);
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
                const int row = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < M) C[(size_t)row * ldc + (size_t)z * N + n] = acc[mt][r] + bv;
            }
    }
}

// ---- from csrc/bf16x6_kernels.h ------------------------------------------------------------
template <int CIN, int COUT, int PR, int PC, int OUT_MODE, int MT>
__global__ __launch_bounds__(256, 2) void conv3x3_bf16x6_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ wpb, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ out, int H, int W, int Hp, int Wp, float2* __restrict__ stats) {
    constexpr int NT = COUT / 32, CK = 16, PSB = 48;
    constexpr int TR = 8 * PR, TC = 4 * PC, TROWS = TR + 2, TCOLS = TC + 2;
    constexpr int RSB = conv_bf16x6_row_bytes(PC);          // padded row stride
    constexpr int PLANE = TROWS * RSB;                      // bytes per plane
    constexpr int G = (CIN / 16) * 9;
    static_assert(PR * PC == 4 * MT && CIN % CK == 0 && (MT == 1 || PR % 2 == 0), "tile shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int b = blockIdx.z, ty0 = blockIdx.y * TR, tx0 = blockIdx.x * TC;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m = lane & 31, h = lane >> 5;
    const int pxl = (m & 1) + 2 * ((m >> 2) & 1);
    const int pyl = ((m >> 1) & 1) + 2 * ((m >> 3) & 1) + 4 * ((m >> 4) & 1);
    int aoff[MT], pr_[MT], pc_[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int pi = MT * wv + mt;
        pr_[mt] = pi % PR;
        pc_[mt] = pi / PR;
        aoff[mt] = (8 * pr_[mt] + pyl) * RSB + (4 * pc_[mt] + pxl) * PSB + h * 16;
    }
    const bool wvalid = __builtin_amdgcn_readfirstlane((tx0 + 4 * pc_[0] < W) ? 1 : 0) != 0;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
    const uint4* wp4 = reinterpret_cast<const uint4*>(wpb);  // uint4 index = ((p*G + g)*COUT + co)*2 + h
    const float* xb = x + (size_t)b * H * W * CIN;

    for (int cc = 0; cc < CIN / CK; ++cc) {
        if (cc) __syncthreads();
        for (int idx = tid; idx < TROWS * TCOLS * 4; idx += 256) {
            const int pix = idx >> 2, part = idx & 3;
            const int tyy = pix / TCOLS, txx = pix - tyy * TCOLS;
            const int gy = ty0 - 1 + tyy, gx = tx0 - 1 + txx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const float4*>(xb + ((size_t)gy * W + gx) * CIN + cc * CK + part * 4);
            uint2 hh, mm, ll;
            split3_quad(v, hh, mm, ll);
            unsigned char* d = ldsb + tyy * RSB + txx * PSB + part * 8;
            *reinterpret_cast<uint2*>(d) = hh;
            *reinterpret_cast<uint2*>(d + PLANE) = mm;
            *reinterpret_cast<uint2*>(d + 2 * PLANE) = ll;
        }
        __syncthreads();
        if (!wvalid) continue;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3, g = cc * 9 + tap;
            bf16x8 bfr[NT][3], afr[MT][3];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    bfr[nt][p] = __builtin_bit_cast(bf16x8, wp4[(((size_t)p * G + g) * COUT + nt * 32 + m) * 2 + h]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    afr[mt][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ldsb + p * PLANE + aoff[mt] + ky * RSB + kx * PSB));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[mt][2], bfr[nt][0], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[mt][0], bfr[nt][2], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[mt][1], bfr[nt][1], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[mt][1], bfr[nt][0], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[mt][0], bfr[nt][1], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[mt][0], bfr[nt][0], acc[mt][nt], 0, 0, 0);
                }
        }
    }
    if (OUT_MODE == 2) {
        // raw epilogue (training forward / data gradient), identical to conv3x3_mfma_kernel's
        float ssum[NT], ssq[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { ssum[nt] = 0.0f; ssq[nt] = 0.0f; }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int xl = (r & 1) + 2 * h, yl = ((r >> 1) & 1) + 2 * ((r >> 2) & 1) + 4 * ((r >> 3) & 1);
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
            float* lds = reinterpret_cast<float*>(ldsb);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                ssum[nt] += __shfl_xor(ssum[nt], 32);
                ssq[nt] += __shfl_xor(ssq[nt], 32);
            }
            __syncthreads();
            if (h == 0) {
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
    // epilogue: BN (folded) -> ReLU -> 2x2 max over registers 4q..4q+3 -> store (as conv3x3_mfma_kernel)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int PX = (tx0 + 4 * pc_[mt]) / 2 + h;
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

// ---- from csrc/train_kernels.h ------------------------------------------------------------
// conv1 is recomputed instead of stored (9 MACs per output): this pass only accumulates the
// per-channel (sum, sum of squares) of the raw conv1 output over a 8 x 64 pixel tile.
static __global__ __launch_bounds__(256) void conv1_stats_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           float2* __restrict__ stats, int H, int W) {
    __shared__ float tile[C1_TR * C1_TC];
    __shared__ float red[8 * 32 * 2];
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
    __syncthreads();
    float s = 0.0f, q = 0.0f;
    for (int i = 0; i < (C1_PROWS * C1_PCOLS) / 8; ++i) {
        const int pp = slot + 8 * i, pyl = pp / C1_PCOLS, pxl = pp % C1_PCOLS;
        float in[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) in[r][k] = tile[(2 * pyl + r) * C1_TC + 2 * pxl + k];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int gy = 2 * (py0 + pyl) + dy, gx = 2 * (px0 + pxl) + dx;
                if (gy >= H || gx >= W) continue;
                float a = 0.0f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) a = fmaf(in[dy + ky][dx + kx], wk[ky * 3 + kx], a);
                s += a;
                q = fmaf(a, a, q);
            }
    }
    red[(slot * 32 + c) * 2] = s;
    red[(slot * 32 + c) * 2 + 1] = q;
    __syncthreads();
    if (tid < 32) {
        float ts = 0.0f, tq = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { ts += red[(k * 32 + tid) * 2]; tq += red[(k * 32 + tid) * 2 + 1]; }
        const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        stats[blk * 32 + tid] = make_float2(ts, tq);
    }
}

// ---- from csrc/train_kernels.h ------------------------------------------------------------
static __global__ void prep_whh_bwd_kernel(const float* __restrict__ w, float* __restrict__ wr4) {
    prep_whh_bwd_elem(w, wr4, blockIdx.x * blockDim.x + threadIdx.x);
}

// ---- from csrc/train_kernels.h ------------------------------------------------------------
static __global__ __launch_bounds__(1024) void gru_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ gates,
                                                        const float* __restrict__ y, const float* __restrict__ wr4,
                                                        float* __restrict__ dgi, float* __restrict__ dgh,
                                                        float* __restrict__ bsum_i, float* __restrict__ bsum_h, int B,
                                                        int S) {
    // bsum_i / bsum_h [B][1536]: per-utterance sums over time of dgi / dgh (bias gradients are their
    // column sums over B rows instead of B*S rows)
    __shared__ __attribute__((aligned(16))) float gsh[GRU_BBW * 768];       // dgh[b][row]
    __shared__ float ps[4 * GRU_BBW * GRU_H];                               // partial[rs][b][k]
    const int dir = blockIdx.y, b0 = blockIdx.x * GRU_BBW;
    const int tid = threadIdx.x, u = tid & 255, ks = tid >> 8;
    const int bme = ks;
    const bool bvalid = (b0 + bme) < B;
    const float4* w4 = reinterpret_cast<const float4*>(wr4) + (size_t)dir * 192 * 256;
    float dh_carry = 0.0f;
    float sum_r = 0.f, sum_z = 0.f, sum_n = 0.f, sum_nr = 0.f;
    for (int step = S - 1; step >= 0; --step) {
        const int t = dir ? (S - 1 - step) : step;                // time index processed at `step`
        const int tp = dir ? t + 1 : t - 1;                       // where h_prev lives (invalid at step 0)
        float drp = 0.f, dzp = 0.f, dnp = 0.f, dnr = 0.f, dhz = 0.f;
        if (bvalid) {
            const size_t row = (size_t)(b0 + bme) * S + t;
            const float* gs = gates + (row * 2 + dir) * 1024;
            const float r = gs[u], zg = gs[256 + u], nn = gs[512 + u], hn = gs[768 + u];
            const float hprev = (step > 0) ? y[((size_t)(b0 + bme) * S + tp) * 512 + dir * 256 + u] : 0.0f;
            const float dh = dy[row * 512 + dir * 256 + u] + dh_carry;
            const float dn = dh * (1.0f - zg);
            const float dz = dh * (hprev - nn);
            dnp = dn * (1.0f - nn * nn);
            drp = dnp * hn * r * (1.0f - r);
            dzp = dz * zg * (1.0f - zg);
            dnr = dnp * r;
            dhz = dh * zg;
            float* gi_o = dgi + row * 1536 + dir * 768;
            float* gh_o = dgh + row * 1536 + dir * 768;
            gi_o[u] = drp; gi_o[256 + u] = dzp; gi_o[512 + u] = dnp;
            gh_o[u] = drp; gh_o[256 + u] = dzp; gh_o[512 + u] = dnr;
            sum_r += drp; sum_z += dzp; sum_n += dnp; sum_nr += dnr;
        }
        gsh[bme * 768 + u] = drp; gsh[bme * 768 + 256 + u] = dzp; gsh[bme * 768 + 512 + u] = dnr;
        __syncthreads();
        float acc[GRU_BBW];
#pragma unroll
        for (int bb = 0; bb < GRU_BBW; ++bb) acc[bb] = 0.0f;
#pragma unroll 4
        for (int r4 = ks * 48; r4 < ks * 48 + 48; ++r4) {
            const float4 wv = w4[(size_t)r4 * 256 + u];
#pragma unroll
            for (int bb = 0; bb < GRU_BBW; ++bb) {
                const float4 g4 = *reinterpret_cast<const float4*>(gsh + bb * 768 + r4 * 4);
                acc[bb] = fmaf(wv.x, g4.x, acc[bb]); acc[bb] = fmaf(wv.y, g4.y, acc[bb]);
                acc[bb] = fmaf(wv.z, g4.z, acc[bb]); acc[bb] = fmaf(wv.w, g4.w, acc[bb]);
            }
        }
#pragma unroll
        for (int bb = 0; bb < GRU_BBW; ++bb) ps[(ks * GRU_BBW + bb) * GRU_H + u] = acc[bb];
        __syncthreads();
        dh_carry = dhz + ps[(0 * GRU_BBW + bme) * GRU_H + u] + ps[(1 * GRU_BBW + bme) * GRU_H + u] +
                   ps[(2 * GRU_BBW + bme) * GRU_H + u] + ps[(3 * GRU_BBW + bme) * GRU_H + u];
        __syncthreads();
    }
    if (bvalid) {
        float* bi = bsum_i + (size_t)(b0 + bme) * 1536 + dir * 768;
        float* bh = bsum_h + (size_t)(b0 + bme) * 1536 + dir * 768;
        bi[u] = sum_r; bi[256 + u] = sum_z; bi[512 + u] = sum_n;
        bh[u] = sum_r; bh[256 + u] = sum_z; bh[512 + u] = sum_nr;
    }
}

// ---- from csrc/train_kernels.h ------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// General fp32-MFMA GEMM for the backward pass.
//   C[m][n] (+)= sum_k opA(m,k) * opB(k,n)
//   A_KM = false: A is [M][lda] (k contiguous)      true: A is [K][lda] (m contiguous)
//   B_KN = false: B is [N][ldb] (k contiguous)      true: B is [K][ldb] (n contiguous)
//   split-K over blockIdx.z: slab z covers k in [z*kchunk, (z+1)*kchunk) and is written to
//   C + z*slab_stride (deterministic partial slabs, summed by slab_reduce_kernel), or, with
//   gridDim.z == 1, directly to C.
//   B2/ksplit: rows k >= ksplit of a k-major B come from B2 (two stacked weight matrices).
//   seq/shift (B_KN only): row k of B is taken from row k+shift of the same length-`seq` sequence,
//   zero outside it (the h_{t-1} / h_{t+1} operand of the W_hh gradient).
// 128 x 64 tile, BK = 32, 4 waves 2x2, wave tile 64 x 32.
// ------------------------------------------------------------------------------------------
template <bool A_KM, bool B_KN>
__global__ __launch_bounds__(256) void gemm_general_kernel(const float* __restrict__ A, int lda,
                                                            const float* __restrict__ B, const float* __restrict__ B2,
                                                            int ksplit, int ldb, float* __restrict__ C, int ldc,
                                                            size_t slab_stride, int M, int N, int K, int kchunk, int seq,
                                                            int shift) {
    constexpr int SA = A_KM ? (GB_M + 4) : GB_S;         // LDS row stride of the A tile
    constexpr int SB = B_KN ? (GB_N + 4) : GB_S;
    __shared__ __attribute__((aligned(16))) float As[A_KM ? GB_K * SA : GB_M * SA];
    __shared__ __attribute__((aligned(16))) float Bs[B_KN ? GB_K * SB : GB_N * SB];
    const int m0 = blockIdx.y * GB_M, n0 = blockIdx.x * GB_N;
    const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1, m = lane & 31, kh = lane >> 5;

    float4 ra[4], rb[2];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            if (!A_KM) {
                const int row = idx >> 3, c4 = idx & 7;          // [128 m][8 x float4 of k]
                const int k = k0 + c4 * 4;
                ra[i] = (m0 + row < M && k < kend) ? *reinterpret_cast<const float4*>(A + (size_t)(m0 + row) * lda + k)
                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                const int kr = idx >> 5, c4 = idx & 31;          // [32 k][32 x float4 of m]
                const int k = k0 + kr, mm = m0 + c4 * 4;
                ra[i] = (k < kend && mm < M) ? *reinterpret_cast<const float4*>(A + (size_t)k * lda + mm)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            if (!B_KN) {
                const int row = idx >> 3, c4 = idx & 7;          // [64 n][8 x float4 of k]
                const int k = k0 + c4 * 4;
                rb[i] = (n0 + row < N && k < kend) ? *reinterpret_cast<const float4*>(B + (size_t)(n0 + row) * ldb + k)
                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                const int kr = idx >> 4, c4 = idx & 15;          // [32 k][16 x float4 of n]
                int k = k0 + kr;
                const int nn = n0 + c4 * 4;
                bool ok = k < kend && nn < N;
                const float* src = B;
                if (seq > 0) {
                    const int t = k % seq + shift;
                    ok = ok && t >= 0 && t < seq;
                    k += shift;
                } else if (B2 && k >= ksplit) {
                    src = B2;
                    k -= ksplit;
                }
                rb[i] = ok ? *reinterpret_cast<const float4*>(src + (size_t)k * ldb + nn) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            if (!A_KM) *reinterpret_cast<float4*>(As + (idx >> 3) * SA + (idx & 7) * 4) = ra[i];
            else *reinterpret_cast<float4*>(As + (idx >> 5) * SA + (idx & 31) * 4) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            if (!B_KN) *reinterpret_cast<float4*>(Bs + (idx >> 3) * SB + (idx & 7) * 4) = rb[i];
            else *reinterpret_cast<float4*>(Bs + (idx >> 4) * SB + (idx & 15) * 4) = rb[i];
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;

    if (kbeg < kend) {
        load_tile(kbeg);
        store_tile();
        __syncthreads();
        for (int k0 = kbeg; k0 < kend; k0 += GB_K) {
            const bool more = k0 + GB_K < kend;
            if (more) load_tile(k0 + GB_K);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float a0[4], a1[4], bq[4];
                if (!A_KM) {
                    const float4 v0 = *reinterpret_cast<const float4*>(As + (wm * 64 + m) * SA + kk * 8 + kh * 4);
                    const float4 v1 = *reinterpret_cast<const float4*>(As + (wm * 64 + 32 + m) * SA + kk * 8 + kh * 4);
                    a0[0] = v0.x; a0[1] = v0.y; a0[2] = v0.z; a0[3] = v0.w;
                    a1[0] = v1.x; a1[1] = v1.y; a1[2] = v1.z; a1[3] = v1.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        a0[i] = As[(kk * 8 + kh * 4 + i) * SA + wm * 64 + m];
                        a1[i] = As[(kk * 8 + kh * 4 + i) * SA + wm * 64 + 32 + m];
                    }
                }
                if (!B_KN) {
                    const float4 v = *reinterpret_cast<const float4*>(Bs + (wn * 32 + m) * SB + kk * 8 + kh * 4);
                    bq[0] = v.x; bq[1] = v.y; bq[2] = v.z; bq[3] = v.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) bq[i] = Bs[(kk * 8 + kh * 4 + i) * SB + wn * 32 + m];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], bq[i], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], bq[i], acc[1], 0, 0, 0);
                }
            }
            __syncthreads();
            if (more) {
                store_tile();
                __syncthreads();
            }
        }
    }
    float* Cz = C + (size_t)blockIdx.z * slab_stride;
    const int n = n0 + wn * 32 + m;
    if (n < N) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < M) Cz[(size_t)row * ldc + n] = acc[mt][r];
            }
    }
}

// ---- from csrc/train_kernels.h ------------------------------------------------------------
// out[i] = sum_z slabs[z][i]
static __global__ void slab_reduce_kernel(const float* __restrict__ slabs, size_t slab_stride, int nslab, size_t n,
                                   float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float a = 0.0f;
        for (int z = 0; z < nslab; ++z) a += slabs[(size_t)z * slab_stride + i];
        out[i] = a;
    }
}

// ---- from csrc/train_kernels.h ------------------------------------------------------------
template <bool GRU_IN>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ z, const float* __restrict__ da,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             float2* __restrict__ part, int B, int H, int W, int C, int Hp,
                                                             int Wp, int pix_per_block) {
    // block = (channel group of 64 lanes x 4 pixel lanes); loops over `pix_per_block` pooled pixels
    __shared__ float rs[256], rq[256];
    const int c4n = C / 4;                                // float4 groups per pixel
    const int lanes_c = c4n < 64 ? c4n : 64;              // threads along channels
    const int pl = 256 / lanes_c;                         // pixel lanes
    const int c4 = threadIdx.x % lanes_c, pslot = threadIdx.x / lanes_c;
    const size_t npix = (size_t)B * Hp * Wp;
    const size_t p0 = (size_t)blockIdx.x * pix_per_block;
    float4 sdy = make_float4(0.f, 0.f, 0.f, 0.f), sdx = sdy;
    for (int cc = c4; cc < c4n; cc += lanes_c) {
        const float4 s = *reinterpret_cast<const float4*>(scale + cc * 4), t = *reinterpret_cast<const float4*>(shift + cc * 4);
        const float4 mu = *reinterpret_cast<const float4*>(mean + cc * 4), is = *reinterpret_cast<const float4*>(invstd + cc * 4);
        for (int i = pslot; i < pix_per_block; i += pl) {
            const size_t p = p0 + i;
            if (p >= npix) break;
            const int px = p % Wp, py = (p / Wp) % Hp, b = p / ((size_t)Wp * Hp);
            const float4 g = load_da4<GRU_IN>(da, b, py, px, cc, Hp, Wp, C);
            float4 zz[4], yy[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                zz[q] = *reinterpret_cast<const float4*>(z + (((size_t)b * H + 2 * py + (q >> 1)) * W + 2 * px + (q & 1)) * C + cc * 4);
                yy[q] = make_float4(fmaf(zz[q].x, s.x, t.x), fmaf(zz[q].y, s.y, t.y), fmaf(zz[q].z, s.z, t.z), fmaf(zz[q].w, s.w, t.w));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float dx_ = route1(yy[0].x, yy[1].x, yy[2].x, yy[3].x, q, g.x);
                const float dy_ = route1(yy[0].y, yy[1].y, yy[2].y, yy[3].y, q, g.y);
                const float dz_ = route1(yy[0].z, yy[1].z, yy[2].z, yy[3].z, q, g.z);
                const float dw_ = route1(yy[0].w, yy[1].w, yy[2].w, yy[3].w, q, g.w);
                sdy.x += dx_; sdy.y += dy_; sdy.z += dz_; sdy.w += dw_;
                sdx.x = fmaf(dx_, (zz[q].x - mu.x) * is.x, sdx.x); sdx.y = fmaf(dy_, (zz[q].y - mu.y) * is.y, sdx.y);
                sdx.z = fmaf(dz_, (zz[q].z - mu.z) * is.z, sdx.z); sdx.w = fmaf(dw_, (zz[q].w - mu.w) * is.w, sdx.w);
            }
        }
        // reduce over the pixel lanes for this channel group, component by component
        const float vs[4] = {sdy.x, sdy.y, sdy.z, sdy.w}, vq[4] = {sdx.x, sdx.y, sdx.z, sdx.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __syncthreads();
            rs[threadIdx.x] = vs[e]; rq[threadIdx.x] = vq[e];
            __syncthreads();
            if (pslot == 0) {
                float a = 0.0f, q2 = 0.0f;
                for (int k = 0; k < pl; ++k) { a += rs[k * lanes_c + c4]; q2 += rq[k * lanes_c + c4]; }
                part[(size_t)blockIdx.x * C + cc * 4 + e] = make_float2(a, q2);
            }
        }
        sdy = make_float4(0.f, 0.f, 0.f, 0.f); sdx = sdy;
    }
}

// ---- from csrc/train_kernels.h ------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// data-gradient weights: the dgrad of a 3x3/pad-1 conv is the same conv with the roles of the
// channel axes swapped and the taps flipped.  Output in the wp[g][co'][8] format of
// prep_conv_w_kernel, where co' runs over the forward INPUT channels and the 8-group over the
// forward OUTPUT channels.
// ------------------------------------------------------------------------------------------
static __global__ void prep_conv_wT_kernel(const float* __restrict__ w, float* __restrict__ wp, int cin_f, int cout_f) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = cin_f * 9 * cout_f;
    if (idx >= total) return;
    const int e = idx & 7, cop = (idx >> 3) % cin_f, g = (idx >> 3) / cin_f;
    const int co_f = (g / 9) * 8 + e, tap = 8 - (g % 9);
    wp[idx] = w[((size_t)co_f * cin_f + cop) * 9 + tap];
}

// ---- from csrc/train_kernels.h ------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// conv weight gradient on fp32 MFMA:  dW[co][ci][ky][kx] = sum_{b,y,x} dz[b][y][x][co] * a[b][y+ky-1][x+kx-1][ci]
//   GEMM view: M = co, N = ci (per tap), K = pixels.  576 threads = 9 waves, wave = tap.
//   One workgroup walks RB consecutive rows of one image: the dz row and a 3-row ring of the input
//   (zero halo) sit in LDS; A operand = dz[pixel][co] and B operand = a[pixel+tap][ci] are both
//   read with conflict-free ds_read_b32 (32 consecutive channels).  Per-workgroup partial
//   gradients go to slab[blk][tap][co][ci] and are summed (and transposed to the torch layout) by
//   wgrad_reduce_kernel: deterministic, no atomics.
//   (Tried: dealing the 9 * MT * NT tiles of a k step evenly to 8 / 4 waves instead of wave = tap, to balance the
//   SIMDs -- 9 waves put three on one SIMD.  Correct but 10 % slower: one more ds_read_b32 per MFMA; the kernel is
//   bound by its LDS operand reads, not by the matrix pipes.)
// ------------------------------------------------------------------------------------------
template <int CIN, int COUT>
__global__ __launch_bounds__(576) void conv_wgrad_mfma_kernel(const float* __restrict__ dz, const float* __restrict__ a,
                                                               float* __restrict__ slab, int H, int W, int RB) {
    constexpr int MT = COUT / 32, NT = CIN / 32;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Wk = (W + 1) & ~1;                 // pixels per row rounded up to a k-pair
    float* dzs = lds;                            // [Wk][COUT]
    float* as_ = lds + (size_t)Wk * COUT;        // [3][Wk + 2][CIN]
    const int arow = (Wk + 2) * CIN;
    const int blocks_per_img = H / RB;
    const int b = blockIdx.x / blocks_per_img, y0 = (blockIdx.x % blocks_per_img) * RB;
    const int tid = threadIdx.x, lane = tid & 63, tap = tid >> 6;
    const int ky = tap / 3, kx = tap % 3;
    const int m = lane & 31, kh = lane >> 5;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    auto load_a_row = [&](int y) {               // input row y -> ring slot (y+1)%3, zero halo / out of range
        float* dst = as_ + ((y + 1) % 3) * arow;
        const bool valid = (y >= 0 && y < H);
        const float* src = a + (((size_t)b * H + (valid ? y : 0)) * W) * CIN;
        for (int i = tid; i < (Wk + 2) * (CIN / 4); i += 576) {
            const int px = i / (CIN / 4), c4 = i % (CIN / 4);
            const int gx = px - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid && gx >= 0 && gx < W) v = *reinterpret_cast<const float4*>(src + (size_t)gx * CIN + c4 * 4);
            *reinterpret_cast<float4*>(dst + px * CIN + c4 * 4) = v;
        }
    };
    load_a_row(y0 - 1);
    load_a_row(y0);
    for (int y = y0; y < y0 + RB; ++y) {
        __syncthreads();                          // previous row fully consumed
        load_a_row(y + 1);
        const float* zsrc = dz + (((size_t)b * H + y) * W) * COUT;
        for (int i = tid; i < Wk * (COUT / 4); i += 576) {
            const int px = i / (COUT / 4), c4 = i % (COUT / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (px < W) v = *reinterpret_cast<const float4*>(zsrc + (size_t)px * COUT + c4 * 4);
            *reinterpret_cast<float4*>(dzs + px * COUT + c4 * 4) = v;
        }
        __syncthreads();
        const float* arow_p = as_ + ((y + ky) % 3) * arow;        // input row y + ky - 1 lives in slot (y+ky)%3
        for (int s = 0; s < Wk / 2; ++s) {
            const int px = 2 * s + kh;
            float av[MT], bv[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt] = dzs[px * COUT + mt * 32 + m];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = arow_p[(px + kx) * CIN + nt * 32 + m];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
        }
    }
    float* o = slab + ((size_t)blockIdx.x * 9 + tap) * COUT * CIN;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                o[(size_t)co * CIN + nt * 32 + m] = acc[mt][nt][r];
            }
}

// ---- from csrc/gru_pair_kernel.h ------------------------------------------------------------
// xbuf  [npairs*2 dirs][2 parity][2 halves][GP_BW][128] 8-byte {tag, value} granules, zeroed before every launch
// status[0] is set to 1 if a spin times out (results are then invalid; never happens when all pairs are resident)
template <bool SAVE>
__global__ __launch_bounds__(GP_THREADS) void gru_pair_kernel(
    const float* __restrict__ gi, const float* __restrict__ whh0, const float* __restrict__ whh1,
    const float* __restrict__ bhh0, const float* __restrict__ bhh1, float* __restrict__ y, int B, int S,
    float* __restrict__ gates, float* xbuf, unsigned int* flags, unsigned int* status, int dbg_nowait = 0) {
    extern __shared__ __attribute__((aligned(16))) float plds[];
    gp_f4* wl4 = reinterpret_cast<gp_f4*>(plds);                       // [GP_LDS4][3][threads] float4
    float* hs = plds + GP_LDS4 * 3 * GP_THREADS * 4;                           // h[b][256]
    const int dir = blockIdx.y, pair = blockIdx.x >> 1, half = blockIdx.x & 1;
    const int b0 = pair * GP_BW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int kp = lane & (GP_KP - 1), ul = (wv << 4) | (lane >> 2);      // unit within this half
    const int u = half * GP_UH + ul;                                     // hidden unit (0..255)
    const float* __restrict__ whh = dir ? whh1 : whh0;                   // original [768][256] layout
    const float* __restrict__ bhh = dir ? bhh1 : bhh0;
    const int pd = pair * 2 + dir;
    unsigned long long* xg = reinterpret_cast<unsigned long long*>(xbuf) + (size_t)pd * 2 * 2 * GP_BW * GP_UH;   // granules

    // this thread's weights: gate rows g*256+u, k in [64*kp, 64*kp+64): resident for the whole sequence
    gp_f4 wr[GP_REG4][3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const gp_f4* src = reinterpret_cast<const gp_f4*>(whh + (size_t)(g * 256 + u) * 256 + kp * (GP_K4 * 4));
#pragma unroll
        for (int i = 0; i < GP_REG4; ++i) wr[i][g] = src[i];
#pragma unroll
        for (int i = 0; i < GP_LDS4; ++i) wl4[(i * 3 + g) * GP_THREADS + tid] = src[GP_REG4 + i];
    }
    const float bh_r = bhh[u], bh_z = bhh[256 + u], bh_n = bhh[512 + u];
    for (int i = tid; i < GP_BW * GP_HB; i += GP_THREADS) hs[i] = 0.0f;
    const int bme = kp;                                      // lane kp finishes utterance kp of unit u
    const bool finisher = true;
    const bool bvalid = finisher && (b0 + bme) < B;
    float hprev = 0.0f;
    __syncthreads();

    // gate pre-activations are fetched one step ahead: their HBM/L2 latency hides behind a whole step
    float gr_n = 0.f, gz_n = 0.f, gn_n = 0.f;
    if (bvalid) {
        const float* g = gi + ((size_t)(b0 + bme) * S + (dir ? S - 1 : 0)) * 1536 + dir * 768;
        gr_n = g[u]; gz_n = g[256 + u]; gn_n = g[512 + u];
    }
    for (int step = 0; step < S; ++step) {
        const int t = dir ? (S - 1 - step) : step;
        const float gr = gr_n, gz = gz_n, gn = gn_n;
        if (bvalid && step + 1 < S) {
            const int tn = dir ? (S - 2 - step) : step + 1;
            const float* g = gi + ((size_t)(b0 + bme) * S + tn) * 1536 + dir * 768;
            gr_n = g[u]; gz_n = g[256 + u]; gn_n = g[512 + u];
        }
        gp_f2 acc2[3][GP_BW];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int bb = 0; bb < GP_BW; ++bb) acc2[g][bb] = (gp_f2)(0.0f, 0.0f);
#pragma unroll
        for (int i = 0; i < GP_REG4; ++i) {                      // register-resident weights
            gp_f4 h4[GP_BW];
#pragma unroll
            for (int bb = 0; bb < GP_BW; ++bb) h4[bb] = *reinterpret_cast<const gp_f4*>(hs + bb * GP_HB + kp * GP_HP + i * 4);
            gp_pkfma4(acc2[0], wr[i][0], h4); gp_pkfma4(acc2[1], wr[i][1], h4); gp_pkfma4(acc2[2], wr[i][2], h4);
        }
        // keep the LDS-resident weights IN LDS (do not let the compiler hoist these loop-invariant loads)
        asm volatile("" ::: "memory");
#pragma unroll 2
        for (int i = 0; i < GP_LDS4; ++i) {                      // LDS-resident weights
            gp_f4 h4[GP_BW];
#pragma unroll
            for (int bb = 0; bb < GP_BW; ++bb)
                h4[bb] = *reinterpret_cast<const gp_f4*>(hs + bb * GP_HB + kp * GP_HP + (GP_REG4 + i) * 4);
#pragma unroll
            for (int g = 0; g < 3; ++g) gp_pkfma4(acc2[g], wl4[(i * 3 + g) * GP_THREADS + tid], h4);
        }
        // sum even/odd partial sums, then the 4 k-parts of each unit (lanes differing in bits 0..1)
        float acc[3][GP_BW];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int bb = 0; bb < GP_BW; ++bb) {
                float v = acc2[g][bb].x + acc2[g][bb].y;
                v += gp_quad_xor1(v);          // DPP quad_perm: no LDS crossbar trip
                v += gp_quad_xor2(v);
                acc[g][bb] = v;
            }
        float hr = bh_r, hz = bh_z, hn = bh_n;
#pragma unroll
        for (int bb = 0; bb < GP_BW; ++bb)
            if (bb == bme) { hr += acc[0][bb]; hz += acc[1][bb]; hn += acc[2][bb]; }
        const float r = sigmoidf_(gr + hr);
        const float zg = sigmoidf_(gz + hz);
        const float nn = tanhf(gn + r * hn);
        const float hnew = (1.0f - zg) * nn + zg * hprev;
        // exchange (R2 of the guide's hand-off recipe: the data IS the flag): every value travels as one
        // naturally aligned 8-byte {tag = step+1, value} granule written by ONE write-through store; the
        // consumer re-reads its granule until the tag matches.  One hop instead of store+flag+poll+load.
        unsigned long long* gslot = xg + ((size_t)(step & 1) * 2 + half) * GP_BW * GP_UH;
        const unsigned long long* gpeer = xg + ((size_t)(step & 1) * 2 + (half ^ 1)) * GP_BW * GP_UH;
        hprev = hnew;
        __hip_atomic_store(gslot + bme * GP_UH + ul, ((unsigned long long)(unsigned)(step + 1) << 32) | __float_as_uint(hnew),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bvalid) {
            y[((size_t)(b0 + bme) * S + t) * 512 + dir * 256 + u] = hnew;
            if (SAVE) {
                float* gs = gates + (((size_t)(b0 + bme) * S + t) * 2 + dir) * 1024;
                gs[u] = r; gs[256 + u] = zg; gs[512 + u] = nn; gs[768 + u] = hn;
            }
        }
        __syncthreads();                                                  // every wave is done reading hs
        hs[gp_hidx(bme, u)] = hnew;                                       // own half of the new h
        unsigned long long pv;
        unsigned spins = 0;
        while (((pv = __hip_atomic_load(gpeer + bme * GP_UH + ul, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) !=
               (unsigned long long)(unsigned)(step + 1)) {
            if (dbg_nowait) break;                                        // timing experiment only (wrong results)
            __builtin_amdgcn_s_sleep(1);
            if (++spins > GP_SPIN_LIMIT) { __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        hs[gp_hidx(bme, (half ^ 1) * GP_UH + ul)] = __uint_as_float((unsigned)pv);
        __syncthreads();
    }
}
