// Superseded kernel generations, kept OUT of libsir_hip.so: the A/B harnesses (devtools/kernel_ab/bench_gemm.hip, devtools/kernel_ab/bench_conv.hip)
// include this header to time them against the product kernels and to compare results.  Nothing under csrc/ references them.
// Round 4: only the two generations a harness mode still compares against are left -- the first bf16x6 GEMM (bench_gemm's "gen1" line)
// and the first bf16x6 convolution (bench_conv's "wave owns all channels" line); the fp32-MFMA kernels, the VALU conv1, the streaming /
// paired fp32 GRU recurrences, the L2-streaming BPTT kernel and the other round-1 forms are in the history (git show fb5b5cd:devtools/kernel_ab/legacy_kernels.h).
#pragma once
#include "../csrc/bf16x6_kernels.h"

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

