// 3x3 convolution as a DIRECT implicit GEMM on the fp16 matrix cores in the f16x3 arithmetic (f16_split.h): M = pixels in 8x4 patches,
// N = output channels, K = (tap, ci).  Companion of the Winograd kernels (conv_wino2_bf16x6_kernel.h): those do 4/9 of the matrix work
// but hand the two-way split SIXTEEN transformed values per 2x2 output tile and channel and are bound by that vector work
// (DESIGN.md "What the profile says to do next"); here every input value is split ONCE while its tile is staged, the matrix pipe does
// 2.25 x the products, and with f16x3 (3 MFMAs per product instead of bf16x6's 6) that is the cheaper side on this chip.
//
// Work split (the register tile is what the bf16x6 direct kernel, conv3x3_bf16x6_ns_kernel, lacked):
//   workgroup = 4 waves, one (8 PR) x (4 PC) pixel tile of one image, all COUT channels;
//   wave      = NT 32-channel slices x MT 32-pixel patches: MT x NT accumulators of 32x32, every A fragment (pixels, from LDS) feeds NT
//               MFMAs and every B fragment (weights, from L2) MT -- at MT x NT = 4 x 2 a tap is 8 ds_read_b128 and 4 global loads for 24
//               MFMAs (768 matrix cycles per SIMD against 256 LDS cycles per CU: the ns kernel's 1 x 4 tile needed 8 reads for 12).
//   LDS image = conv_ns_row_bytes' conflict-free layout, TWO fp16 planes (hi, lo'), double-buffered over the 16-channel chunks: chunk
//               cc + 1 is fetched at tap 0, split and written one staging item per tap behind the MFMAs of chunk cc; one barrier per chunk.
//   weights   = planes [2][g = ci/16*9 + tap][co][16] fp16 (Wh, Wl') from prep_conv_w(T)_f16x3_elem; Wh 2^11 is formed per tap
//               (4 v_pk_mul_f16 per slice).  Terms per product and accumulator: (Al', Wh), (Ah, Wl'), (Ah, Wh 2^11) -- cross terms first;
//               the accumulators run 2^11 too large and the epilogue scales by the exact 2^-11.
// STATUS (round 4): an EXPERIMENT of the A/B harness (bench_conv direct16), not part of libsir_hip.so.  Results equal the bf16x6 direct kernel's to
// 1e-5 on every shape and mode, ragged ones included; timings (profiles/r04/bench_conv_direct_f16x3.txt): 100-105 us for conv3's inference form in
// every tiling tried (2 x 2, 4 x 1, spill-free 4 x 2 in two-wave workgroups) where the Winograd kernel takes 84-87 -- the knock-outs add up instead of
// overlapping (two waves per SIMD, weight fragments only 1-2 taps ahead of an L2 round trip of 1-2 us).
// OUT_MODE 0: BN (folded) + ReLU + 2x2 max-pool, NHWC; 1: the same in the GRU layout [B][Wp][COUT * Hp] (+ the f16x2 planes of the
// following GEMM's operand when `stats` is given); 2: raw output + per-workgroup channel sums / sums of squares in `stats`.
#pragma once
#include "../../speech-intent-recognizer_amd/csrc/bf16x6_kernels.h"
#include "../../speech-intent-recognizer_amd/csrc/conv_wino_bf16x6_kernel.h"   // split_w_f16x3

constexpr size_t conv_d16_lds_bytes(int PR, int PC) { return (size_t)2 * 2 * (8 * PR + 2) * conv_ns_row_bytes(PC); }

// forward weights [cout][cin][3][3] -> planes [2][g][co][16]
__device__ __forceinline__ void prep_conv_w_f16x3_elem(const float* __restrict__ w, unsigned short* __restrict__ wph, int cin, int cout, int idx,
                                                       unsigned int* status) {
    const int total = cin * 9 * cout;
    if (idx >= total) return;
    const int e = idx & 15, co = (idx >> 4) % cout, g = (idx >> 4) / cout;
    split_w_f16x3(w[((size_t)co * cin + (g / 9) * 16 + e) * 9 + g % 9], wph[idx], wph[(size_t)total + idx], status);
}
// data gradient: channel roles swapped, taps flipped (cin_f / cout_f = the FORWARD layer's channels; this launch's outputs are cin_f)
__device__ __forceinline__ void prep_conv_wT_f16x3_elem(const float* __restrict__ w, unsigned short* __restrict__ wph, int cin_f, int cout_f, int idx,
                                                        unsigned int* status) {
    const int total = cin_f * 9 * cout_f;
    if (idx >= total) return;
    const int e = idx & 15, cop = (idx >> 4) % cin_f, g = (idx >> 4) / cin_f;
    split_w_f16x3(w[((size_t)((g / 9) * 16 + e) * cin_f + cop) * 9 + 8 - g % 9], wph[idx], wph[(size_t)total + idx], status);
}
static __global__ void prep_conv_w_f16x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ wph, int cin, int cout, unsigned int* status) {
    prep_conv_w_f16x3_elem(w, wph, cin, cout, blockIdx.x * blockDim.x + threadIdx.x, status);
}
static __global__ void prep_conv_wT_f16x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ wph, int cin_f, int cout_f, unsigned int* status) {
    prep_conv_wT_f16x3_elem(w, wph, cin_f, cout_f, blockIdx.x * blockDim.x + threadIdx.x, status);
}

// KNOCK (timing experiments of devtools/kernel_ab/bench_conv.hip, results invalid; 0 in the product): bit 0 = weights loaded once,
// bit 1 = tile staged once, bit 2 = no MFMAs, bit 3 = no output stores
// NW = waves per workgroup (4, or 2: half the pixels per workgroup with the SAME register tile -- e.g. 16x8 pixels x 128 channels as 2 waves of 4 x 2)
template <int CIN, int COUT, int PR, int PC, int OUT_MODE, int NT = 2, int MINB = 2, int KNOCK = 0, int WR = 3, bool AH2 = true, int NW = 4>
__global__ __launch_bounds__(64 * NW, MINB) void conv3x3_f16x3_direct_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ wph, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ out, int H, int W, int Hp, int Wp, float2* __restrict__ stats) {
    constexpr int BLK = 64 * NW, WN = COUT / (32 * NT), WM = NW / WN, MT = PR * PC / WM, CK = 16, PSB = 16;
    constexpr bool HALF = MT * NT < 8;                      // big register tiles: no half-tile path (hipcc keeps the idle accumulators AND out-of-place copies)
    constexpr int TR = 8 * PR, TC = 4 * PC, TROWS = TR + 2, TCOLS = TC + 2;
    constexpr int RSB = conv_ns_row_bytes(PC), HSB = TCOLS * 16, PLANE = TROWS * RSB, TILEB = 2 * PLANE;
    constexpr int G = (CIN / 16) * 9, NCH = CIN / CK;
    constexpr int NITEMS = TROWS * TCOLS * 4, NIT = (NITEMS + BLK - 1) / BLK;
    static_assert(COUT % (32 * NT) == 0 && WN <= NW && NW % WN == 0 && (PR * PC) % WM == 0 && CIN % CK == 0 && MT >= 1, "tile shape");
    constexpr int SPT = (NIT + 7) / 8, T0 = 9 - (NIT + SPT - 1) / SPT;   // SPT staging items split and written per tap, over the LAST taps of a chunk
    static_assert(MT == 1 || MT % 2 == 0, "half-tile dispatch");
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int b = blockIdx.z, ty0 = blockIdx.y * TR, tx0 = blockIdx.x * TC;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the weight addresses below are SGPR base + one lane offset
    const int wn = wv % WN, wm = wv / WN;
    const int m = lane & 31, h = lane >> 5;
    const int pxl = (m & 1) + 2 * ((m >> 2) & 1);
    const int pyl = ((m >> 1) & 1) + 2 * ((m >> 3) & 1) + 4 * ((m >> 4) & 1);
    // patch pi = wm * MT + mt: row block pi % PR, column block pi / PR (column blocks ascend with mt)
    const int lane_off = pyl * RSB + pxl * PSB + h * HSB;
    int nvalid = 0;                                          // patches of this wave that start inside the image (a prefix)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) nvalid += (tx0 + 4 * ((wm * MT + mt) / PR) < W) ? 1 : 0;
    nvalid = __builtin_amdgcn_readfirstlane(nvalid);
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
    const uint4* wp4 = reinterpret_cast<const uint4*>(wph) + (size_t)(wn * NT * 32) * 2;            // wave-uniform; + ((p * G + g) * COUT + nt * 32) * 2 + lane part
    const int wlane = m * 2 + h;
    const float* xb = x + (size_t)b * H * W * CIN;
    auto load_w = [&](int g, uint4 (&wf)[NT][2]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int p = 0; p < 2; ++p) wf[nt][p] = (wp4 + (((size_t)p * G + g) * COUT + nt * 32) * 2)[wlane];
    };

    // staging items: (pixel, 4-channel part) -> element offset in the image (-1: zero padding), byte offset in the tile (-1: none).  16
    // consecutive lanes = 8 neighbouring pixels x the two 8-byte halves of ONE 16-byte slot class: a ds_write_b64 group covers 128 B.
    int goff[NIT], doff[NIT];
    float4 pre[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int idx = tid + BLK * k;
        const int hsel = idx / (2 * TROWS * TCOLS), rem = idx - hsel * (2 * TROWS * TCOLS);
        const int pix = rem >> 1, part = 2 * hsel + (rem & 1);
        const int tyy = pix / TCOLS, txx = pix - tyy * TCOLS;
        const int gy = ty0 - 1 + tyy, gx = tx0 - 1 + txx;
        const bool item = idx < NITEMS;
        goff[k] = (item && gy >= 0 && gy < H && gx >= 0 && gx < W) ? (gy * W + gx) * CIN + part * 4 : -1;
        doff[k] = item ? tyy * RSB + txx * PSB + (part >> 1) * HSB + (part & 1) * 8 : -1;
    }
    auto stage_load = [&](int cc) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            pre[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (goff[k] >= 0) pre[k] = *reinterpret_cast<const float4*>(xb + goff[k] + cc * CK);
        }
    };
    auto stage_store_item = [&](unsigned char* buf, int k) {
        uint2 hh, ll;
        split2h_quad(pre[k], hh, ll);
        if (doff[k] >= 0) {
            *reinterpret_cast<uint2*>(buf + doff[k]) = hh;
            *reinterpret_cast<uint2*>(buf + doff[k] + PLANE) = ll;
        }
    };

    // weight fragments: a ring of three taps (9 % 3 == 0: static indices across the chunk loop), fetched TWO taps (24-48 MFMAs) ahead --
    // one f16x3 tap of a 2 x 2 register tile is 384 matrix cycles, less than an L2 round trip
    static_assert(WR == 3 || WR == 2, "weight ring");
    uint4 wq[3][NT][2];                                      // WR == 2: slots 0 (current) and 1 (next), copied at the end of a tap
    load_w(0, wq[0]);
    if (WR == 3 && G > 1) load_w(1, wq[1]);
    stage_load(0);
#pragma unroll
    for (int k = 0; k < NIT; ++k) stage_store_item(ldsb, k);
    __syncthreads();
    for (int cc = 0; cc < NCH; ++cc) {
        const unsigned char* tile = ldsb + (cc & 1) * TILEB;
        unsigned char* tile_next = ldsb + ((cc + 1) & 1) * TILEB;
        const bool stage_next = cc + 1 < NCH && !(KNOCK & 2);
        // NP = patches this wave computes: all MT, or only the first half where the rest starts right of the image
        auto taps = [&](auto npc) {
            constexpr int NP = decltype(npc)::value;
            auto rd = [&](int tap, int p, f16x8 (&a)[NP]) {
                const unsigned char* tb = tile + lane_off + (tap / 3) * RSB + (tap % 3) * PSB + p * PLANE;
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    const int pi = wm * MT + i, poff = 8 * (pi % PR) * RSB + 4 * (pi / PR) * PSB;
                    a[i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(tb + poff));
                }
            };
            f16x8 alo[NP], ahi[NP], ahin[NP];
            rd(0, 1, alo);
            rd(0, 0, ahi);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int g = cc * 9 + tap;
                if (tap == 0 && stage_next) stage_load(cc + 1);
                if (!(KNOCK & 1) && g + WR - 1 < G) load_w(g + WR - 1, wq[WR == 3 ? (tap + 2) % 3 : 1]);
                f16x8 wh[NT], wl[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    wh[nt] = __builtin_bit_cast(f16x8, wq[((KNOCK & 1) || WR == 2) ? 0 : tap % 3][nt][0]);
                    wl[nt] = __builtin_bit_cast(f16x8, wq[((KNOCK & 1) || WR == 2) ? 0 : tap % 3][nt][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!(KNOCK & 4)) {
#pragma unroll
                    for (int i = 0; i < NP; ++i)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[i], wh[nt], acc[i][nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (tap < 8) { rd(tap + 1, 1, alo); if (AH2) rd(tap + 1, 0, ahin); }   // the next tap's fragments ride behind the second and third terms
                __builtin_amdgcn_sched_barrier(0);
                if (!(KNOCK & 4)) {
#pragma unroll
                    for (int i = 0; i < NP; ++i)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[i], wl[nt], acc[i][nt], 0, 0, 0);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) wh[nt] = wh[nt] * (_Float16)2048.0f;
                if (!(KNOCK & 4)) {
#pragma unroll
                    for (int i = 0; i < NP; ++i)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[i], wh[nt], acc[i][nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (tap >= T0 && stage_next) {               // the split's vector work rides between the MFMAs, as late as the chunk allows (fetch latency)
#pragma unroll
                    for (int q = 0; q < SPT; ++q)
                        if ((tap - T0) * SPT + q < NIT) stage_store_item(tile_next, (tap - T0) * SPT + q);
                }
                if (tap < 8) {
                    if (AH2) {
#pragma unroll
                        for (int i = 0; i < NP; ++i) ahi[i] = ahin[i];
                    } else rd(tap + 1, 0, ahi);              // free once the third term has issued; the next tap's first term covers the latency
                }
                if (WR == 2 && !(KNOCK & 1) && g + 1 < G) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int p = 0; p < 2; ++p) wq[0][nt][p] = wq[1][nt][p];
                }
            }
        };
        if (2 * nvalid > MT || MT == 1 || (!HALF && nvalid > 0)) taps(std::integral_constant<int, MT>{});
        else if (HALF && nvalid > 0) taps(std::integral_constant<int, (MT + 1) / 2>{});
        else if (stage_next) {                               // a wave without pixels still stages its share
            stage_load(cc + 1);
#pragma unroll
            for (int k = 0; k < NIT; ++k) stage_store_item(tile_next, k);
        }
        if (cc + 1 < NCH) __syncthreads();                   // the other buffer is complete; this one may be overwritten
    }
    if (KNOCK & 8) {
        float sink = 0.0f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) sink += acc[mt][nt][0] + acc[mt][nt][7];
        if (sink == 123.456f) out[0] = sink;
        return;
    }
    if (OUT_MODE == 2) {
        // raw epilogue (training forward / data gradient) + per-workgroup channel statistics
        float ssum[NT], ssq[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { ssum[nt] = 0.0f; ssq[nt] = 0.0f; }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int pi = wm * MT + mt, pr = pi % PR, pc = pi / PR;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int xl = (r & 1) + 2 * h, yl = ((r >> 1) & 1) + 2 * ((r >> 2) & 1) + 4 * ((r >> 3) & 1);
                const int gy = ty0 + 8 * pr + yl, gx = tx0 + 4 * pc + xl;
                if (gy < H && gx < W) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float v = acc[mt][nt][r] * H3_LO_INV;
                        out[(((size_t)b * H + gy) * W + gx) * COUT + (wn * NT + nt) * 32 + m] = v;
                        ssum[nt] += v;
                        ssq[nt] = fmaf(v, v, ssq[nt]);
                    }
                }
            }
        }
        if (stats) {
            float* lds = reinterpret_cast<float*>(ldsb);
            __syncthreads();
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = (wn * NT + nt) * 32 + m;
                const float s = ssum[nt] + __shfl_xor(ssum[nt], 32), q = ssq[nt] + __shfl_xor(ssq[nt], 32);
                if (h == 0) {
                    lds[(wm * COUT + co) * 2] = s;
                    lds[(wm * COUT + co) * 2 + 1] = q;
                }
            }
            __syncthreads();
            const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            for (int c = tid; c < COUT; c += BLK) {
                float s = 0.0f, q = 0.0f;
#pragma unroll
                for (int w4 = 0; w4 < WM; ++w4) { s += lds[(w4 * COUT + c) * 2]; q += lds[(w4 * COUT + c) * 2 + 1]; }
                stats[blk * COUT + c] = make_float2(s, q);
            }
        }
        return;
    }
    // epilogue: BN (folded) -> ReLU -> 2x2 max over registers 4q..4q+3 -> store
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = (wn * NT + nt) * 32 + m;
        const float s = scale[co] * H3_LO_INV, t = shift[co];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int pi = wm * MT + mt, pr = pi % PR, pc = pi / PR;
            const int PX = (tx0 + 4 * pc) / 2 + h;
            const int PYb = (ty0 + 8 * pr) / 2;
            if (PX >= Wp) continue;
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
                const size_t oidx = ((size_t)b * Wp + PX) * (COUT * Hp) + (size_t)co * Hp + PYb;
                float* o = out + oidx;
                if ((Hp & 3) == 0) {
                    const float4 v4 = make_float4(pooled[0], pooled[1], pooled[2], pooled[3]);
                    *reinterpret_cast<float4*>(o) = v4;
                    if (stats) {
                        // OUT_MODE 1 only: `stats` carries the f16x2 plane buffer [2][B * Wp][COUT * Hp] of the following GEMM's A operand
                        unsigned short* planes = reinterpret_cast<unsigned short*>(stats);
                        const size_t plane = (size_t)gridDim.z * Wp * (COUT * Hp);
                        uint2 hh, ll;
                        split2h_quad(v4, hh, ll);
                        *reinterpret_cast<uint2*>(planes + oidx) = hh;
                        *reinterpret_cast<uint2*>(planes + plane + oidx) = ll;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (PYb + q < Hp) o[q] = pooled[q];
                }
            }
        }
    }
}
