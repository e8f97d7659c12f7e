// Convolution weight gradient in Winograd F(2x2, 3x3) form on the bf16 matrix cores (bf16x6 products, fp32 accuracy):
//
//   forward        Y = A^T [ (G g G^T) . (B^T d B) ] A          (d: 4 x 4 input patch of a 2 x 2 output tile)
//   so             dU[xi][co][ci] = sum_tiles (A dY A^T)[xi][co] * (B^T d B)[xi][ci],      dg = G^T dU G
//
// 16 products per tile, output and input channel instead of the 36 of the nine taps (conv_wgrad_bf16x6_kernel): per frequency xi a
// GEMM M = co, N = ci, K = tiles -- the token-reduction shape of gemm_tn2_bf16x6_kernel, and built the same way:
//   * 1024 threads, one workgroup per CU: 8 PRODUCER waves load the dY / input rows of a stage of TPS tiles (all addresses valid:
//     a pixel outside the image is read from the handle's zero page, no load sits under a branch), transform, split into three bf16
//     planes and store k-major images [frequency][plane][tile][channel] into stage buffer s & 1; 8 CONSUMER waves multiply stage
//     s - 1 through the hardware-transposed LDS reads; one bare s_barrier per stage.  A producer thread owns the same items in every
//     stage and re-issues an item's loads for stage s + 1 right after it has consumed those of stage s: one register set, a whole
//     stage of latency cover.
//   * a workgroup handles ONE ROW i of the 4 x 4 frequency grid (blockIdx -> (strip of stages, i)): (A dY A^T)[i][.] needs the row
//     combination alpha y0 + beta y1 of the tile's two dY rows, (B^T d B)[i][.] the combination ca d_ra + cb d_rb of two input
//     rows, then the four column combinations -- the four kinds together do the transform work of one pass, and their accumulators
//     (4 frequencies x COUT x CIN fp32 = 128 KB for 64 -> 128) fit the consumers' registers.  The four kinds of a strip run on the
//     same XCD (they read the same rows).
//   * every workgroup writes one slab [4 j][co][ci]; wgrad_wino_sum_kernel adds the strips in a fixed order and
//     wgrad_wino_finish_kernel applies G^T . G and writes the torch layout.  Deterministic.
#pragma once
#include "gemm_tn2_bf16x6_kernel.h"

constexpr int WGW_THREADS = 1024;
// KPW = rows of the frequency grid per workgroup: 1 for 64 -> 128 (4 frequencies x 128 x 64 accumulators fill the consumers' registers),
// 2 for 32 -> 64 (8 x 64 x 32: a quarter of that) -- the dY rows are then loaded once for two rows and the input patch needs 3 rows
// for two instead of 2 for one: 0.6x the bytes per tile and row, and the 32 -> 64 kernel is bound by what a CU can pull in.
template <int CIN, int COUT> struct WgwCfg {
    static constexpr int KPW = COUT >= 128 ? 1 : 2;
    static constexpr int NF = 4 * KPW;                                    // frequencies per workgroup
    static constexpr int TPS = 16;                                        // tiles per stage (K of one MFMA step)
    static constexpr int ZW = COUT * 2, AW = CIN * 2;                     // row bytes of the P / V images (one plane)
    static constexpr int PPLANE = TPS * ZW, VPLANE = TPS * AW;
    static constexpr int STAGE = NF * 3 * (PPLANE + VPLANE);
    static constexpr size_t lds_bytes = 2 * (size_t)STAGE;                // 147,456 B for both shapes
    static constexpr int groups = 4 / KPW;                                // workgroups per strip
};
// `num_cus`: CUs of the device (sir_handle::num_cus); at most 256 workgroups in all, which is what the slab plan of
// model_train.hip (64 strips of conv3, 128 of conv2) is sized for
inline int wgrad_wino_strips(int B, int H, int W, int tps, int groups, int num_cus = 256) {
    const long long ntiles = (long long)B * (H / 2) * ((W + 1) / 2), nst = (ntiles + tps - 1) / tps;
    const int cap = (num_cus < 256 ? (num_cus < groups ? groups : num_cus) : 256) / groups;     // one workgroup per CU
    int s = nst < cap ? (int)nst : cap;
    if (s >= 8) s &= ~7;                                                    // multiples of 8: the XCD-aware order below
    return s < 1 ? 1 : s;
}

template <int XW>
__device__ __forceinline__ int wgw_off(int k, int xb) { return XW == 64 ? k * 64 + xb : tn_kmaj_off<(XW == 64 ? 128 : XW)>(k, xb); }

__device__ __forceinline__ float4 wgw_lin(float a, const float4& x, float b, const float4& y) {
    return make_float4(a * x.x + b * y.x, a * x.y + b * y.y, a * x.z + b * y.z, a * x.w + b * y.w);
}
__device__ __forceinline__ float4 wgw_add(const float4& x, const float4& y) { return make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w); }
__device__ __forceinline__ float4 wgw_sub(const float4& x, const float4& y) { return make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w); }
__device__ __forceinline__ float4 wgw_neg(const float4& x) { return make_float4(-x.x, -x.y, -x.z, -x.w); }

// dz: [B][H][W][COUT] (gradient of the raw conv output), a: [B][H][W][CIN] (the layer input), slab: [strips][4 i][4 j][COUT][CIN]
// F16 (round 4): the products on the fp16 matrix cores (f16_split.h), as in gemm_tn2_bf16x6_kernel's F16 form: the transformed output
// gradient P (it carries the backward's loss scale) as two planes (Ph, Pl'), the transformed input V scaled by 2^-5 (|V| <= 4 max |input|:
// exact below |input| = 256, clamped beyond) as two as well (Vh, Vl'; the consumers form Vh 2^11 in registers); Pl' Vh + Ph Vl' +
// Ph (Vh 2^11) into the one accumulator set, 2^-11 x 2^5 in the epilogue.  The frequency stride of the LDS images stays three planes.
constexpr int WGW_VS_LOG2 = -5;
template <int CIN, int COUT, bool F16 = false>
__global__ __launch_bounds__(WGW_THREADS, WGW_THREADS / 256) void conv_wgrad_wino_bf16x6_kernel(
    const float* __restrict__ dz, const float* __restrict__ a, float* __restrict__ slab, int B, int H, int W) {
    using C = WgwCfg<CIN, COUT>;
    constexpr int KPW = C::KPW, NF = C::NF, TPS = C::TPS, ZW = C::ZW, AW = C::AW, PPLANE = C::PPLANE, VPLANE = C::VPLANE, STAGE = C::STAGE;
    constexpr int WPF = 8 / NF;                                 // consumer waves per frequency (2 or 1)
    constexpr int MTW = COUT / 32 / WPF, NT = CIN / 32;         // 32 x 32 accumulators per consumer wave: MTW x NT
    constexpr int PQ = COUT / 4, VQ = CIN / 4;                  // channel quads per tile
    // producer items: KPW = 1: a dY item (4 loads -> 4 frequencies) on every thread, an input item (8 loads -> 4) on waves 0-3;
    //                 KPW = 2: a dY item (4 loads -> 8 frequencies) on waves 0-3, an input item of ONE row (8 loads -> 4) on waves 4-7
    static_assert(TPS * PQ == (KPW == 1 ? 512 : 256) && TPS * VQ * KPW == 256, "item counts");
    extern __shared__ __attribute__((aligned(16))) unsigned char wgl[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifndef SIR_W2_CLAMP
    if (F16) sir_fp16_ovfl_on();                             // producers: unclamped splits; consumers: Vh 2^11 saturates instead of overflowing
#endif
    const int TH = H >> 1, TW = (W + 1) >> 1;
    const int ntiles = B * TH * TW, nstages = (ntiles + TPS - 1) / TPS;
    // blockIdx -> (strip, group of rows): the groups of a strip on one XCD (workgroup L runs on XCD L % 8) when the strips come in eights
    constexpr int NG = C::groups;
    const int nstrips = gridDim.x / NG;
    int strip, grp;
    if ((nstrips & 7) == 0) { strip = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * NG)); grp = (blockIdx.x >> 3) % NG; }
    else { strip = blockIdx.x / NG; grp = blockIdx.x % NG; }
    const int s_begin = (int)((long long)nstages * strip / nstrips), s_end = (int)((long long)nstages * (strip + 1) / nstrips);
    const int nst = s_end - s_begin;

    if (wv < 8) {
        // ================= producers ==============================================================================================
        __builtin_amdgcn_s_setprio(3);
        // row i of A (4 x 2): (1,0), (1,1), (1,-1), (0,-1); row i of B^T (4 x 4): d0 - d2, d1 + d2, d2 - d1, d1 - d3
        auto a_row = [](int i, float& al, float& be) { al = i == 3 ? 0.0f : 1.0f; be = i == 0 ? 0.0f : (i == 1 ? 1.0f : -1.0f); };
        const bool p_role = KPW == 1 || wv < 4, v_role = KPW == 1 ? wv < 4 : wv >= 4;
        const int vi = KPW == 1 ? grp : 2 * grp + ((wv - 4) >> 1);             // the frequency row of this wave's input items
        const int ra = vi == 0 ? 0 : 1, rb = vi == 3 ? 3 : 2;
        const float ca = vi == 2 ? -1.0f : 1.0f, cb = (vi == 1 || vi == 2) ? 1.0f : -1.0f;
        const float inv_tw = 1.0f / (float)TW, inv_th = 1.0f / (float)TH;
        const int pk = tid / PQ, pq = tid % PQ;                  // dY item: tile pk of the stage, channel quad pq
        const int vt = KPW == 1 ? tid : (tid & 127);             // input item index inside its row
        const int vk = vt / VQ, vq = vt % VQ;
        auto coords = [&](int tile, int& b, int& ty, int& tx) { // tile -> (image, tile row, tile column); tile < 2^24
            int row = (int)((float)tile * inv_tw);
            int r = tile - row * TW;
            row += (r >= TW) - (r < 0);
            r = tile - row * TW;
            int bb = (int)((float)row * inv_th);
            int q = row - bb * TH;
            bb += (q >= TH) - (q < 0);
            b = bb; ty = row - bb * TH; tx = r;
        };
        // Buffer loads: 32-bit byte offsets against one descriptor per tensor, and the hardware range check returns zeros for
        // an offset past the end -- a pixel outside the image (or a tile past the last one) gets offset 2^31 (the tensors are smaller:
        // checked by the host) instead of a pointer select against a zero page: 1-2 VALU instructions per load instead of 5-6.
        const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dz), 0, B * H * W * COUT * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a), 0, B * H * W * CIN * 4, 0x00020000);
        auto ld = [&](const __amdgpu_buffer_rsrc_t r, unsigned off) {
            return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
        };
        auto fetch_p = [&](int stage, float4 (&y)[2][2]) {
            const int tile = stage * TPS + pk;
            int b, ty, tx;
            coords(tile, b, ty, tx);
            const unsigned base = tile < ntiles ? (unsigned)(((b * H + 2 * ty) * W + 2 * tx) * COUT + 4 * pq) * 4u : 0x80000000u;
            const unsigned rowb = (unsigned)W * COUT * 4u;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                // (a group of ONE row that uses one dY row only -- rows 0 and 3: alpha or beta = 0 -- does not fetch the other)
                const bool used = KPW == 2 || !((r == 1 && grp == 0) || (r == 0 && grp == 3));
                y[r][0] = ld(rz, used ? base + r * rowb : 0x80000000u);
                y[r][1] = ld(rz, used && 2 * tx + 1 < W ? base + r * rowb + COUT * 4u : 0x80000000u);
            }
        };
        auto fetch_v = [&](int stage, float4 (&d)[2][4]) {
            const int tile = stage * TPS + vk;
            int b, ty, tx;
            coords(tile, b, ty, tx);
            const bool tv = tile < ntiles;
            const unsigned rowb = (unsigned)W * CIN * 4u;
            const unsigned base = (unsigned)(((b * H + 2 * ty - 1) * W + 2 * tx - 1) * CIN + 4 * vq) * 4u;      // patch corner (may lie outside: never used then)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int rr = r ? rb : ra, row = 2 * ty - 1 + rr;
                const bool rv = tv && row >= 0 && row < H;
                const unsigned rbase = base + rr * rowb;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int col = 2 * tx - 1 + c;
                    d[r][c] = ld(ra_, rv && col >= 0 && col < W ? rbase + c * CIN * 4u : 0x80000000u);
                }
            }
        };
        auto put = [&](unsigned char* img, int plane, const float4& v) {           // img: the item's address in a frequency's first plane
            uint2 hh, mm, ll;
            if constexpr (F16) {
#ifdef SIR_W2_CLAMP
                if (plane == PPLANE) split2h_quad(v, hh, ll);                        // P (gradient side): (Ph, Pl')
                else tn2_split_b<WGW_VS_LOG2>(v, hh, ll);                            // V: (Vh, Vl') of V / 32
#else                                                        // (the kernel runs with MODE.FP16_OVFL = 1: no clamps, see tn2_split_b_ovfl)
                if (plane == PPLANE) split2h_quad_ovfl(v, hh, ll);
                else tn2_split_b_ovfl<WGW_VS_LOG2>(v, hh, ll);
#endif
                *reinterpret_cast<uint2*>(img) = hh;
                *reinterpret_cast<uint2*>(img + plane) = ll;
                return;
            } else {
                split3_quad(v, hh, mm, ll);
            }
            *reinterpret_cast<uint2*>(img) = hh;
            *reinterpret_cast<uint2*>(img + plane) = mm;
            *reinterpret_cast<uint2*>(img + 2 * plane) = ll;
        };
        auto stage_p = [&](unsigned char* buf, const float4 (&y)[2][2]) {
#pragma unroll
            for (int il = 0; il < KPW; ++il) {
                float al, be;
                a_row(KPW * grp + il, al, be);
                const float4 p0 = wgw_lin(al, y[0][0], be, y[1][0]), p1 = wgw_lin(al, y[0][1], be, y[1][1]);
                unsigned char* img = buf + il * 12 * PPLANE + wgw_off<ZW>(pk, 8 * pq);
                put(img, PPLANE, p0);
                put(img + 3 * PPLANE, PPLANE, wgw_add(p0, p1));
                put(img + 6 * PPLANE, PPLANE, wgw_sub(p0, p1));
                put(img + 9 * PPLANE, PPLANE, wgw_neg(p1));
            }
        };
        auto stage_v = [&](unsigned char* buf, const float4 (&d)[2][4]) {
            float4 v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = wgw_lin(ca, d[0][c], cb, d[1][c]);
            unsigned char* img = buf + NF * 3 * PPLANE + (vi - KPW * grp) * 12 * VPLANE + wgw_off<AW>(vk, 8 * vq);
            put(img, VPLANE, wgw_sub(v[0], v[2]));
            put(img + 3 * VPLANE, VPLANE, wgw_add(v[1], v[2]));
            put(img + 6 * VPLANE, VPLANE, wgw_sub(v[2], v[1]));
            put(img + 9 * VPLANE, VPLANE, wgw_sub(v[1], v[3]));
        };
        // Loads run TWO stages ahead where the registers allow it (two sets: an item's loads for stage s + 2 are issued right after
        // stage s has consumed the set): with one set the loads of stage s + 1 are issued at the END of step s and only the barrier
        // wait covers them -- one stage of bytes in flight per CU is what the 32 -> 64 kernel's 33 GB/s per CU came from.  Steps in
        // pairs; the consumers add the odd barrier.  (64 -> 128: waves 0-3 carry a dY item AND an input item, 48 registers per set;
        // two sets there cost the consumers' accumulators their room, so those waves keep one.)
        const int npair = nst / 2 + 1;
        if (p_role && v_role) {
            float4 y[2][2], d[2][4];
            fetch_p(s_begin, y);
            fetch_v(s_begin, d);
#pragma unroll 1
            for (int s = 0; s < 2 * npair; ++s) {
                unsigned char* buf = wgl + (s & 1) * STAGE;
                stage_p(buf, y);
                fetch_p(s_begin + s + 1, y);                     // (past the strip's end: valid or range-checked addresses, never used)
                stage_v(buf, d);
                fetch_v(s_begin + s + 1, d);
                tn2_barrier();
            }
        } else if (p_role) {
            float4 y0[2][2], y1[2][2];
            fetch_p(s_begin, y0);
            fetch_p(s_begin + 1, y1);
#pragma unroll 1
            for (int s = 0; s < 2 * npair; s += 2) {
                stage_p(wgl, y0);
                fetch_p(s_begin + s + 2, y0);
                tn2_barrier();
                stage_p(wgl + STAGE, y1);
                fetch_p(s_begin + s + 3, y1);
                tn2_barrier();
            }
        } else {
            float4 d0[2][4], d1[2][4];
            fetch_v(s_begin, d0);
            fetch_v(s_begin + 1, d1);
#pragma unroll 1
            for (int s = 0; s < 2 * npair; s += 2) {
                stage_v(wgl, d0);
                fetch_v(s_begin + s + 2, d0);
                tn2_barrier();
                stage_v(wgl + STAGE, d1);
                fetch_v(s_begin + s + 3, d1);
                tn2_barrier();
            }
        }
        return;
    }

    // ================= consumers ==================================================================================================
    const int cw = wv - 8, f = cw / WPF, mh = cw % WPF, i32 = lane & 31, kgrp = lane >> 5;      // frequency f = (row inside the group) * 4 + j
    f32x16 acc[MTW][NT];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
    const int tq = (lane >> 2) & 3, tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3), tk = 8 * kgrp + tq;
    int poff[MTW], voff[NT];
#pragma unroll
    for (int m = 0; m < MTW; ++m) poff[m] = f * 3 * PPLANE + wgw_off<ZW>(tk, 2 * ((mh * MTW + m) * 32 + tcol));
#pragma unroll
    for (int n = 0; n < NT; ++n) voff[n] = NF * 3 * PPLANE + f * 3 * VPLANE + wgw_off<AW>(tk, 2 * (n * 32 + tcol));
    tn2_barrier();                                               // step 0: the producers write stage 0
#pragma unroll 1
    for (int s = 1; s <= nst; ++s) {
        const unsigned char* sb = wgl + ((s - 1) & 1) * STAGE;
#pragma unroll
        for (int ks = 0; ks < TPS / 16; ++ks) {
            constexpr int NPA = F16 ? 2 : 3;
            bf16x8 af[MTW][NPA], bf[NT][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                if (p < NPA) {
#pragma unroll
                    for (int m = 0; m < MTW; ++m) af[m][p < NPA ? p : 0] = tn_tr_fragment<ZW>(sb + poff[m] + p * PPLANE + ks * 16 * ZW);
                }
                if (p < NPA) {
#pragma unroll
                    for (int n = 0; n < NT; ++n) bf[n][F16 && p == 1 ? 2 : p] = tn_tr_fragment<AW>(sb + voff[n] + p * VPLANE + ks * 16 * AW);
                }
            }
            if constexpr (F16) {
#pragma unroll
                for (int n = 0; n < NT; ++n) bf[n][1] = tn2_hi2(bf[n][0]);
                constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 2, 1};       // Pl' Vh, Ph Vl', Ph (Vh 2^11)
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[m][HA[t3]]), __builtin_bit_cast(f16x8, bf[n][HB[t3]]),
                                                                              acc[m][n], 0, 0, 0);
            } else {
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // small terms first
#pragma unroll
                for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][PA[t6] % NPA], bf[n][PB[t6]], acc[m][n], 0, 0, 0);
            }
        }
        tn2_barrier();
    }
    if (!(nst & 1)) tn2_barrier();                               // the odd step of the producers' last pair
    float* out = slab + ((size_t)strip * 16 + (KPW * grp) * 4 + f) * COUT * CIN;     // [strip][i][j]: i = KPW grp + f / 4, j = f % 4
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (mh * MTW + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kgrp, ci = n * 32 + i32;
                out[(size_t)co * CIN + ci] = acc[m][n][r] * (F16 ? H3_LO_INV * (float)(1 << (-WGW_VS_LOG2)) : 1.0f);
            }
}

// dU[e] = sum_strips slab[strip][e] (fixed order), e over [4 i][4 j][co][ci]; float4 per thread
static __global__ __launch_bounds__(256) void wgrad_wino_sum_kernel(const float* __restrict__ slab, int nstrips, int total4, float* __restrict__ du) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total4) return;
    const float4* s4 = reinterpret_cast<const float4*>(slab);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 8 <= nstrips; k += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = s4[(size_t)(k + u) * total4 + idx];
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; k < nstrips; ++k) {
        const float4 v = s4[(size_t)k * total4 + idx];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4*>(du)[idx] = acc;
}

// dw[co][ci][ky][kx] = (G^T dU G)[ky][kx],  dU: [4 i][4 j][co][ci];  G = (1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1)
static __global__ __launch_bounds__(256) void wgrad_wino_finish_kernel(const float* __restrict__ du, int cin, int cout, float* __restrict__ dw,
                                                                       float unscale) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // (co, ci), ci fastest
    const int n = cout * cin;
    if (idx >= n) return;
    float u[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) u[i][jj] = du[(size_t)(i * 4 + jj) * n + idx];
    float t[3][4];                                               // G^T dU
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        t[0][jj] = u[0][jj] + 0.5f * (u[1][jj] + u[2][jj]);
        t[1][jj] = 0.5f * (u[1][jj] - u[2][jj]);
        t[2][jj] = 0.5f * (u[1][jj] + u[2][jj]) + u[3][jj];
    }
    float* o = dw + (size_t)idx * 9;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        o[ky * 3 + 0] = (t[ky][0] + 0.5f * (t[ky][1] + t[ky][2])) * unscale;     // (the backward's loss scale leaves here: a power of two)
        o[ky * 3 + 1] = (0.5f * (t[ky][1] - t[ky][2])) * unscale;
        o[ky * 3 + 2] = (0.5f * (t[ky][1] + t[ky][2]) + t[ky][3]) * unscale;
    }
}
