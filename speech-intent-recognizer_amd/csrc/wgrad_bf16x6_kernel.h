// Convolution weight gradient on the bf16 matrix cores (bf16x6 products, fp32 accuracy).
//
//   dW[co][ci][ky][kx] = sum_{b,y,x} dz[b][y][x][co] * a[b][y+ky-1][x+kx-1][ci]
//   GEMM view per tap: M = co, N = ci, K = pixels of a row.  v_mfma_f32_32x32x16_bf16 takes 16 pixels per instruction but
//   wants, per lane, EIGHT CONSECUTIVE PIXELS of one channel -- the transpose of the NHWC activations.  The LDS hardware
//   transposes on the way out: a row is staged in its natural [pixel][channel] order (thread = (pixel, 4 channels) reads a
//   float4 -- lanes along the channels: coalesced --, splits it into the three bf16 planes and issues one ds_write_b64 per
//   plane) into dzP[plane][pixel][co] / aP[ring row][plane][pixel + 1][ci], and the fragments come from
//   ds_read_b64_tr_b16 (lane i gets column i of a 4-pixel x 16-channel block; two reads = the lane's eight pixels).
//   Tap shift kx: with pixel-major rows the B fragment of tap kx simply starts kx rows later.  (The first version wrote
//   channel-major rows with twelve 2-byte stores per float4 and shifted with v_alignbit: timing knock-outs put its
//   staging + fetch at half of the kernel.)
//   8 waves: wave = (mt, nt, ks); every wave accumulates ALL NINE taps of its 32 x 32 (co, ci) tile (9 accumulators)
//   and takes the k-steps s = ks (mod KSPLIT) of a row, KSPLIT = 8 / (MT * NT) (1 for 64->128, 4 for 32->64); each
//   workgroup adds its KSPLIT partials in LDS and writes ONE slab [tap][co][ci] -- deterministic two-pass reduce afterwards.
#pragma once
#include "bf16x6_kernels.h"

#include <type_traits>
#include "gemm_tn_bf16x6_kernel.h"      // tn_kmaj_off / tn_tr_fragment: the swizzled k-major LDS image and its transposed reads

constexpr int wgrad_x6_kpx(int W) { return (W + 15) / 16 * 16; }
constexpr int wgrad_x6_arows(int W) { return wgrad_x6_kpx(W) + 8; }                   // pixel rows of one a-image: + halo and shifted reads
inline size_t wgrad_x6_lds_bytes(int cin, int cout, int W) {
    return (size_t)3 * wgrad_x6_kpx(W) * cout * 2 + (size_t)9 * wgrad_x6_arows(W) * cin * 2;
}
constexpr int wgrad_x6_ksplit(int cin, int cout) { return 8 / ((cout / 32) * (cin / 32)); }   // waves sharing a (co, ci) tile; reduced in the workgroup
constexpr int wgrad_x6_max_w(int cout) { return cout >= 128 ? 64 : 128; }   // row widths the register prefetch is sized for

template <int CIN, int COUT>
__global__ __launch_bounds__(512) void conv_wgrad_bf16x6_kernel(const float* __restrict__ dz, const float* __restrict__ a,
                                                                 float* __restrict__ slab, int H, int W, int RB) {
    constexpr int MT = COUT / 32, NT = CIN / 32, KSPLIT = 8 / (MT * NT);
    constexpr int ZW = COUT * 2, AW = CIN * 2;                 // bytes per pixel row of the dz / a images (one plane)
    static_assert(MT * NT * KSPLIT == 8, "8 waves");
    static_assert(ZW == 128 || ZW == 256 || ZW == 512, "swizzle rule");
    static_assert(AW == 64 || AW == 128 || AW == 256, "swizzle rule");
    extern __shared__ __attribute__((aligned(16))) unsigned char wl[];
    const int kpx = (W + 15) / 16 * 16, arows = kpx + 8;
    const int zplane = kpx * ZW, aplane = arows * AW;
    unsigned char* dzP = wl;                                   // [3][kpx][COUT]
    unsigned char* aP = wl + (size_t)3 * zplane;               // [3 ring][3][arows][CIN], pixel gx at row gx + 1
    const size_t lds_bytes = (size_t)3 * zplane + (size_t)9 * aplane;
    const int blocks_per_img = H / RB;
    const int b = blockIdx.x / blocks_per_img, y0 = (blockIdx.x % blocks_per_img) * RB;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int mt = wv / (NT * KSPLIT), nt = (wv / KSPLIT) % NT, ks = wv % KSPLIT;
    const int i32 = lane & 31, kgrp = lane >> 5;

    for (int i = tid; i < (int)(lds_bytes / 16); i += 512) reinterpret_cast<uint4*>(wl)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    // byte offset of (pixel row r, channel byte xb) inside one plane; AW = 64: a row IS one 64-byte chunk, and four
    // consecutive rows already cover the four quarters of the bank line
    auto zoff = [](int r, int xb) { return tn_kmaj_off<ZW>(r, xb); };
    auto aoff = [](int r, int xb) { return AW == 64 ? r * 64 + xb : tn_kmaj_off<(AW == 64 ? 128 : AW)>(r, xb); };

    // A row is staged in two halves so that its global-load latency hides behind the previous row's MFMAs:
    //   fetch_row : float4 loads of the thread's items into registers,
    //   write_row : split into the three bf16 planes, one 8-byte store per plane at [pixel + poff][4 channels].
    // Staging items: thread = (channel group c4 = tid % (C / 4), pixel = tid / (C / 4) + (2048 / C) k): lanes run along the
    // channels (a pixel's channels are contiguous in NHWC: coalesced loads, and a 16-lane store group covers contiguous LDS
    // bytes), all index arithmetic is shifts.  (Items numbered it = tid + 512 k with it % W / it / W kept one 64-bit offset
    // per item alive; those spilled, and every reload -- a scratch load, counted in vmcnt -- put an s_waitcnt vmcnt(0) in
    // front of the next global load: the row's six loads were issued one at a time.)
    constexpr int MAXW = wgrad_x6_max_w(COUT);                 // the host checks W <= MAXW (the LDS image bounds it anyway)
    constexpr int NZ = MAXW / (2048 / COUT), NA = MAXW / (2048 / CIN);     // float4 items per thread
    auto fetch_row = [&](const float* src, auto cc, auto& regs) {
        constexpr int C = decltype(cc)::value;
        const int c4 = tid % (C / 4), px0 = tid / (C / 4);
#pragma unroll
        for (int k = 0; k < (int)(sizeof(regs) / sizeof(float4)); ++k) {
            const int px = px0 + (2048 / C) * k;
            regs[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (src && px < W) regs[k] = *reinterpret_cast<const float4*>(src + px * C + 4 * c4);
        }
    };
    auto write_row = [&](const auto& regs, auto cc, unsigned char* base, int plane, int poff, auto off) {
        constexpr int C = decltype(cc)::value;
        const int c4 = tid % (C / 4), px0 = tid / (C / 4);
#pragma unroll
        for (int k = 0; k < (int)(sizeof(regs) / sizeof(float4)); ++k) {
            const int px = px0 + (2048 / C) * k;
            if (px >= W) continue;
            uint2 hh, mm, ll;
            split3_quad(regs[k], hh, mm, ll);
            unsigned char* d = base + off(px + poff, 8 * c4);
            *reinterpret_cast<uint2*>(d) = hh;
            *reinterpret_cast<uint2*>(d + plane) = mm;
            *reinterpret_cast<uint2*>(d + 2 * plane) = ll;
        }
    };
    using CI = std::integral_constant<int, CIN>;
    using CO = std::integral_constant<int, COUT>;
    auto a_src = [&](int y) -> const float* { return (y >= 0 && y < H) ? a + (((size_t)b * H + y) * W) * CIN : nullptr; };
    auto a_slot = [&](int y) { return aP + (size_t)((y + 1) % 3) * 3 * aplane; };            // input row y -> ring slot (y + 1) % 3

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    float4 pz[NZ], pa[NA];
    fetch_row(a_src(y0 - 1), CI{}, pa);
    write_row(pa, CI{}, a_slot(y0 - 1), aplane, 1, aoff);
    fetch_row(a_src(y0), CI{}, pa);
    write_row(pa, CI{}, a_slot(y0), aplane, 1, aoff);
    fetch_row(a_src(y0 + 1), CI{}, pa);                        // rows y0 + 1 (a) and y0 (dz) are in flight
    fetch_row(dz + (((size_t)b * H + y0) * W) * COUT, CO{}, pz);
    const int nks = kpx / 16;
    // transposed-read addresses (see gemm_tn_bf16x6_kernel): lane 4 q + p of a 16-lane group supplies row q, channels 4 p .. 4 p + 3
    const int tq = (lane >> 2) & 3, tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3), tk = 8 * kgrp + tq;
    const int zlane = zoff(tk, 2 * (mt * 32 + tcol));
    int alane[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) alane[kx] = aoff(tk + kx, 2 * (nt * 32 + tcol));       // pixel gx + kx - 1 sits at row gx + kx
    for (int y = y0; y < y0 + RB; ++y) {
        __syncthreads();                                       // previous row fully consumed
        write_row(pa, CI{}, a_slot(y + 1), aplane, 1, aoff);
        write_row(pz, CO{}, dzP, zplane, 0, zoff);
        __syncthreads();
        // The next row's loads are issued AFTER the barrier: __syncthreads() drains vmcnt, so loads issued in front of it
        // would be waited for right there; issued here they fly during this row's MFMAs and are only waited for at the
        // next row's first barrier.
        if (y + 1 < y0 + RB) {
            fetch_row(a_src(y + 2), CI{}, pa);
            fetch_row(dz + (((size_t)b * H + y + 1) * W) * COUT, CO{}, pz);
        }
        for (int s = ks; s < nks; s += KSPLIT) {
            bf16x8 afr[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) afr[p] = tn_tr_fragment<ZW>(dzP + p * zplane + zlane + s * 16 * ZW);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const unsigned char* ring = aP + (size_t)((y + ky) % 3) * 3 * aplane + s * 16 * AW;
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // small terms first
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    // one tap at a time: three plane fragments live instead of nine (a chain of v_mfma_f32_32x32x16_bf16 on ONE
                    // accumulator issues at the same 32 cycles per instruction as interleaved chains)
                    bf16x8 bfr[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) bfr[p] = tn_tr_fragment<AW>(ring + p * aplane + alane[kx]);
#pragma unroll
                    for (int t6 = 0; t6 < 6; ++t6)
                        acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[PA[t6]], bfr[PB[t6]], acc[ky * 3 + kx], 0, 0, 0);
                }
            }
        }
    }
    // The KSPLIT waves of a (co, ci) tile hold partial sums over disjoint k-steps: they are added inside the workgroup (tap by
    // tap through LDS, fixed order ks = 0, 1, ... -> bit-reproducible) so that ONE slab per workgroup leaves the chip
    // instead of KSPLIT (conv2: 19 MB of slab writes per launch instead of 75 MB, and a reduce kernel that reads a quarter).
    if (KSPLIT > 1) {
        float* red = reinterpret_cast<float*>(wl);             // [KSPLIT - 1][MT * NT][16][64]
        const int tile = mt * NT + nt;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            __syncthreads();                                   // the LDS images / the previous tap's partials are consumed
            if (ks > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(((ks - 1) * (MT * NT) + tile) * 16 + r) * 64 + lane] = acc[tap][r];
            }
            __syncthreads();
            if (ks == 0) {
#pragma unroll
                for (int k2 = 1; k2 < KSPLIT; ++k2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tap][r] += red[(((k2 - 1) * (MT * NT) + tile) * 16 + r) * 64 + lane];
            }
        }
        if (ks != 0) return;
    }
    const size_t sl = blockIdx.x;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        float* o = slab + (sl * 9 + tap) * COUT * CIN;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kgrp;
            o[(size_t)co * CIN + nt * 32 + i32] = acc[tap][r];
        }
    }
}
