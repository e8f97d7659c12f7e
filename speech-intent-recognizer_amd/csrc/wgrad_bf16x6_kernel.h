// Convolution weight gradient on the bf16 matrix cores (bf16x6 products, fp32 accuracy).
//
//   dW[co][ci][ky][kx] = sum_{b,y,x} dz[b][y][x][co] * a[b][y+ky-1][x+kx-1][ci]
//   GEMM view per tap: M = co, N = ci, K = pixels of a row.  conv_wgrad_mfma_kernel does this with
//   v_mfma_f32_32x32x2_f32 (two pixels per instruction, one ds_read_b32 per operand and lane: LDS-read bound,
//   ~95 TF).  v_mfma_f32_32x32x16_bf16 takes 16 pixels per instruction but wants, per lane, EIGHT CONSECUTIVE
//   PIXELS of one channel -- the transpose of the NHWC activations.  The transposition happens while a row is
//   staged: thread = (pixel, 4 channels) reads a float4, splits it into the three bf16 planes and writes the 12
//   halves with ds_write_b16 into channel-major LDS rows dzT[plane][co][pixel] / aT[ring row][plane][ci][pixel + 1]
//   (lanes run along the pixels, so the 2-byte stores of a wave fill whole bank words).
//   Tap shift kx: the B fragment of tap kx starts kx pixels later than the aligned 16-byte chunk; a lane reads the
//   aligned chunk plus the next word (5 words) once per (ky, plane) and forms the three kx fragments with
//   v_alignbit_b32 (kx = 1) or by register selection (kx = 0, 2).
//   8 waves: wave = (mt, nt, ks); every wave accumulates ALL NINE taps of its 32 x 32 (co, ci) tile (9 accumulators)
//   and takes the k-steps s = ks (mod KSPLIT) of a row, KSPLIT = 8 / (MT * NT) (1 for 64->128, 4 for 32->64); each
//   wave writes its own slab [tap][co][ci] -- same slab layout and deterministic two-pass reduce as before.
#pragma once
#include "bf16x6_kernels.h"

constexpr int wgrad_x6_kpx(int W) { return (W + 15) / 16 * 16; }
constexpr int wgrad_x6_rowb_z(int W) { return wgrad_x6_kpx(W) * 2 + 16; }             // dzT row bytes (odd number of 16-B slots)
constexpr int wgrad_x6_rowb_a(int W) { return (wgrad_x6_kpx(W) + 16) * 2 + 16; }      // aT row bytes: + halo and shifted reads
inline size_t wgrad_x6_lds_bytes(int cin, int cout, int W) {
    return (size_t)3 * cout * wgrad_x6_rowb_z(W) + (size_t)9 * cin * wgrad_x6_rowb_a(W);
}
constexpr int wgrad_x6_ksplit(int cin, int cout) { return 8 / ((cout / 32) * (cin / 32)); }
constexpr int wgrad_x6_max_w(int cout) { return cout >= 128 ? 64 : 128; }   // row widths the register prefetch is sized for

template <int CIN, int COUT>
__global__ __launch_bounds__(512) void conv_wgrad_bf16x6_kernel(const float* __restrict__ dz, const float* __restrict__ a,
                                                                 float* __restrict__ slab, int H, int W, int RB) {
    constexpr int MT = COUT / 32, NT = CIN / 32, KSPLIT = 8 / (MT * NT);
    static_assert(MT * NT * KSPLIT == 8, "8 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char wl[];
    const int kpx = (W + 15) / 16 * 16;
    const int rowz = kpx * 2 + 16, rowa = (kpx + 16) * 2 + 16;
    unsigned char* dzT = wl;                                   // [3][COUT][rowz]
    unsigned char* aT = wl + (size_t)3 * COUT * rowz;          // [3 ring][3][CIN][rowa], pixel gx at element gx + 1
    const size_t lds_bytes = (size_t)3 * COUT * rowz + (size_t)9 * CIN * rowa;
    const int blocks_per_img = H / RB;
    const int b = blockIdx.x / blocks_per_img, y0 = (blockIdx.x % blocks_per_img) * RB;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int mt = wv / (NT * KSPLIT), nt = (wv / KSPLIT) % NT, ks = wv % KSPLIT;
    const int i32 = lane & 31, kgrp = lane >> 5;

    for (int i = tid; i < (int)(lds_bytes / 16); i += 512) reinterpret_cast<uint4*>(wl)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    // A row is staged in two halves so that its global-load latency hides behind the previous row's MFMAs:
    //   fetch_row : float4 loads of the thread's items (pixel fastest, then 4-channel group) into registers,
    //   write_row : split into the three bf16 planes and transposed 2-byte stores planes[P][c][pixel + poff].
    constexpr int MAXW = wgrad_x6_max_w(COUT);                 // the host checks W <= MAXW (the LDS image bounds it anyway)
    // staging items: thread = (pixel px = tid % MAXW, channel group c4 = tid / MAXW + (512 / MAXW) * k) -- a power-of-two
    // split, so an item's address is two small integers away from the row base.  (Items numbered it = tid + 512 k with
    // it % W / it / W kept one 64-bit offset per item alive; those spilled, and every reload -- a scratch load, counted in
    // vmcnt -- put an s_waitcnt vmcnt(0) in front of the next global load: the row's six loads were issued one at a time.)
    constexpr int GPP = 512 / MAXW;                            // channel groups covered per pass of the 512 threads
    constexpr int NZ = (COUT / 4) / GPP, NA = (CIN / 4) / GPP; // float4 items per thread
    static_assert((COUT / 4) % GPP == 0 && (CIN / 4) % GPP == 0, "staging passes");
    const int spx = tid & (MAXW - 1), sg = tid / MAXW;
    const bool spx_ok = spx < W;
    auto fetch_row = [&](const float* src, int C, auto& regs) {
#pragma unroll
        for (int k = 0; k < (int)(sizeof(regs) / sizeof(float4)); ++k) {
            regs[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (src && spx_ok) regs[k] = *reinterpret_cast<const float4*>(src + spx * C + 4 * (sg + GPP * k));
        }
    };
    auto write_row = [&](const auto& regs, int C, unsigned char* base, int rowb, int poff) {
        const size_t plane = (size_t)C * rowb;
        if (!spx_ok) return;
#pragma unroll
        for (int k = 0; k < (int)(sizeof(regs) / sizeof(float4)); ++k) {
            const int c4 = sg + GPP * k;
            uint2 hh, mm, ll;
            split3_quad(regs[k], hh, mm, ll);
            unsigned char* d = base + (size_t)(4 * c4) * rowb + (spx + poff) * 2;
            const unsigned hw[4] = {hh.x & 0xFFFFu, hh.x >> 16, hh.y & 0xFFFFu, hh.y >> 16};
            const unsigned mw[4] = {mm.x & 0xFFFFu, mm.x >> 16, mm.y & 0xFFFFu, mm.y >> 16};
            const unsigned lw[4] = {ll.x & 0xFFFFu, ll.x >> 16, ll.y & 0xFFFFu, ll.y >> 16};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                *reinterpret_cast<unsigned short*>(d + (size_t)e * rowb) = (unsigned short)hw[e];
                *reinterpret_cast<unsigned short*>(d + plane + (size_t)e * rowb) = (unsigned short)mw[e];
                *reinterpret_cast<unsigned short*>(d + 2 * plane + (size_t)e * rowb) = (unsigned short)lw[e];
            }
        }
    };
    auto a_src = [&](int y) -> const float* { return (y >= 0 && y < H) ? a + (((size_t)b * H + y) * W) * CIN : nullptr; };
    auto a_slot = [&](int y) { return aT + (size_t)((y + 1) % 3) * 3 * CIN * rowa; };          // input row y -> ring slot (y + 1) % 3

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    float4 pz[NZ], pa[NA];
    fetch_row(a_src(y0 - 1), CIN, pa);
    write_row(pa, CIN, a_slot(y0 - 1), rowa, 1);
    fetch_row(a_src(y0), CIN, pa);
    write_row(pa, CIN, a_slot(y0), rowa, 1);
    fetch_row(a_src(y0 + 1), CIN, pa);                         // rows y0 + 1 (a) and y0 (dz) are in flight
    fetch_row(dz + (((size_t)b * H + y0) * W) * COUT, COUT, pz);
    const int nks = kpx / 16;
    for (int y = y0; y < y0 + RB; ++y) {
        __syncthreads();                                       // previous row fully consumed
        write_row(pa, CIN, a_slot(y + 1), rowa, 1);
        write_row(pz, COUT, dzT, rowz, 0);
        __syncthreads();
        // The next row's loads are issued AFTER the barrier: __syncthreads() drains vmcnt, so loads issued in front of it
        // were waited for right there (knock-out timing: 86 of the kernel's 275 us were exposed fetch time); issued here
        // they fly during this row's MFMAs and are only waited for at the next row's first barrier.
        if (y + 1 < y0 + RB) {
            fetch_row(a_src(y + 2), CIN, pa);
            fetch_row(dz + (((size_t)b * H + y + 1) * W) * COUT, COUT, pz);
        }
        const unsigned char* zrow = dzT + (size_t)(mt * 32 + i32) * rowz + kgrp * 16;
        for (int s = ks; s < nks; s += KSPLIT) {
            bf16x8 afr[3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
                afr[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(zrow + (size_t)p * COUT * rowz + s * 32));
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const unsigned char* arow = aT + (size_t)((y + ky) % 3) * 3 * CIN * rowa + (size_t)(nt * 32 + i32) * rowa + kgrp * 16 + s * 32;
                unsigned w5[3][5];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const uint4 c0 = *reinterpret_cast<const uint4*>(arow + (size_t)p * CIN * rowa);
                    w5[p][0] = c0.x; w5[p][1] = c0.y; w5[p][2] = c0.z; w5[p][3] = c0.w;
                    w5[p][4] = *reinterpret_cast<const unsigned*>(arow + (size_t)p * CIN * rowa + 16);
                }
                bf16x8 bfr[3][3];                              // [kx][plane]
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        uint4 v;
                        if (kx == 0) v = make_uint4(w5[p][0], w5[p][1], w5[p][2], w5[p][3]);
                        else if (kx == 2) v = make_uint4(w5[p][1], w5[p][2], w5[p][3], w5[p][4]);
                        else v = make_uint4(__builtin_amdgcn_alignbit(w5[p][1], w5[p][0], 16), __builtin_amdgcn_alignbit(w5[p][2], w5[p][1], 16),
                                            __builtin_amdgcn_alignbit(w5[p][3], w5[p][2], 16), __builtin_amdgcn_alignbit(w5[p][4], w5[p][3], 16));
                        bfr[kx][p] = __builtin_bit_cast(bf16x8, v);
                    }
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // small terms first
#pragma unroll
                for (int t6 = 0; t6 < 6; ++t6)                 // the three taps of this ky interleaved: independent accumulators
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[PA[t6]], bfr[kx][PB[t6]], acc[ky * 3 + kx], 0, 0, 0);
            }
        }
    }
    const size_t sl = (size_t)blockIdx.x * KSPLIT + ks;
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
