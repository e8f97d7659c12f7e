// EXPERIMENT (round 4, harness only -- devtools/kernel_ab/bench_gemm.hip): the f16x3 projection GEMM of csrc/f16x3_kernels.h with FOUR fat waves instead of
// eight.  Same contract, same 160 x 256 workgroup tile, same three-stage LDS-DMA ring and swizzle as gemm_nt_f16x3_kernel; what changes:
//   * wave = 160 x 64 (five A fragments x TWO B fragments): 14 ds_read_b128 per 30 MFMAs instead of 12 per 15 -- the product kernel spends 49 of its
//     61 us without any MFMA (profiles/r04/ab_f16x3.txt): it is bound by its fragment reads (6.3 MB per workgroup at ~64 B / clock / CU);
//   * ONE accumulator per product, 2^11 too large (terms (Al', Bh), (Ah, Bl'), (Ah, Bh 2^11) as in the Winograd kernels; needs |B| < 32: weights):
//     160 accumulator registers for the 5 x 2 tile, the epilogue scales by the exact 2^-11;
//   * one wave per SIMD saturates the matrix pipe (profiles/r04/ubench_mfma_valu_mix.txt), so there is no partner wave to hide the reads behind: the
//     fragments are double-buffered in registers -- step 1's are read during step 0's MFMAs, the NEXT tile's step 0 during step 1's -- with the tile
//     barrier moved between the two steps (tile kt + 1 must have landed before its fragments are read; the stage of tile kt is free once every wave
//     has its step-1 fragments).
// RESULT: correct, 25-65 % slower than the product kernel in both forms (profiles/r04/gemm_f16x3_four_waves.txt): not bound by LDS bytes.
#pragma once
#include "../../speech-intent-recognizer_amd/csrc/f16x3_kernels.h"

constexpr int H3W4_PPW = H3_PIECES / 4;                      // 13 LDS-DMA pieces per wave and tile
static_assert(H3_PIECES % 4 == 0, "pieces per wave");

// 14 fragment reads of one 16-deep step, NOT waited for (h3w4_wait ties the registers to the wait)
__device__ __forceinline__ void h3w4_read_issue(unsigned aaddr, unsigned baddr, f16x8 (&ah)[5], f16x8 (&al)[5], f16x8 (&bh)[2], f16x8 (&bl)[2]) {
    asm volatile(
        "ds_read_b128 %0, %14\n\t"
        "ds_read_b128 %10, %15 offset:20480\n\t"
        "ds_read_b128 %5, %14 offset:10240\n\t"
        "ds_read_b128 %12, %15 offset:36864\n\t"
        "ds_read_b128 %11, %15 offset:22528\n\t"
        "ds_read_b128 %13, %15 offset:38912\n\t"
        "ds_read_b128 %1, %14 offset:2048\n\t"
        "ds_read_b128 %6, %14 offset:12288\n\t"
        "ds_read_b128 %2, %14 offset:4096\n\t"
        "ds_read_b128 %7, %14 offset:14336\n\t"
        "ds_read_b128 %3, %14 offset:6144\n\t"
        "ds_read_b128 %8, %14 offset:16384\n\t"
        "ds_read_b128 %4, %14 offset:8192\n\t"
        "ds_read_b128 %9, %14 offset:18432"
        : "=&v"(ah[0]), "=&v"(ah[1]), "=&v"(ah[2]), "=&v"(ah[3]), "=&v"(ah[4]),
          "=&v"(al[0]), "=&v"(al[1]), "=&v"(al[2]), "=&v"(al[3]), "=&v"(al[4]), "=&v"(bh[0]), "=&v"(bh[1]), "=&v"(bl[0]), "=&v"(bl[1])
        : "v"(aaddr), "v"(baddr)
        : "memory");
}
__device__ __forceinline__ void h3w4_wait(f16x8 (&ah)[5], f16x8 (&al)[5], f16x8 (&bh)[2], f16x8 (&bl)[2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(ah[4]), "+v"(al[0]), "+v"(al[1]), "+v"(al[2]), "+v"(al[3]), "+v"(al[4]),
                   "+v"(bh[0]), "+v"(bh[1]), "+v"(bl[0]), "+v"(bl[1])
                 :: "memory");
}

// KNOCK: bit 0 = no staging, bit 2 = no MFMAs (timing only)
// XT = true: the next tile's step-0 fragments are read across the tile boundary (barrier between the two steps: a tile's pieces have ONE tile to land);
// XT = false: wait + barrier at the end of a tile as in the product kernel (two tiles to land), the first step's reads of a tile are exposed
template <int KNOCK = 0, bool XT = true>
static __global__ __launch_bounds__(256) void gemm_nt_f16x3_w4_kernel(
    const unsigned short* __restrict__ Ap, const unsigned short* __restrict__ Bp0, const unsigned short* __restrict__ Bp1,
    const float* __restrict__ bias0, const float* __restrict__ bias1, float* __restrict__ C, int ldc, int M, int N, int K) {
    constexpr int NST = 3;
    extern __shared__ __attribute__((aligned(1024))) unsigned char h3_smem[];
    const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, rem = nwg & 7;
    const int wgid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (orig >> 3);
    const int nbd = N / H3_BN, nb = 2 * nbd;
    const int mblk = wgid / nb, nbk = wgid - mblk * nb, z = nbk / nbd;
    const int m0 = mblk * H3_BM, n0 = (nbk - z * nbd) * H3_BN;
    const unsigned short* __restrict__ Bp = z ? Bp1 : Bp0;
    const float* __restrict__ bias = z ? bias1 : bias0;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), m = lane & 31, h = lane >> 5;
    const size_t planeA = (size_t)M * K, planeB = (size_t)N * K;

    // LDS-DMA pieces of this wave: g = wv + 4 i (the piece numbering and LDS image of gemm_nt_f16x3_kernel)
    const int lr = lane >> 2;
    const int csrc = (lane & 3) ^ ((lr >> 2) & 3);
    unsigned int poff[H3W4_PPW];
#pragma unroll
    for (int i = 0; i < H3W4_PPW; ++i) {
        const int g = wv + 4 * i;
        if (g < H3_APIECES) {
            const int pl = g / 10, rg = g - pl * 10;
            int row = m0 + rg * 16 + lr;
            row = row < M ? row : M - 1;
            poff[i] = (unsigned int)(pl * planeA + (size_t)row * K + csrc * 8);
        } else {
            const int gb = g - H3_APIECES, pl = (gb >> 4) & 1, rg = gb & 15;
            poff[i] = (unsigned int)(pl * planeB + (size_t)(n0 + rg * 16 + lr) * K + csrc * 8);
        }
    }
    auto piece = [&](int i, int kt, int buf) {
        if (KNOCK & 1) return;
        const int g = wv + 4 * i;
        const unsigned short* src = (g < H3_APIECES ? Ap : Bp) + poff[i] + (size_t)kt * H3_BK;
        __builtin_amdgcn_global_load_lds((sir_gptr_t)src, (sir_lptr_t)(h3_smem + buf * H3_STAGE + g * 1024), 16, 0, 0);
    };

    f32x16 acc[5][2];
#pragma unroll
    for (int mt = 0; mt < 5; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    const unsigned sbase = (unsigned)(uintptr_t)h3_smem;
    unsigned fa[2], fb[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        fa[ks] = sbase + m * 64 + ((((ks << 1) | h) ^ ((m >> 2) & 3)) << 4);
        fb[ks] = fa[ks] + wv * 4096;                        // this wave's two 32-row blocks of B: + 0 / + 2048 (immediate offsets of the reads)
    }
    const int nk = K / H3_BK;

    // thirty MFMAs of one step on fragment set (ah, al, bh, bl); every second one is followed by one LDS-DMA piece of tile `ktn` when ISSUE
    auto mfmas = [&](f16x8 (&ah)[5], f16x8 (&al)[5], f16x8 (&bh)[2], f16x8 (&bl)[2], auto issue_c, int ktn, int bufn) {
        constexpr bool ISSUE = decltype(issue_c)::value;
        f16x8 b2k[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) b2k[nt] = bh[nt] * (_Float16)2048.0f;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int mt = 0; mt < 5; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if (KNOCK & 4) {
                        acc[mt][nt][0] += (float)ah[mt][0] * (float)bh[nt][0] + (float)al[mt][1] * (float)bl[nt][1];
                    } else {
                        if (t == 0) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                        if (t == 1) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
                        if (t == 2) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], b2k[nt], acc[mt][nt], 0, 0, 0);
                    }
                    const int idx = (t * 5 + mt) * 2 + nt;
                    if (ISSUE && (idx & 1) && (idx >> 1) < H3W4_PPW) piece(idx >> 1, ktn, bufn);
                }
    };

    f16x8 ah0[5], al0[5], bh0[2], bl0[2], ah1[5], al1[5], bh1[2], bl1[2];
    // one tile: step 0 on set 0 (ready at entry) while set 1 is read; barrier for tile kt + 1; step 1 on set 1 while tile kt + 1's set 0 is read
    auto tile = [&](int buf, auto issue_c, auto next_c, int ktn, int bufn) {
        constexpr bool ISSUE = decltype(issue_c)::value, NEXT = decltype(next_c)::value;
        const unsigned so = (unsigned)(buf * H3_STAGE);
        if (!XT) {
            h3w4_read_issue(fa[0] + so, fb[0] + so, ah0, al0, bh0, bl0);
            h3w4_wait(ah0, al0, bh0, bl0);
        }
        h3w4_read_issue(fa[1] + so, fb[1] + so, ah1, al1, bh1, bl1);
        mfmas(ah0, al0, bh0, bl0, issue_c, ktn, bufn);
        __builtin_amdgcn_sched_barrier(0);
        h3w4_wait(ah1, al1, bh1, bl1);
        auto next_tile_landed = [&]() {
            // tile kt + 1 (its 13 pieces per wave issued a tile ago) must be in LDS; the 13 pieces of tile kt + 2, just issued, may still fly
            if (ISSUE) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(H3W4_PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        };
        if (NEXT && XT) {
            next_tile_landed();
            const unsigned sn = (unsigned)((buf + 1 == NST ? 0 : buf + 1) * H3_STAGE);
            h3w4_read_issue(fa[0] + sn, fb[0] + sn, ah0, al0, bh0, bl0);
        }
        mfmas(ah1, al1, bh1, bl1, std::false_type{}, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (NEXT && XT) h3w4_wait(ah0, al0, bh0, bl0);
        if (NEXT && !XT) next_tile_landed();
    };

    // prologue: two tiles in flight, the first one landed and its step-0 fragments read
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) {
#pragma unroll
            for (int i = 0; i < H3W4_PPW; ++i) piece(i, s, s);
        }
    if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(H3W4_PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (XT) {
        h3w4_read_issue(fa[0], fb[0], ah0, al0, bh0, bl0);
        h3w4_wait(ah0, al0, bh0, bl0);
    }

    int buf = 0, bufn = NST - 1, kt = 0;
    for (; kt + NST - 1 < nk; ++kt) {                       // stages tile kt + 2
        tile(buf, std::true_type{}, std::true_type{}, kt + NST - 1, bufn);
        buf = buf + 1 == NST ? 0 : buf + 1;
        bufn = bufn + 1 == NST ? 0 : bufn + 1;
    }
    for (; kt + 1 < nk; ++kt) {                             // nothing left to stage, one more tile follows
        tile(buf, std::false_type{}, std::true_type{}, 0, 0);
        buf = buf + 1 == NST ? 0 : buf + 1;
    }
    if (kt < nk) tile(buf, std::false_type{}, std::false_type{}, 0, 0);

    float* crow = C + (size_t)(m0 + 4 * h) * ldc + (size_t)z * N + n0 + wv * 64 + m;
    const bool whole = m0 + H3_BM <= M;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const float bv = bias ? bias[n0 + wv * 64 + nt * 32 + m] : 0.0f;
#pragma unroll
        for (int mt = 0; mt < 5; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = mt * 32 + (r & 3) + 8 * (r >> 2);
                if (whole || m0 + 4 * h + ro < M) crow[(size_t)ro * ldc + nt * 32] = fmaf(acc[mt][nt][r], H3_LO_INV, bv);
            }
    }
}
