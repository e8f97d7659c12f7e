// fp32-accurate contractions on the fp16 matrix cores with a TWO-way split ("f16x3").
//
// bf16x6 (bf16x6_kernels.h) pays six matrix products and three operand planes (6 bytes) per fp32 product; at the part's
// power cap the time of a contraction is energy per product, so fewer products and fewer operand bytes are what shortens
// it.  An fp32 value x splits into two fp16 numbers
//     hi = fp16(x)  (round to nearest, 11 significant bits)         lo = fp16((x - hi) * 2^11)
// (x - hi is exact in fp32; the residual is scaled by 2^11 so that it sits in fp16's normal range wherever hi does and
// no bit is lost to fp16's narrow exponent).  hi + lo * 2^-11 carries 22-23 significant bits of x.  Of the four cross
// products three are kept:
//     x * y  ~  hi_x * hi_y  +  2^-11 (hi_x * lo_y + lo_x * hi_y)            (lo * lo, weight 2^-22, dropped)
// as three v_mfma_f32_32x32x16_f16 into TWO f32 accumulators -- one for hi * hi, one for the two cross terms, which are
// 2^11 too large and are folded in once at the end (acc0 + 2^-11 acc1) -- and two operand planes (4 bytes per value).
// Per 16-deep k step: 96 matrix-pipe cycles instead of 192, 4 instead of 6 bytes per operand value through L2 / LDS.
// Accuracy: operand representation 2^-23 relative (against exact for bf16x3), dropped term 2^-22 * |lo_x lo_y| <= 2^-24
// relative; measured against a float64 product on the real GRU projection operands in profiles/r04/ab_f16x3.txt.
// Range: fp16's, |x| < 65504 (the GRU inputs are BatchNorm + ReLU outputs and GRU states; weights are O(0.1)); values
// below 2^-14 keep an ABSOLUTE error of 2^-36.
#pragma once
#include "bf16x6_kernels.h"     // (includes f16_split.h: the two-way split itself)

// ------------------------------------------------------------------------------------------
// C[m][z*N + n] = sum_k A[m][k] * Bz[n][k] + biasz[n] with A, B given as f16x2 planes (same contract as
// gemm_nt_bf16x6_v3_kernel with two planes per operand): Ap [2][M][K], Bp0 / Bp1 [2][N][K] fp16.
// Tile 160 x 256, BK = 32 (64 bytes per row and plane), 8 waves, wave = 160 x 32 strip: five hi*hi and five cross accumulators
// (160 registers).  Stage = A [2][160][64 B] + B [2][256][64 B] = 53,248 B; NST = 3 stages (159,744 B, the whole LDS of a CU):
// a K tile is now 30 MFMAs per wave (960 cycles; two waves per SIMD: ~0.8 us) -- shorter than an L2 round trip under load, so
// a tile's LDS-DMA pieces are issued TWO tiles ahead and the wait in front of the tile barrier is a counted vmcnt that leaves
// the newest tile's pieces in flight.  hipcc orders every LDS read it can see behind ALL outstanding LDS-DMA of the wave
// (vmcnt(0)), so the fragment reads are inline asm (12 ds_read_b128 + their wait per 16-deep step) and the barrier is a bare
// s_barrier.  Swizzle as in the bf16x6 kernel: 16-byte chunk c of row r sits at chunk c ^ ((r >> 2) & 3), applied on the SOURCE
// address of the DMA.  NST = 2 is the bf16x6 kernel's schedule (vmcnt(0) per tile), kept for the A/B.
// KNOCK (devtools/kernel_ab/bench_gemm.hip): bit 0 = no staging, bit 2 = no MFMAs (timing only).
// ------------------------------------------------------------------------------------------
constexpr int H3_BM = 160, H3_BN = 256, H3_BK = 32;
constexpr int H3_APLANE = H3_BM * 64, H3_BPLANE = H3_BN * 64;          // 10,240 / 16,384
constexpr int H3_BOFF = 2 * H3_APLANE;                                  // 20,480
constexpr int H3_STAGE = H3_BOFF + 2 * H3_BPLANE;                       // 53,248
constexpr int H3_APIECES = 2 * H3_BM / 16, H3_PIECES = H3_APIECES + 2 * H3_BN / 16;   // 20, 52
constexpr int H3_PPW = (H3_PIECES + 7) / 8;                             // 7 (waves 0-3: 7 pieces, waves 4-7: 6)
constexpr int h3_lds_bytes(int nst) { return nst * H3_STAGE; }

// five A fragments of one plane pair + the wave's B fragments of one 16-deep step, then the wait: 12 x ds_read_b128
__device__ __forceinline__ void h3_read_step(unsigned aaddr, unsigned baddr, f16x8 (&ah)[5], f16x8 (&al)[5], f16x8& bh, f16x8& bl) {
    asm volatile(
        "ds_read_b128 %0, %12\n\t"
        "ds_read_b128 %10, %13 offset:20480\n\t"
        "ds_read_b128 %5, %12 offset:10240\n\t"
        "ds_read_b128 %11, %13 offset:36864\n\t"
        "ds_read_b128 %1, %12 offset:2048\n\t"
        "ds_read_b128 %6, %12 offset:12288\n\t"
        "ds_read_b128 %2, %12 offset:4096\n\t"
        "ds_read_b128 %7, %12 offset:14336\n\t"
        "ds_read_b128 %3, %12 offset:6144\n\t"
        "ds_read_b128 %8, %12 offset:16384\n\t"
        "ds_read_b128 %4, %12 offset:8192\n\t"
        "ds_read_b128 %9, %12 offset:18432\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(ah[0]), "=&v"(ah[1]), "=&v"(ah[2]), "=&v"(ah[3]), "=&v"(ah[4]),
          "=&v"(al[0]), "=&v"(al[1]), "=&v"(al[2]), "=&v"(al[3]), "=&v"(al[4]), "=&v"(bh), "=&v"(bl)
        : "v"(aaddr), "v"(baddr)
        : "memory");
}

template <int NST = 3, int KNOCK = 0>
static __global__ __launch_bounds__(512) void gemm_nt_f16x3_kernel(
    const unsigned short* __restrict__ Ap, const unsigned short* __restrict__ Bp0, const unsigned short* __restrict__ Bp1,
    const float* __restrict__ bias0, const float* __restrict__ bias1, float* __restrict__ C, int ldc, int M, int N, int K) {
    static_assert(NST == 2 || NST == 3, "two or three stages");
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

    // LDS-DMA pieces of this wave: g = wv + 8 i; pieces 0..19 = A (plane g / 10, rows 16 (g % 10)..), 20..51 = B (plane (g - 20) / 16,
    // rows 16 ((g - 20) % 16)..); piece g lands at byte g * 1024 of its stage
    const int lr = lane >> 2;
    const int csrc = (lane & 3) ^ ((lr >> 2) & 3);
    unsigned int poff[H3_PPW];
#pragma unroll
    for (int i = 0; i < H3_PPW; ++i) {
        const int g = wv + 8 * i;
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
        const int g = wv + 8 * i;
        if (g < H3_PIECES) {
            const unsigned short* src = (g < H3_APIECES ? Ap : Bp) + poff[i] + (size_t)kt * H3_BK;
            __builtin_amdgcn_global_load_lds((sir_gptr_t)src, (sir_lptr_t)(h3_smem + buf * H3_STAGE + g * 1024), 16, 0, 0);
        }
    };

    f32x16 acc0[5], acc1[5];
#pragma unroll
    for (int mt = 0; mt < 5; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[mt][r] = 0.0f; acc1[mt][r] = 0.0f; }

    // LDS byte address of this lane's chunk inside its row, per 16-deep step (stage 0)
    const unsigned sbase = (unsigned)(uintptr_t)h3_smem;
    unsigned fa[2], fb[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        fa[ks] = sbase + m * 64 + ((((ks << 1) | h) ^ ((m >> 2) & 3)) << 4);
        fb[ks] = fa[ks] + wv * 2048;
    }
    const int nk = K / H3_BK;

    // ISSUE: tile `ktn` (< nk) is staged into `bufn` between the MFMAs: waves 0-3 during step 0, their SIMD partners 4-7 during step 1
    auto compute = [&](int buf, auto issue_c, int ktn, int bufn) {
        constexpr bool ISSUE = decltype(issue_c)::value;
        const unsigned so = (unsigned)(buf * H3_STAGE);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 ah[5], al[5], bh, bl;
            h3_read_step(fa[ks] + so, fb[ks] + so, ah, al, bh, bl);
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int mt = 0; mt < 5; ++mt) {
                    if (KNOCK & 4) {
                        acc0[mt][0] += (float)ah[mt][0] * (float)bh[0] + (float)al[mt][1] * (float)bl[1];
                    } else {
                        if (t == 0) acc1[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh, acc1[mt], 0, 0, 0);
                        if (t == 1) acc1[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl, acc1[mt], 0, 0, 0);
                        if (t == 2) acc0[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh, acc0[mt], 0, 0, 0);
                    }
                    const int idx = t * 5 + mt;
                    if (ISSUE && (idx & 1) && (idx >> 1) < H3_PPW && ks == (wv >> 2)) piece(idx >> 1, ktn, bufn);
                }
            __builtin_amdgcn_sched_barrier(0);              // (left alone hipcc sinks this step's MFMAs below the next step's reads: both fragment sets live, spills)
        }
    };

    // prologue: NST - 1 tiles in flight
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) {
#pragma unroll
            for (int i = 0; i < H3_PPW; ++i) piece(i, s, s);
        }
    // tile kt must have landed before its barrier; the pieces of the tiles behind it (NST = 3: tile kt + 1, issued during the previous
    // tile) may still fly.  Main loop (stages the tile NST - 1 ahead) and tail (nothing left to stage) are two loops, not two branches
    // of one: with both bodies under one loop hipcc accumulated out of place (twice the accumulator registers, 290 spilled)
    auto tile_wait = [&](bool more) {
        if (NST == 2 || !more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (wv < 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(H3_PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(H3_PPW - 1) : "memory");
        asm volatile("s_barrier" ::: "memory");             // every wave's pieces of the tile are in LDS; the stage of the tile before it is free
    };
    int buf = 0, bufn = NST - 1, kt = 0;
    for (; kt + NST - 1 < nk; ++kt) {
        tile_wait(true);
        compute(buf, std::true_type{}, kt + NST - 1, bufn);
        buf = buf + 1 == NST ? 0 : buf + 1;
        bufn = bufn + 1 == NST ? 0 : bufn + 1;
    }
    for (; kt < nk; ++kt) {
        tile_wait(kt + 1 < nk);
        compute(buf, std::false_type{}, 0, 0);
        buf = buf + 1 == NST ? 0 : buf + 1;
    }

    const int n = n0 + wv * 32 + m;
    const float bv = bias ? bias[n] : 0.0f;
    float* crow = C + (size_t)(m0 + 4 * h) * ldc + (size_t)z * N + n;
    if (m0 + H3_BM <= M) {                                  // whole tile in range: straight-line stores
#pragma unroll
        for (int mt = 0; mt < 5; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                crow[(size_t)(mt * 32 + (r & 3) + 8 * (r >> 2)) * ldc] = fmaf(acc1[mt][r], H3_LO_INV, acc0[mt][r]) + bv;
    } else {
#pragma unroll
        for (int mt = 0; mt < 5; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = mt * 32 + (r & 3) + 8 * (r >> 2);
                if (m0 + 4 * h + ro < M) crow[(size_t)ro * ldc] = fmaf(acc1[mt][r], H3_LO_INV, acc0[mt][r]) + bv;
            }
    }
}

static inline bool gemm_f16x3_ok(int M, int N, int K) {
    return N % H3_BN == 0 && K % H3_BK == 0 && (size_t)2 * M * K < ((size_t)1 << 31) && (size_t)2 * N * K < ((size_t)1 << 31);
}

// C[M][2 N] (+ bias) = A x [B0; B1]^T from pre-split f16x2 planes (the GRU input projections: N = 768 per direction)
static inline hipError_t launch_gemm_nt_f16x3(sir_handle* h, hipStream_t st, const unsigned short* Ap, const unsigned short* Bp0,
                                              const unsigned short* Bp1, const float* bias0, const float* bias1, float* C, int ldc,
                                              int M, int N, int K) {
    if (!gemm_f16x3_ok(M, N, K)) return hipErrorInvalidValue;
    if (!h->attr_gemm_v3) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_f16x3_kernel<3, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, h3_lds_bytes(3));
        if (e != hipSuccess) return e;
        h->attr_gemm_v3 = true;
    }
    const int nwg = ((M + H3_BM - 1) / H3_BM) * 2 * (N / H3_BN);
    hipLaunchKernelGGL((gemm_nt_f16x3_kernel<3, 0>), dim3(nwg), dim3(512), h3_lds_bytes(3), st, Ap, Bp0, Bp1, bias0, bias1, C, ldc, M, N, K);
    return hipGetLastError();
}
