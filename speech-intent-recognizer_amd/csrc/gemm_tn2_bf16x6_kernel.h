// Token-reduction GEMMs of the GRU backward (dW = dG^T X, dX = dG [W; W_reverse]) with SPECIALISED waves: the same tiles, LDS
// images, transposed fragment reads and epilogue as gemm_tn_bf16x6_kernel (gemm_tn_bf16x6_kernel.h: read that header first), but
// a BM x 256 x 32 stage is prepared by 8 PRODUCER waves (fp32 loads a whole stage ahead into a second register set, three-way
// bf16 split, ds_write_b64 into stage buffer s & 1) while 8 CONSUMER waves multiply stage s - 1 from the other buffer: one bare
// s_barrier per stage, 1024 threads, one workgroup per CU.  In the first kernel all 8 waves do both with two barriers per stage, and
// the split sits between the MFMA phases (matrix pipe 35-39 % busy, profiles/r03/roofline.md).  The recipe is the producer /
// consumer Winograd kernel's (conv_wino2_bf16x6_kernel.h): a role's registers are live only in its own branch, the wave index is a
// scalar, the barrier does not drain vmcnt.  What the knock-outs (template parameter dbg: 1 = no split / LDS stores, 2 = no
// loads, 16 = idle consumers) showed on the way, B = 256 layer-0 dX, first kernel 140 us:
//   * 4 producer waves, loads under per-lane bounds branches: 171 us -- the compiler cannot count loads issued under branches and
//     waits for vmcnt(0) before every use, draining the next stage's loads too;
//   * branch-free loads a stage ahead (vmcnt(22) .. vmcnt(12) in the ISA): 156 us, producers alone 77 us = 1.6 us per stage for
//     580 instructions: ONE wave per SIMD issues an instruction every ~6 cycles and that, not the VALU, was the producers' limit;
//   * 8 producer waves (two per SIMD, half the items each): 120 us; producers alone 61, consumers alone 88 us (MFMA 73 % busy).
//   * tried and removed: the B operand of dX (the weights) taken from the forward GEMM's bf16x3 planes by LDS-DMA with the chunk
//     swizzle on the source address -- no split, no LDS stores, 4 producer waves.  Parity-correct, slower: 139 us (consumers alone
//     88, producers alone 73, without the DMA 97, without the A operand 116).  6 bytes per element instead of 4 come through L2, and
//     whatever the producers do -- VALU or DMA -- ADDS to the consumers' time instead of hiding behind it: with the matrix pipe this
//     busy the step runs at the board's power limit, so energy per stage, not issue slots, is what counts.
// Measured in the training step (profiles/r03): dX l0 140 -> 120 us, dX l1 (64-row tiles) 100 -> 84, dW 114 -> 110 (mean of layers).
#pragma once
#include "gemm_tn_bf16x6_kernel.h"

constexpr int TN2_NPW = 8;                                   // producer waves (two per SIMD: one wave issues too slowly, see below)
constexpr int TN2_THREADS = 64 * (TN2_NPW + 8);              // + 8 consumer waves
constexpr int TN2_BM = 128;
constexpr size_t tn2_lds_bytes(bool a_km, int bm = TN2_BM) { return 2 * tn_lds_bytes(a_km, bm); }     // two stage buffers: 147,456 / 159,744 B (BM = 128)

// F16 (round 4): the products on the fp16 matrix cores with the two-way split of f16_split.h.  A (the gate gradients, which carry the
// backward's loss scale and therefore sit in fp16's range) is staged as TWO planes (Ah, Al' = residual 2^11); B (activations / weights)
// is first scaled by TN2_BS = 2^-4 (exact) and staged as two planes as well (Bh, Bl'); the consumers form Bh 2^11 in registers (four
// v_pk_mul_f16 per fragment), so that the three products Al' Bh + Ah Bl' + Ah (Bh 2^11) = 2^11 A B go into ONE accumulator (the
// consumers' 64 accumulator registers are what the 128-register budget of a 1024-thread workgroup allows) and the epilogue scales by
// 2^-11 / TN2_BS = 2^-7.  Bh 2^11 is exact while |B| TN2_BS < 32, i.e. |B| < 512 (larger values are clamped: activations behind
// BatchNorm and weights are nowhere near).  3 instead of 6 MFMAs per product, 4 instead of 6 planes through LDS, a shorter split.
constexpr float TN2_BS = 0.0625f;
constexpr float TN2_BLIM = 31.984375f;                       // 65504 / 2048
// (Bh, Bl') of v * SCALE (SCALE a power of two; |v| SCALE clamped below 32 so that Bh 2^11 stays finite)
template <int SCALE_LOG2 = -4>
__device__ __forceinline__ void tn2_split_b(const float4& v, uint2& hh, uint2& ll) {
    constexpr float SC = SCALE_LOG2 >= 0 ? (float)(1 << (SCALE_LOG2 >= 0 ? SCALE_LOG2 : 0)) : 1.0f / (float)(1 << (SCALE_LOG2 < 0 ? -SCALE_LOG2 : 0));
    auto pair = [](float a, float b, unsigned& h, unsigned& l) {
        sir_f32x2 x = {__builtin_fminf(__builtin_fmaxf(a * SC, -TN2_BLIM), TN2_BLIM), __builtin_fminf(__builtin_fmaxf(b * SC, -TN2_BLIM), TN2_BLIM)};
        const sir_f16x2 hi = __builtin_convertvector(x, sir_f16x2);
        x -= __builtin_convertvector(hi, sir_f32x2);
        x *= H3_LO_SCALE;
        const sir_f16x2 lo = __builtin_convertvector(x, sir_f16x2);
        h = __builtin_bit_cast(unsigned, hi); l = __builtin_bit_cast(unsigned, lo);
    };
    pair(v.x, v.y, hh.x, ll.x);
    pair(v.z, v.w, hh.y, ll.y);
}
// the same for waves running with MODE.FP16_OVFL = 1: no clamps -- a B value beyond the range saturates in the conversion, and its hi 2^11 in the
// consumers' v_pk_mul_f16 (they run in the same mode), instead of being clamped to 31.98 here: finite either way, one v_med3 per value less
template <int SCALE_LOG2 = -4>
__device__ __forceinline__ void tn2_split_b_ovfl(const float4& v, uint2& hh, uint2& ll) {
    constexpr float SC = SCALE_LOG2 >= 0 ? (float)(1 << (SCALE_LOG2 >= 0 ? SCALE_LOG2 : 0)) : 1.0f / (float)(1 << (SCALE_LOG2 < 0 ? -SCALE_LOG2 : 0));
    split2h_pair_ovfl(v.x * SC, v.y * SC, hh.x, ll.x);
    split2h_pair_ovfl(v.z * SC, v.w * SC, hh.y, ll.y);
}
// fragment of Bh -> fragment of Bh 2^11 (exact: |Bh| < 32)
__device__ __forceinline__ bf16x8 tn2_hi2(const bf16x8& bh) {
    return __builtin_bit_cast(bf16x8, __builtin_bit_cast(f16x8, bh) * (_Float16)2048.0f);
}

__device__ __forceinline__ void tn2_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool A_KM, int dbg = 0, int BM = TN2_BM, bool F16 = false>
__global__ __launch_bounds__(TN2_THREADS, TN2_THREADS / 256) void gemm_tn2_bf16x6_kernel(TnJobs jobs, int M, int K, int kchunk, int seq) {
    constexpr int NC = BM >= 128 ? 2 : 1;                    // BM = 128: consumers 2 x 4, wave tile 64 x 64; BM = 64: 1 x 8, wave tile 64 x 32
    constexpr int AXW = BM * 2, BXW = TN_BN * 2;             // row bytes of the k-major images
    constexpr int APLANE = A_KM ? TN_BK * AXW : BM * TN_ROWB, BPLANE = TN_BK * BXW;
    constexpr int NPA = F16 ? 2 : 3, NPB = NPA;              // operand planes in LDS
    constexpr int STAGE = NPA * APLANE + NPB * BPLANE;
    constexpr int NPT = 64 * TN2_NPW;                         // producer threads
    constexpr int NAQ = BM * 8 / NPT, NBQ = TN_BN * 8 / NPT;  // staging items per PRODUCER thread
    extern __shared__ __attribute__((aligned(16))) unsigned char tl2[];
    // XCD-aware order: workgroup L = x + X y runs on XCD L % 8 (round-robin dispatch), and all workgroups walk k in step.  Give each
    // XCD a CONTIGUOUS range of the (k split, tile) list -- for dW one (split, direction): 6 x 5 tiles that share the split's rows
    // of dG and of the layer input; for dX 25 consecutive tiles = 6 row tiles x all column tiles -- so that an operand row crosses
    // the fabric once per XCD that needs it instead of once per workgroup (L2-miss traffic 3.0-3.1x the unique bytes before; A/B: 2 us per launch).
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int X = gridDim.x, total = X * gridDim.y, L = bx + X * by, g = L & 7, q8 = total >> 3, r8 = total & 7;
        const int f = g * q8 + min(g, r8) + (L >> 3);
        by = f / X;
        bx = f - by * X;
    }
    int j = 0;
    while (j + 1 < jobs.njobs && bx >= jobs.tile0[j + 1]) ++j;
    const int tile = bx - jobs.tile0[j];
    const int N = jobs.N[j], lda = jobs.lda[j], ldb = jobs.ldb[j], shift = jobs.shift[j];
    const int ntn = (N + TN_BN - 1) / TN_BN;
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * TN_BN;
    const float* __restrict__ A = jobs.A[j];
    const float* __restrict__ B = jobs.B[j];
    const float* __restrict__ B2 = jobs.B2[j];
    const int brows = jobs.brows[j];
    const int k_begin = by * kchunk, k_end = min(K, k_begin + kchunk);
    const int nst = (k_end - k_begin + TN_BK - 1) / TN_BK;     // stages of this workgroup
    const int npair = nst / 2 + 1;                            // steps 0 .. nst, rounded up to pairs: 2 * npair barriers in both roles
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // fp16 overflow mode + unclamped splits for the weight-gradient form only (dW 77 -> 74 / 114 -> 109 us; the dX form measured 1.5-2 us SLOWER
    // with it and keeps the clamps: profiles/r04/ab_fp16_ovfl.txt)
#ifdef SIR_W2_CLAMP
    constexpr bool OVFL = false;
#else
    constexpr bool OVFL = F16 && A_KM;
#endif
    if (OVFL) sir_fp16_ovfl_on();                            // producers: unclamped splits; consumers: Bh 2^11 saturates instead of overflowing

    if (wv < TN2_NPW) {
        // ================= producers ==============================================================================================
        __builtin_amdgcn_s_setprio(3);                       // their VALU stream competes with two MFMA-issuing waves per SIMD
        const int ptid = tid;                                // 0 .. NPT - 1
        float4 pa0[NAQ], pb0[NBQ], pa1[NAQ], pb1[NBQ];      // two register sets: the loads of stage s + 1 are issued BEFORE stage s is split
        // A producer wave issues ONE instruction stream: every instruction of its step counts (~4.3 cycles each, measured: 450
        // instructions of splitting and LDS stores = 0.8 us), so the fetch is kept to ~15 scalar + 1 vector instruction per item:
        //  * loads are UNCONDITIONAL and the loop body is branch-free (the steps are rounded up to pairs): under per-lane or per-step
        //    branches the compiler cannot count the outstanding loads and waits for vmcnt(0) before every use, which drains the NEXT
        //    stage's loads too;
        //  * the B row of an item (wave + 4 q) is wave-uniform: row address, the utterance-boundary test of the shifted W_hh
        //    operand (t = token % seq is carried from stage to stage, not divided out) and the [W; W_reverse] selection stay on the
        //    scalar unit; a row that must read as zero (past k_end, across an utterance boundary) is loaded from the zero page;
        //  * rows m >= M of A and columns n >= N of B only feed outputs that are never stored: they read a valid address, nothing
        //    more; a row of A past k_end is clamped to the last valid one (finite values times the zero row of B).
        const unsigned ncol = (n0 + 4 * lane < N) ? 4u * lane : 0u;        // column offset inside the tile's 256-column row segment
        unsigned aoffv[NAQ];                                               // loop-invariant part of the A addresses (elements)
        int arow[NAQ];                                                     // A_KM: token row of the item inside a stage
#pragma unroll
        for (int q = 0; q < NAQ; ++q) {
            const int it = ptid + NPT * q;
            if (A_KM) {
                const int m = m0 + 4 * (it % (BM / 4));
                arow[q] = it / (BM / 4);
                aoffv[q] = m < M ? m : 0;
            } else {
                const int m = m0 + (it >> 3);
                arow[q] = 0;
                aoffv[q] = (unsigned)(m < M ? m : 0) * (unsigned)lda;
            }
        }
        const int akk = 4 * (tid & 7);                                     // !A_KM: k offset of the lane inside a stage
        const int dmod = shift != 0 ? TN_BK % seq : 0;
        int tq[NBQ];                                                       // (token of B item q) % seq, carried; scalar
#pragma unroll
        for (int q = 0; q < NBQ; ++q) tq[q] = shift != 0 ? (k_begin + wv + TN2_NPW * q) % seq : 0;
        const float* __restrict__ zeros = jobs.zeros;
        auto fetch = [&](int k0, float4 (&pa)[NAQ], float4 (&pb)[NBQ]) {
            if (A_KM) {
#pragma unroll
                for (int q = 0; q < NAQ; ++q) {
                    const int tok = min(k0 + arow[q], k_end - 1);
                    pa[q] = *reinterpret_cast<const float4*>(A + ((unsigned)tok * (unsigned)lda + aoffv[q]));
                }
            } else {
                const unsigned kc = k0 + akk < k_end ? k0 + akk : k_end - 4;
#pragma unroll
                for (int q = 0; q < NAQ; ++q) pa[q] = *reinterpret_cast<const float4*>(A + (aoffv[q] + kc));
            }
#pragma unroll
            for (int q = 0; q < NBQ; ++q) {
                const int tok = k0 + wv + TN2_NPW * q;                           // scalar
                bool ok = tok < k_end;
                if (shift != 0) {                                          // neighbouring time step of the same utterance
                    const int t = tq[q] + shift;
                    ok = ok && t >= 0 && t < seq;
                    tq[q] += dmod;
                    tq[q] -= tq[q] >= seq ? seq : 0;
                }
                const int src = tok + shift;
                const float* row = (brows > 0 && src >= brows) ? B2 + (unsigned)(src - brows) * (unsigned)ldb + n0 : B + (unsigned)src * (unsigned)ldb + n0;
                pb[q] = *reinterpret_cast<const float4*>((ok ? row : zeros) + ncol);
            }
        };
        auto stage = [&](unsigned char* AT, unsigned char* BT, const float4 (&pa)[NAQ], const float4 (&pb)[NBQ]) {
#pragma unroll
            for (int q = 0; q < NAQ; ++q) {
                const int it = ptid + NPT * q;
                unsigned char* d = A_KM ? AT + tn_kmaj_off<AXW>(it / (BM / 4), 8 * (it % (BM / 4)))
                                        : AT + (size_t)(it >> 3) * TN_ROWB + (it & 7) * 8;
                if constexpr (F16) {
                    uint2 hh, ll;
                    if constexpr (OVFL) split2h_quad_ovfl(pa[q], hh, ll);
                    else split2h_quad(pa[q], hh, ll);
                    *reinterpret_cast<uint2*>(d) = hh;
                    *reinterpret_cast<uint2*>(d + APLANE) = ll;
                } else {
                    uint2 hh, mm, ll;
                    split3_quad(pa[q], hh, mm, ll);
                    *reinterpret_cast<uint2*>(d) = hh;
                    *reinterpret_cast<uint2*>(d + APLANE) = mm;
                    *reinterpret_cast<uint2*>(d + 2 * APLANE) = ll;
                }
            }
#pragma unroll
            for (int q = 0; q < NBQ; ++q) {
                const int it = ptid + NPT * q;
                unsigned char* d = BT + tn_kmaj_off<BXW>(it >> 6, 8 * (it & 63));
                if constexpr (F16) {
                    uint2 hh, ll;
                    if constexpr (OVFL) tn2_split_b_ovfl(pb[q], hh, ll);   // (Bh, Bl') of B / 16
                    else tn2_split_b(pb[q], hh, ll);
                    *reinterpret_cast<uint2*>(d) = hh;
                    *reinterpret_cast<uint2*>(d + BPLANE) = ll;
                } else {
                    uint2 hh, mm, ll;
                    split3_quad(pb[q], hh, mm, ll);
                    *reinterpret_cast<uint2*>(d) = hh;
                    *reinterpret_cast<uint2*>(d + BPLANE) = mm;
                    *reinterpret_cast<uint2*>(d + 2 * BPLANE) = ll;
                }
            }
        };
        fetch(k_begin, pa0, pb0);
#pragma unroll 1
        for (int s = 0; s < 2 * npair; s += 2) {
            if (!(dbg & 2)) fetch(k_begin + (s + 1) * TN_BK, pa1, pb1);     // a whole step ahead, in flight across the barrier
            if (!(dbg & 1)) stage(tl2, tl2 + NPA * APLANE, pa0, pb0);       // (past the last stage: zeros into the idle buffer)
            tn2_barrier();
            if (!(dbg & 2)) fetch(k_begin + (s + 2) * TN_BK, pa0, pb0);
            if (!(dbg & 1)) stage(tl2 + STAGE, tl2 + STAGE + NPA * APLANE, pa1, pb1);
            tn2_barrier();
        }
        return;
    }

    // ================= consumers ==================================================================================================
    const int cw = wv - TN2_NPW, wm = BM >= 128 ? cw >> 2 : 0, wn = BM >= 128 ? cw & 3 : cw, i32 = lane & 31, kgrp = lane >> 5;
    f32x16 acc[2][NC];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.0f;
    // fragment addresses inside a stage buffer (see gemm_tn_bf16x6_kernel: transposed reads of the k-major images)
    const int tq = (lane >> 2) & 3, tp = lane & 3, tcol = 16 * ((lane >> 4) & 1) + 4 * tp, tk = 8 * (lane >> 5) + tq;
    int aoff[2], boff[NC];
#pragma unroll
    for (int a = 0; a < 2; ++a)
        aoff[a] = A_KM ? tn_kmaj_off<AXW>(tk, 2 * (wm * 64 + a * 32 + tcol)) : (wm * 64 + a * 32 + i32) * TN_ROWB + kgrp * 16;
#pragma unroll
    for (int c = 0; c < NC; ++c) boff[c] = NPA * APLANE + tn_kmaj_off<BXW>(tk, 2 * (wn * 32 * NC + c * 32 + tcol));
    tn2_barrier();                                           // step 0: the producers write stage 0
#pragma unroll 1
    for (int s = 1; s <= nst; ++s) {
        const unsigned char* sb = tl2 + ((s - 1) & 1) * STAGE;
        if (dbg & 16) { tn2_barrier(); continue; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[2][NPA], bf[NC][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                if (p < NPA) {
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                        af[a][p < NPA ? p : 0] = A_KM ? tn_tr_fragment<AXW>(sb + aoff[a] + p * APLANE + ks * 16 * AXW)
                                                      : __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + aoff[a] + p * APLANE + ks * 32));
                }
                if (p < NPB) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) bf[c][F16 && p == 1 ? 2 : p] = tn_tr_fragment<BXW>(sb + boff[c] + p * BPLANE + ks * 16 * BXW);
                }
            }
            if constexpr (F16) {
#pragma unroll
                for (int c = 0; c < NC; ++c) bf[c][1] = tn2_hi2(bf[c][0]);
                // af: 0 = Ah, 1 = Al'; bf: 0 = Bh, 1 = Bh 2^11, 2 = Bl' -- the two cross terms first, then the main one
                constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 2, 1};
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                            acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[a][HA[t3]]), __builtin_bit_cast(f16x8, bf[c][HB[t3]]),
                                                                              acc[a][c], 0, 0, 0);
            } else {
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // small terms first
#pragma unroll
                for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                            acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][PA[t6] % NPA], bf[c][PB[t6]], acc[a][c], 0, 0, 0);
            }
        }
        tn2_barrier();
    }
    if (!(nst & 1)) tn2_barrier();                           // the odd step of the producers' last pair
    float* out = jobs.slab[j] + (size_t)by * jobs.slab_stride[j];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int n = n0 + wn * 32 * NC + c * 32 + i32;
            if (n >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kgrp;
                if (m < M) {
                    float v = acc[a][c][r] * (F16 ? H3_LO_INV / TN2_BS : 1.0f);
                    if (jobs.drop_p > 0.0f) v = tn_dropout_keep(jobs.drop_seed, (size_t)m * N + n, jobs.drop_p) ? v * (1.0f / (1.0f - jobs.drop_p)) : 0.0f;
                    out[(size_t)m * N + n] = v;
                }
            }
        }
}

// out = (half0 + half1) [* dropout mask / (1 - p)]: the two K halves of a split layer-input gradient (float4 per thread; n4 = elements / 4)
static __global__ __launch_bounds__(256) void dx_halves_add_kernel(const float* __restrict__ halves, size_t n4, float* __restrict__ out, float drop_p,
                                                                   unsigned long long seed) {
    const float4* h0 = reinterpret_cast<const float4*>(halves);
    const float4* h1 = h0 + n4;
    const float sc = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 a = h0[i], b = h1[i];
        float4 v = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        if (drop_p > 0.0f) {
            v.x = tn_dropout_keep(seed, 4 * i, drop_p) ? v.x * sc : 0.0f;
            v.y = tn_dropout_keep(seed, 4 * i + 1, drop_p) ? v.y * sc : 0.0f;
            v.z = tn_dropout_keep(seed, 4 * i + 2, drop_p) ? v.z * sc : 0.0f;
            v.w = tn_dropout_keep(seed, 4 * i + 3, drop_p) ? v.w * sc : 0.0f;
        }
        reinterpret_cast<float4*>(out)[i] = v;
    }
}
