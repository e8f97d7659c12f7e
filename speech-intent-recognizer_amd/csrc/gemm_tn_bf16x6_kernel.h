// Token-reduction ("TN") GEMMs of the GRU weight gradients on the bf16 matrix cores (bf16x6, fp32 accuracy).
//
//   dW[m][n] = sum_tok A[tok][m] * B[tok (+shift)][n]       A = dgi / dgh slice [tokens][768], B = layer input or h_prev
// Both operands are token-major in memory, i.e. TRANSPOSED with respect to what v_mfma_f32_32x32x16_bf16 wants (eight
// consecutive k = tokens per lane).  As in conv_wgrad_bf16x6_kernel the transposition happens while a 32-token stage is
// written to LDS: thread = (token, 4 columns) reads a float4, splits it into the three bf16 planes and stores the 12
// halves with ds_write_b16 into AT[plane][m][token] / BT[plane][n][token] (lanes run along the tokens); the next stage's
// global loads are issued before the MFMAs of the current one.
// One launch covers up to four jobs (both directions x {W_ih, W_hh} of a layer): blockIdx.x walks the 128 x 256 output
// tiles of all jobs, blockIdx.y the K splits; every (tile, split) writes its partial to the job's slab z (deterministic
// slab_reduce afterwards).  8 waves, wave tile 64 x 64 (2 x 2 accumulators), 24 MFMAs per 16-token step.
// Tried and removed: 16-token stages in two LDS buffers with the next stage's transposing stores placed between the MFMAs
// (sched_group_barrier 2 MFMA : 10 VALU : 3 LDS stores) and one barrier per stage -- 153-158 us against 140 us for the
// dW launch and 167-178 against 147 us for dX: the 48-byte rows make the 2-byte stores 2-way bank-conflicted and the
// shorter MFMA runs between barriers expose more latency than the overlap hides.
//   seq / shift: row tok of B is taken from row tok + shift of the same length-`seq` sequence, zero outside it (the
//   h_{t-1} / h_{t+1} operand of the W_hh gradient), as in gemm_general_kernel.
#pragma once
#include "bf16x6_kernels.h"

constexpr int TN_BM = 128, TN_BN = 256, TN_BK = 32;
constexpr int TN_ROWB = TN_BK * 2 + 16;                      // 80 B per LDS row: 5 sixteen-byte slots (odd -> conflict-free b128 reads)
constexpr size_t TN_LDS_BYTES = (size_t)3 * (TN_BM + TN_BN) * TN_ROWB;   // 92,160 B
constexpr size_t TN_LDS_BYTES_64 = (size_t)3 * (64 + TN_BN) * TN_ROWB;    // 76,800 B (BM = 64)

struct TnJobs {
    const float* A[4]; const float* B[4]; float* slab[4];   // slab[j] + z * slab_stride[j] receives split z of job j
    const float* B2[4]; int brows[4];                        // rows k >= brows[j] of B come from B2[j] (two stacked matrices); 0 = off
    int lda[4], ldb[4], N[4], shift[4];
    size_t slab_stride[4];
    int tile0[5];                                            // first tile index of each job (prefix sums), tile0[njobs] = total
    int njobs;
};

// A_KM = true : A is [K][lda] (m contiguous: dW = dG^T X), staged transposed like B.
// A_KM = false: A is [M][lda] (k contiguous: dX = dG [W; W_reverse]), staged with one 8-byte store per plane.
// BM = 128: waves 2 x 4, wave tile 64 x 64.  BM = 64: waves 1 x 8, wave tile 64 x 32 -- twice the workgroups for outputs
// that would otherwise leave most CUs idle (layer-1 dX: 6400 x 512 is only 100 tiles of 128 x 256).
template <bool A_KM, int BM = TN_BM>
__global__ __launch_bounds__(512) void gemm_tn_bf16x6_kernel(TnJobs jobs, int M, int K, int kchunk, int seq) {
    static_assert(BM == 128 || BM == 64, "tile rows");
    constexpr int NC = BM == 128 ? 2 : 1;                    // 32-column accumulators per wave
    constexpr int NAQ = BM * 8 / 512;                        // A staging items per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char tl[];
    unsigned char* AT = tl;                                  // [3][BM][80]
    unsigned char* BT = tl + (size_t)3 * BM * TN_ROWB;       // [3][256][80]
    int j = 0;
    while (j + 1 < jobs.njobs && (int)blockIdx.x >= jobs.tile0[j + 1]) ++j;
    const int tile = blockIdx.x - jobs.tile0[j];
    const int N = jobs.N[j], lda = jobs.lda[j], ldb = jobs.ldb[j], shift = jobs.shift[j];
    const int ntn = (N + TN_BN - 1) / TN_BN;
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * TN_BN;
    const float* __restrict__ A = jobs.A[j];
    const float* __restrict__ B = jobs.B[j];
    const float* __restrict__ B2 = jobs.B2[j];
    const int brows = jobs.brows[j];
    const int k_begin = blockIdx.y * kchunk, k_end = min(K, k_begin + kchunk);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = BM == 128 ? wv >> 2 : 0, wn = BM == 128 ? wv & 3 : wv, i32 = lane & 31, kgrp = lane >> 5;

    // staging items: token = it % 32, 4-column group = it / 32.  A: 128 / 4 * 32 = 1024 items (2 per thread), B: 2048 (4)
    float4 pa[NAQ], pb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < NAQ; ++q) {
            const int it = tid + 512 * q;
            if (A_KM) {
                const int tok = k0 + (it & 31), m = m0 + 4 * (it >> 5);
                pa[q] = (tok < k_end && m < M) ? *reinterpret_cast<const float4*>(A + (size_t)tok * lda + m) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {                                         // item = (row it / 8, four consecutive k)
                const int m = m0 + (it >> 3), k = k0 + 4 * (it & 7);
                pa[q] = (k < k_end && m < M) ? *reinterpret_cast<const float4*>(A + (size_t)m * lda + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int it = tid + 512 * q, tok = k0 + (it & 31), n = n0 + 4 * (it >> 5);
            bool ok = tok < k_end && n < N;
            int src = tok;
            if (shift != 0) {                                // neighbouring time step of the same utterance
                const int t = tok % seq + shift;
                ok = ok && t >= 0 && t < seq;
                src = tok + shift;
            }
            const float* bsrc = (brows > 0 && src >= brows) ? B2 + (size_t)(src - brows) * ldb : B + (size_t)src * ldb;
            pb[q] = ok ? *reinterpret_cast<const float4*>(bsrc + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto put = [&](const float4& v, unsigned char* base, size_t plane, int row4, int tok) {
        uint2 hh, mm, ll;
        split3_quad(v, hh, mm, ll);
        unsigned char* d = base + (size_t)row4 * TN_ROWB + tok * 2;
        const unsigned hw[4] = {hh.x & 0xFFFFu, hh.x >> 16, hh.y & 0xFFFFu, hh.y >> 16};
        const unsigned mw[4] = {mm.x & 0xFFFFu, mm.x >> 16, mm.y & 0xFFFFu, mm.y >> 16};
        const unsigned lw[4] = {ll.x & 0xFFFFu, ll.x >> 16, ll.y & 0xFFFFu, ll.y >> 16};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            *reinterpret_cast<unsigned short*>(d + (size_t)e * TN_ROWB) = (unsigned short)hw[e];
            *reinterpret_cast<unsigned short*>(d + plane + (size_t)e * TN_ROWB) = (unsigned short)mw[e];
            *reinterpret_cast<unsigned short*>(d + 2 * plane + (size_t)e * TN_ROWB) = (unsigned short)lw[e];
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < NAQ; ++q) {
            const int it = tid + 512 * q;
            if (A_KM) put(pa[q], AT, (size_t)BM * TN_ROWB, 4 * (it >> 5), it & 31);
            else {
                uint2 hh, mm, ll;
                split3_quad(pa[q], hh, mm, ll);
                unsigned char* d = AT + (size_t)(it >> 3) * TN_ROWB + (it & 7) * 8;
                *reinterpret_cast<uint2*>(d) = hh;
                *reinterpret_cast<uint2*>(d + (size_t)BM * TN_ROWB) = mm;
                *reinterpret_cast<uint2*>(d + 2 * (size_t)BM * TN_ROWB) = ll;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int it = tid + 512 * q;
            put(pb[q], BT, (size_t)TN_BN * TN_ROWB, 4 * (it >> 5), it & 31);
        }
    };

    f32x16 acc[2][NC];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.0f;

    const unsigned char* arow = AT + (size_t)(wm * 64 + i32) * TN_ROWB + kgrp * 16;
    const unsigned char* brow = BT + (size_t)(wn * 32 * NC + i32) * TN_ROWB + kgrp * 16;
    fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += TN_BK) {
        __syncthreads();                                     // previous stage consumed
        stage();
        if (k0 + TN_BK < k_end) fetch(k0 + TN_BK);           // in flight during this stage's MFMAs
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[2][3], bf[NC][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    af[a][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(arow + (size_t)p * BM * TN_ROWB + (size_t)a * 32 * TN_ROWB + s * 32));
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    bf[c][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(brow + (size_t)p * TN_BN * TN_ROWB + (size_t)c * 32 * TN_ROWB + s * 32));
            }
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // small terms first
#pragma unroll
            for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][PA[t6]], bf[c][PB[t6]], acc[a][c], 0, 0, 0);
        }
    }
    float* out = jobs.slab[j] + (size_t)blockIdx.y * jobs.slab_stride[j];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int n = n0 + wn * 32 * NC + c * 32 + i32;
            if (n >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kgrp;
                if (m < M) out[(size_t)m * N + n] = acc[a][c][r];
            }
        }
}
