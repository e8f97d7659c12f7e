// Token-reduction ("TN") GEMMs of the GRU weight gradients on the bf16 matrix cores (bf16x6, fp32 accuracy).
//
//   dW[m][n] = sum_tok A[tok][m] * B[tok (+shift)][n]       A = dgi / dgh slice [tokens][768], B = layer input or h_prev
// Both operands are token-major in memory, i.e. TRANSPOSED with respect to what v_mfma_f32_32x32x16_bf16 wants (eight
// consecutive k = tokens per lane).  The transposition is done by the LDS hardware on the way OUT: a 32-token stage is
// written in its natural [token][column] order -- thread = (token, 4 columns) reads a float4 (lanes along the columns:
// 512-byte coalesced rows), splits it into the three bf16 planes and issues ONE ds_write_b64 per plane -- and the MFMA
// fragments are fetched with ds_read_b64_tr_b16 (gfx950), which hands lane i column i of a 4-token x 16-column block:
// two of them make the lane's eight consecutive tokens.  64-byte chunks of a row are XOR-swizzled with the token index so
// that the four rows of a transposed read fall on the four quarters of the bank line (conflict-free reads AND stores).
// The first version transposed on the way IN with twelve 2-byte LDS stores per float4 (lanes along the tokens): timing
// knock-outs put that staging at 67 of the dW launch's 214 us and 48 of dX's 167 us, serial with the MFMAs.
// One launch covers up to four jobs (both directions x {W_ih, W_hh} of a layer): blockIdx.x walks the 128 x 256 output
// tiles of all jobs, blockIdx.y the K splits; every (tile, split) writes its partial to the job's slab z (deterministic
// slab_reduce afterwards).  8 waves, wave tile 64 x 64 (2 x 2 accumulators), 24 MFMAs per 16-token step.
// Tried and removed: 16-token stages in two LDS buffers with the next stage's stores placed between the MFMAs
// (sched_group_barrier 2 MFMA : 10 VALU : 3 LDS stores) and one barrier per stage -- slower than the plain two-barrier loop.
//   seq / shift: row tok of B is taken from row tok + shift of the same length-`seq` sequence, zero outside it (the
//   h_{t-1} / h_{t+1} operand of the W_hh gradient).
#pragma once
#include "bf16x6_kernels.h"

constexpr int TN_BM = 128, TN_BN = 256, TN_BK = 32;
constexpr int TN_BM_DW = 256;                               // row tile of the weight-gradient launches (dW = dG^T X)
constexpr int TN_ROWB = TN_BK * 2 + 16;                      // k-contiguous image (A of dX): 80 B per row, 5 sixteen-byte slots (odd -> conflict-free b128 reads)
// k-major image of an operand given as [k][x] (x contiguous): per plane 32 rows of 2 X bytes, 64-byte chunks swizzled
constexpr size_t tn_lds_bytes(bool a_km, int bm) {
    return (size_t)3 * (a_km ? TN_BK * bm * 2 : bm * TN_ROWB) + (size_t)3 * TN_BK * TN_BN * 2;
}
constexpr size_t TN_LDS_BYTES = tn_lds_bytes(false, 128);    // 79,872 B: the larger of the two BM = 128 instantiations (dW: 73,728)
constexpr size_t TN_LDS_BYTES_64 = tn_lds_bytes(false, 64);  // 64,512 B (BM = 64)
constexpr size_t TN_LDS_BYTES_256 = tn_lds_bytes(true, 256); // 98,304 B (dW with 256-row tiles: one workgroup per CU)

// byte offset of element (row k, byte xb of the row) in a k-major image with XW-byte rows.  The four rows k0 .. k0+3 of a
// transposed read (k0 a multiple of 4) put their 64-byte chunk on four different quarters of the 256-byte bank line:
//   XW >= 256: chunk c -> c ^ (k & 3);   XW = 128: chunk c -> c ^ ((k >> 1) & 1)   (rows k and k + 2 share a line half)
template <int XW>
__device__ __forceinline__ int tn_kmaj_off(int k, int xb) {
    const int key = XW == 128 ? ((k >> 1) & 1) : (k & 3);
    return k * XW + ((((xb >> 6) ^ key)) << 6) + (xb & 63);
}
typedef short tn_v4i16 __attribute__((ext_vector_type(4)));
// eight consecutive k of one column as an MFMA fragment: two hardware-transposed reads (k .. k+3 and k+4 .. k+7), XW-byte rows
template <int XW>
__device__ __forceinline__ bf16x8 tn_tr_fragment(const unsigned char* p) {
    struct { tn_v4i16 lo, hi; } f;
    f.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tn_v4i16 __attribute__((address_space(3)))*)(p));
    f.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tn_v4i16 __attribute__((address_space(3)))*)(p + 4 * XW));
    return __builtin_bit_cast(bf16x8, f);
}

// keep mask of the inter-layer dropout (same function as train_kernels.h dropout_keep; the dX GEMM of layer 1 applies the
// dropout BACKWARD in its epilogue: its output IS d(dropout(y0)), and y0's gradient is that times the mask / (1 - p))
__device__ __forceinline__ bool tn_dropout_keep(unsigned long long seed, size_t idx, float p) {
    unsigned long long x = seed ^ (idx * 0x9E3779B97F4A7C15ull);
    x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull; x ^= x >> 33;
    return (float)(unsigned)(x >> 40) * (1.0f / 16777216.0f) >= p;
}

struct TnJobs {
    const float* A[4]; const float* B[4]; float* slab[4];   // slab[j] + z * slab_stride[j] receives split z of job j
    const float* B2[4]; int brows[4];                        // rows k >= brows[j] of B come from B2[j] (two stacked matrices); 0 = off
    int lda[4], ldb[4], N[4], shift[4];
    size_t slab_stride[4];
    int tile0[5];                                            // first tile index of each job (prefix sums), tile0[njobs] = total
    int njobs;
    float drop_p;                                            // > 0: out[m][n] *= keep(drop_seed, m * N + n) / (1 - drop_p)
    unsigned long long drop_seed;
    const float* zeros;                                      // >= 256 zero floats (the handle's zero page): B rows that must read as zero (gemm_tn2)
};

// A_KM = true : A is [K][lda] (m contiguous: dW = dG^T X), staged transposed like B.
// A_KM = false: A is [M][lda] (k contiguous: dX = dG [W; W_reverse]), staged with one 8-byte store per plane.
// BM = 128: waves 2 x 4, wave tile 64 x 64.  BM = 64: waves 1 x 8, wave tile 64 x 32 -- twice the workgroups for outputs
// that would otherwise leave most CUs idle (layer-1 dX: 6400 x 512 is only 100 tiles of 128 x 256).
template <bool A_KM, int BM = TN_BM>
__global__ __launch_bounds__(512) void gemm_tn_bf16x6_kernel(TnJobs jobs, int M, int K, int kchunk, int seq) {
    static_assert(BM == 256 || BM == 128 || BM == 64, "tile rows");
    constexpr int NC = BM >= 128 ? 2 : 1;                    // 32-column accumulators per wave
    constexpr int NA = BM == 256 ? 4 : 2, WMR = 32 * NA;     // 32-row accumulators per wave; rows of a wave tile
    constexpr int NAQ = BM * 8 / 512;                        // A staging items per thread
    constexpr int AXW = BM * 2, BXW = TN_BN * 2;             // row bytes of the k-major images
    constexpr int APLANE = A_KM ? TN_BK * AXW : BM * TN_ROWB, BPLANE = TN_BK * BXW;
    extern __shared__ __attribute__((aligned(16))) unsigned char tl[];
    unsigned char* AT = tl;                                  // A_KM: [3][32 tok][BM] k-major, swizzled; else [3][BM][80] k-contiguous
    unsigned char* BT = tl + (size_t)3 * APLANE;             // [3][32 k][256] k-major, swizzled
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
    const int wm = BM >= 128 ? wv >> 2 : 0, wn = BM >= 128 ? wv & 3 : wv, i32 = lane & 31, kgrp = lane >> 5;

    // staging items, lanes along the columns (coalesced 16-byte loads, 128 contiguous LDS bytes per 16-lane store group):
    //   A_KM : A item = (4-column group it % (BM/4), token it / (BM/4)); else A item = (row it / 8, four consecutive k)
    //   B    : item = (4-column group it % 64, token it / 64)
    float4 pa[NAQ], pb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < NAQ; ++q) {
            const int it = tid + 512 * q;
            if (A_KM) {
                const int tok = k0 + it / (BM / 4), m = m0 + 4 * (it % (BM / 4));
                pa[q] = (tok < k_end && m < M) ? *reinterpret_cast<const float4*>(A + (size_t)tok * lda + m) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                const int m = m0 + (it >> 3), k = k0 + 4 * (it & 7);
                pa[q] = (k < k_end && m < M) ? *reinterpret_cast<const float4*>(A + (size_t)m * lda + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int it = tid + 512 * q, tok = k0 + (it >> 6), n = n0 + 4 * (it & 63);
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
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < NAQ; ++q) {
            const int it = tid + 512 * q;
            uint2 hh, mm, ll;
            split3_quad(pa[q], hh, mm, ll);
            unsigned char* d = A_KM ? AT + tn_kmaj_off<AXW>(it / (BM / 4), 8 * (it % (BM / 4)))
                                    : AT + (size_t)(it >> 3) * TN_ROWB + (it & 7) * 8;
            *reinterpret_cast<uint2*>(d) = hh;
            *reinterpret_cast<uint2*>(d + APLANE) = mm;
            *reinterpret_cast<uint2*>(d + 2 * APLANE) = ll;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int it = tid + 512 * q;
            uint2 hh, mm, ll;
            split3_quad(pb[q], hh, mm, ll);
            unsigned char* d = BT + tn_kmaj_off<BXW>(it >> 6, 8 * (it & 63));
            *reinterpret_cast<uint2*>(d) = hh;
            *reinterpret_cast<uint2*>(d + BPLANE) = mm;
            *reinterpret_cast<uint2*>(d + 2 * BPLANE) = ll;
        }
    };

    f32x16 acc[NA][NC];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.0f;

    // fragment addresses.  Transposed read (cdna_hip_programming.md T10): the 16 lanes 16 g .. 16 g + 15 fetch a block of
    // 4 k-rows x 16 columns; lane 4 q + p supplies the address of row q, columns 4 p .. 4 p + 3 and receives column
    // (lane & 15).  For the 32 x 32 x 16 operand, group g covers columns 16 (g & 1) .. + 15 of the wave's 32-column tile
    // and k = 8 (g >> 1) .. + 7 of the 16-deep step (two reads: k .. k + 3, k + 4 .. k + 7).
    const int tq = (lane >> 2) & 3, tp = lane & 3, tcol = 16 * ((lane >> 4) & 1) + 4 * tp, tk = 8 * (lane >> 5) + tq;
    const unsigned char* arow[NA];
    const unsigned char* brow[NC];
#pragma unroll
    for (int a = 0; a < NA; ++a)
        arow[a] = A_KM ? AT + tn_kmaj_off<AXW>(tk, 2 * (wm * WMR + a * 32 + tcol))
                       : AT + (size_t)(wm * WMR + a * 32 + i32) * TN_ROWB + kgrp * 16;
#pragma unroll
    for (int c = 0; c < NC; ++c) brow[c] = BT + tn_kmaj_off<BXW>(tk, 2 * (wn * 32 * NC + c * 32 + tcol));
    fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += TN_BK) {
        __syncthreads();                                     // previous stage consumed
        stage();
        __syncthreads();
        // issued AFTER the barrier (a __syncthreads() in front of the loads' consumers would wait for them: it drains vmcnt):
        // in flight during this stage's MFMAs, waited for at the next stage's first barrier
        if (k0 + TN_BK < k_end) fetch(k0 + TN_BK);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[NA][3], bf[NC][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int a = 0; a < NA; ++a)
                    af[a][p] = A_KM ? tn_tr_fragment<AXW>(arow[a] + p * APLANE + s * 16 * AXW)
                                    : __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(arow[a] + (size_t)p * APLANE + s * 32));
#pragma unroll
                for (int c = 0; c < NC; ++c) bf[c][p] = tn_tr_fragment<BXW>(brow[c] + p * BPLANE + s * 16 * BXW);
            }
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // small terms first
#pragma unroll
            for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][PA[t6]], bf[c][PB[t6]], acc[a][c], 0, 0, 0);
        }
    }
    float* out = jobs.slab[j] + (size_t)blockIdx.y * jobs.slab_stride[j];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int n = n0 + wn * 32 * NC + c * 32 + i32;
            if (n >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WMR + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kgrp;
                if (m < M) {
                    float v = acc[a][c][r];
                    if (jobs.drop_p > 0.0f) v = tn_dropout_keep(jobs.drop_seed, (size_t)m * N + n, jobs.drop_p) ? v * (1.0f / (1.0f - jobs.drop_p)) : 0.0f;
                    out[(size_t)m * N + n] = v;
                }
            }
        }
}
