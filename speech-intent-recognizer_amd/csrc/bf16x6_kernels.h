// fp32-accurate contractions on the bf16 matrix cores ("bf16x6").
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate.  An fp32 value splits EXACTLY into
// three bf16 numbers x = hi + mid + lo (8+8+8 mantissa bits); of the nine cross products of two
// such splits the six with weight >= 2^-16 (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid) carry the
// product to ~2^-24 relative, i.e. fp32 accuracy, and accumulate in the MFMA's f32 accumulator.
// Six v_mfma_f32_32x32x16_bf16 (K = 16, 32 cycles each) replace eight f32 MFMAs (K = 2, 64 cycles
// each) per 16-deep K step: 192 vs 512 matrix-pipe cycles.  Operands are split once, outside the
// GEMM, into three bf16 planes (6 bytes per value).
#pragma once
#include <type_traits>
#include "model_kernels.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef __bf16 sir_bf16x2 __attribute__((ext_vector_type(2)));
typedef float sir_f32x2 __attribute__((ext_vector_type(2)));

// exact three-way split of two floats at once; packed results (a in the low half).  The conversions
// compile to v_cvt_pk_bf16_f32 (round to nearest even) and the residuals to v_pk_add_f32: 9 VALU
// instructions per pair.
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    sir_f32x2 v = {a, b};
    const sir_bf16x2 hi = __builtin_convertvector(v, sir_bf16x2);
    v -= __builtin_convertvector(hi, sir_f32x2);            // exact
    const sir_bf16x2 mid = __builtin_convertvector(v, sir_bf16x2);
    v -= __builtin_convertvector(mid, sir_f32x2);           // exact
    const sir_bf16x2 lo = __builtin_convertvector(v, sir_bf16x2);
    h = __builtin_bit_cast(unsigned, hi);
    m = __builtin_bit_cast(unsigned, mid);
    l = __builtin_bit_cast(unsigned, lo);
}
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    unsigned ph, pm, pl;
    split3_pair(x, 0.0f, ph, pm, pl);
    h = (unsigned short)ph; m = (unsigned short)pm; l = (unsigned short)pl;
}
// float4 -> one 8-byte group per plane
__device__ __forceinline__ void split3_quad(const float4& v, uint2& h, uint2& m, uint2& l) {
    split3_pair(v.x, v.y, h.x, m.x, l.x);
    split3_pair(v.z, v.w, h.y, m.y, l.y);
}

// in [rows][K] fp32 (row stride ld_in) -> planes [3][rows][K] bf16; one thread = 8 consecutive k
// (gidx / nthreads: linear thread index and thread count of the launch -- or of the job's share of a multi-job launch)
__device__ __forceinline__ void split3_rows(const float* __restrict__ in, int ld_in, unsigned short* __restrict__ out, size_t rows, int K,
                                            size_t gidx, size_t nthreads) {
    const int k8n = K / 8;
    const size_t total = rows * k8n, plane = rows * (size_t)K;
    for (size_t idx = gidx; idx < total; idx += nthreads) {
        const size_t row = idx / k8n;
        const int k8 = idx % k8n;
        const float4 v0 = *reinterpret_cast<const float4*>(in + row * ld_in + k8 * 8);
        const float4 v1 = *reinterpret_cast<const float4*>(in + row * ld_in + k8 * 8 + 4);
        uint2 h0, m0, l0, h1, m1, l1;
        split3_quad(v0, h0, m0, l0);
        split3_quad(v1, h1, m1, l1);
        const size_t o = row * K + (size_t)k8 * 8;
        *reinterpret_cast<uint4*>(out + o) = make_uint4(h0.x, h0.y, h1.x, h1.y);
        *reinterpret_cast<uint4*>(out + plane + o) = make_uint4(m0.x, m0.y, m1.x, m1.y);
        *reinterpret_cast<uint4*>(out + 2 * plane + o) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}
#include "f16_split.h"

static __global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ in, int ld_in, unsigned short* __restrict__ out,
                                                             size_t rows, int K) {
    split3_rows(in, ld_in, out, rows, K, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
}

// C[m][z*N + n] = sum_k A[m][k] * Bz[n][k] + biasz[n] with A, B given as bf16x3 planes.
//   Ap: [3][M][K], Bp0/Bp1: [3][N][K] (z = blockIdx.z).  128 x 64 tile, BK = 32 (two 16-deep MFMA
//   steps), 4 waves 2x2, wave tile 64 x 32.  LDS rows are 64 B of data + 16 B pad = 80 B: five 16-byte
//   slots per row make the ds_read_b128 fragment reads conflict-free.
constexpr int XB_ROW = 80;                                  // bytes per LDS row

// ------------------------------------------------------------------------------------------
// Second generation of the bf16x6 GEMM (same contract as gemm_nt_bf16x6_kernel): 8 waves (two per
// SIMD), tile 160 rows x 256 columns, BK = 32.  Both operands are staged by LDS-DMA
// (global_load_lds_dwordx4: no VGPR staging, no ds_write pass) in pieces of 16 rows x 64 B per wave
// instruction; fragment-shaped register loads of B (32 rows x 16 B per instruction) were tried and
// cost the vector-memory address path ~28 us per launch.  Stage = A [3][160][64 B] + B [3][256][64 B] = 79,872 B, two stages.
// Swizzle on the source address: chunk c of row r is stored at chunk c ^ ((r >> 2) & 3) (four 64-B
// rows share a 256-B bank line), conflict-free for the ds_read_b128 lane groups.  A wave owns a
// 160 x 32 strip (five accumulators), reads 15 A + 3 B fragments per 16-deep step for 30 MFMAs; the
// second wave on the SIMD covers the LDS latency.  The next tile's DMA pieces are issued between
// the MFMAs (a piece costs the issuing wave ~100 cycles): waves 0-3 during step 0, their SIMD partners
// 4-7 during step 1, so one wave per SIMD always feeds the matrix pipe.  blockIdx is remapped so that
// the column blocks sharing an A row block run on the same XCD.  240 workgroups for the 6400 x 1536
// projections: one round on 256 CUs.  Measured (devtools/kernel_ab/bench_gemm.hip, random operands): 107 us at
// K = 1024 (188 TF algorithmic) against 159 us for the first kernel; the kernel without any staging
// runs 94 us and a pure-MFMA probe of the same instruction mix 69 us (1.75 PF executed).
// ------------------------------------------------------------------------------------------
constexpr int G3_BM = 160, G3_BN = 256, G3_BK = 32;
constexpr int G3_APLANE = G3_BM * 64, G3_BPLANE = G3_BN * 64;
constexpr int G3_BOFF = 3 * G3_APLANE;                      // 30,720
constexpr int G3_STAGE = G3_BOFF + 3 * G3_BPLANE;           // 79,872
constexpr int G3_LDS_BYTES = 2 * G3_STAGE;                  // 159,744
constexpr int G3_APIECES = 3 * G3_BM / 16, G3_PIECES = G3_APIECES + 3 * G3_BN / 16;   // 30, 78
typedef __attribute__((address_space(1))) const void* sir_gptr_t;
typedef __attribute__((address_space(3))) void* sir_lptr_t;
constexpr int G3_DEFAULT = 16;                              // product configuration of the KNOCK template word: SPREAD = 2

// KNOCK: experiment word of devtools/kernel_ab/bench_gemm.hip.  bit 0 = no staging, bit 2 = no MFMAs (timing only), bits 3-4 = SPREAD,
// bit 5 = PREF (see below)
template <int KNOCK = G3_DEFAULT>
static __global__ __launch_bounds__(512) void gemm_nt_bf16x6_v3_kernel(
    const unsigned short* __restrict__ Ap, const unsigned short* __restrict__ Bp0, const unsigned short* __restrict__ Bp1,
    const float* __restrict__ bias0, const float* __restrict__ bias1, float* __restrict__ C, int ldc, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char g3_smem[];
    const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, rem = nwg & 7;
    const int wgid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (orig >> 3);
    const int nbd = N / G3_BN, nb = 2 * nbd;
    const int mblk = wgid / nb, nbk = wgid - mblk * nb, z = nbk / nbd;
    const int m0 = mblk * G3_BM, n0 = (nbk - z * nbd) * G3_BN;
    const unsigned short* __restrict__ Bp = z ? Bp1 : Bp0;
    const float* __restrict__ bias = z ? bias1 : bias0;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m = lane & 31, h = lane >> 5;
    const size_t planeA = (size_t)M * K, planeB = (size_t)N * K;

    // LDS-DMA pieces of this wave: g = wv + 8 i (i < 10, g < 78); pieces 0..29 = A (plane g / 10, rows 16 (g % 10)..),
    // 30..77 = B (plane (g - 30) / 16, rows 16 ((g - 30) % 16)..)
    const int lr = lane >> 2;
    const int csrc = (lane & 3) ^ ((lr >> 2) & 3);
    unsigned int poff[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int g = wv + 8 * i;
        if (g < G3_APIECES) {
            const int pl = g / 10, rg = g - pl * 10;
            int row = m0 + rg * 16 + lr;
            row = row < M ? row : M - 1;
            poff[i] = (unsigned int)(pl * planeA + (size_t)row * K + csrc * 8);
        } else {
            const int gb = g - G3_APIECES, pl = gb >> 4, rg = gb & 15;
            poff[i] = (unsigned int)(pl * planeB + (size_t)(n0 + rg * 16 + lr) * K + csrc * 8);
        }
    }
    auto piece = [&](int i, int kt, int buf) {
        if (KNOCK & 1) return;
        const int g = wv + 8 * i;
        if (g < G3_PIECES) {
            const unsigned short* src = (g < G3_APIECES ? Ap : Bp) + poff[i] + (size_t)kt * G3_BK;
            __builtin_amdgcn_global_load_lds((sir_gptr_t)src, (sir_lptr_t)(g3_smem + buf * G3_STAGE + g * 1024), 16, 0, 0);
        }
    };
    auto stage = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < 10; ++i) piece(i, kt, buf);
    };

    f32x16 acc[5];
#pragma unroll
    for (int mt = 0; mt < 5; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;

    int fo[2];                                              // byte offset of this lane's chunk inside its row, per MFMA step
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) fo[ks] = m * 64 + ((((ks << 1) | h) ^ ((m >> 2) & 3)) << 4);

    // SPREAD (KNOCK bits 3-4): 0 = the next tile's DMA pieces are issued in bulk before the MFMAs, 1 = one piece after
    // every third MFMA of step 0, 2 = waves 0-3 spread over step 0 and waves 4-7 (their SIMD partners) over step 1.
    // PREF (bit 5): both steps' fragments are read before the first MFMA (pinned with sched_barrier).
    constexpr int SPREAD = (KNOCK >> 3) & 3;
    constexpr bool PREF = (KNOCK >> 5) & 1;
    const int nk = K / G3_BK;
    bf16x8 a[2][3][5], bb[2][3];
    auto read_frags = [&](int buf, int ks) {
        const unsigned char* abase = g3_smem + buf * G3_STAGE;
        const unsigned char* bbase = abase + G3_BOFF + wv * 2048;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            bb[ks][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(bbase + p * G3_BPLANE + fo[ks]));
#pragma unroll
            for (int mt = 0; mt < 5; ++mt)
                a[ks][p][mt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(abase + p * G3_APLANE + mt * 2048 + fo[ks]));
        }
    };
    auto compute = [&](int buf, int ktn) {
        // past the last tile the pieces re-load the last tile into the idle buffer (keeps the loop branch-free)
        const int ktl = ktn < nk ? ktn : nk - 1;
        read_frags(buf, 0);
        if (PREF) {
            read_frags(buf, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (!PREF && ks == 1) read_frags(buf, 1);
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // small terms first
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int mt = 0; mt < 5; ++mt) {
                    if (KNOCK & 4) acc[mt][0] += __builtin_bit_cast(float4, a[ks][PA[t]][mt]).x * __builtin_bit_cast(float4, bb[ks][PB[t]]).x;
                    else acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][PA[t]][mt], bb[ks][PB[t]], acc[mt], 0, 0, 0);
                    const int idx = t * 5 + mt;
                    if (SPREAD && idx % 3 == 2) {
                        if (SPREAD == 1 && ks == 0) piece(idx / 3, ktl, buf ^ 1);
                        if (SPREAD == 2 && ks == (wv >> 2)) piece(idx / 3, ktl, buf ^ 1);
                    }
                }
        }
    };

    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                                    // vmcnt(0) + barrier: tile kt landed, the other buffer is free
        if (SPREAD == 0 && kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        compute(kt & 1, kt + 1);
    }

    const int n = n0 + wv * 32 + m;
    const float bv = bias ? bias[n] : 0.0f;
    float* crow = C + (size_t)(m0 + 4 * h) * ldc + (size_t)z * N + n;
    if (m0 + G3_BM <= M) {                                  // whole tile in range: straight-line stores
#pragma unroll
        for (int mt = 0; mt < 5; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) crow[(size_t)(mt * 32 + (r & 3) + 8 * (r >> 2)) * ldc] = acc[mt][r] + bv;
    } else {
#pragma unroll
        for (int mt = 0; mt < 5; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = mt * 32 + (r & 3) + 8 * (r >> 2);
                if (m0 + 4 * h + ro < M) crow[(size_t)ro * ldc] = acc[mt][r] + bv;
            }
    }
}

// host side of the two GEMM generations: the second needs N % 256 == 0, K % 32 == 0 and 32-bit plane offsets
static inline bool gemm_bf16x6_v3_ok(int M, int N, int K) {
    return N % G3_BN == 0 && K % G3_BK == 0 && (size_t)3 * M * K < ((size_t)1 << 31) && (size_t)3 * N * K < ((size_t)1 << 31);
}
// C[M][2 N] (+ bias) = A x [B0; B1]^T from pre-split bf16x3 planes (the GRU input projections: N = 768 per direction)
static inline hipError_t launch_gemm_nt_bf16x6(sir_handle* h, hipStream_t st, const unsigned short* Ap, const unsigned short* Bp0,
                                               const unsigned short* Bp1, const float* bias0, const float* bias1, float* C, int ldc,
                                               int M, int N, int K) {
    if (!gemm_bf16x6_v3_ok(M, N, K)) return hipErrorInvalidValue;
    if (!h->attr_gemm_v3) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_bf16x6_v3_kernel<G3_DEFAULT>, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES);
        if (e != hipSuccess) return e;
        h->attr_gemm_v3 = true;
    }
    const int nwg = ((M + G3_BM - 1) / G3_BM) * 2 * (N / G3_BN);
    hipLaunchKernelGGL(gemm_nt_bf16x6_v3_kernel<G3_DEFAULT>, dim3(nwg), dim3(512), G3_LDS_BYTES, st, Ap, Bp0, Bp1, bias0, bias1, C, ldc, M, N, K);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// conv 3x3 + BN + ReLU + 2x2 max-pool as an implicit GEMM on the bf16 matrix cores with the
// bf16x6 split (same tiling idea as conv3x3_mfma_kernel: M = pixels in 8x4 patches, N = cout,
// K = (tap, ci); pooling in registers).  The fp32 input tile is split into three bf16 planes while
// it is staged into LDS (16 input channels per chunk, 48-byte pixel rows); the weights arrive
// pre-split from prep_conv_w_bf16x3_kernel as wpb[plane][g = ci/16*9 + tap][co][16].
// ------------------------------------------------------------------------------------------
// LDS image of the channel-split kernel (conv3x3_bf16x6_ns_kernel): per plane and tile row, the FIRST 16-byte halves
// (channels 0-7 of the 16-channel chunk) of all 4 PC + 2 pixels, then their SECOND halves -- row = 2 (4 PC + 2)
// sixteen-byte slots = 4 (mod 8) slots, no padding.  A ds_read_b128 serves the lanes in the groups {0-3,12-15,20-27},
// {4-11,16-19,28-31} (+32 for the second half-wave); with the 8x4 patch bit layout of an A fragment a group is
// rows {0,1,6,7} x columns {0,1} + rows {2,3,4,5} x columns {2,3} (or the complement), and slot = 4 row + column
// (mod 16) gives those 16 lanes 16 different slots of the 256-byte bank line for every tap and patch offset.  The
// pixel-major image (32- or padded 48-byte pixels) it replaces had 2-way conflicts on two lane pairs per group
// (PMC r01: SQ_LDS_BANK_CONFLICT = 50 % of the LDS cycles of conv2) unless every row was padded, which cost the third
// workgroup per CU.  This one is both conflict-free and a third smaller (conv2: 32,640 B against 48,960 B).
constexpr int conv_ns_row_bytes(int PC) { return 2 * (4 * PC + 2) * 16; }
constexpr size_t conv_ns_lds_bytes(int PR, int PC, int bufs = 1) { return (size_t)bufs * 3 * (8 * PR + 2) * conv_ns_row_bytes(PC); }

__device__ __forceinline__ void prep_conv_w_bf16x3_elem(const float* __restrict__ w, unsigned short* __restrict__ wpb, int cin, int cout, int idx) {
    const int total = cin * 9 * cout;
    if (idx >= total) return;
    const int e = idx & 15, co = (idx >> 4) % cout, g = (idx >> 4) / cout;
    const int ci = (g / 9) * 16 + e, tap = g % 9;
    unsigned short h, m, l;
    split3(w[((size_t)co * cin + ci) * 9 + tap], h, m, l);
    wpb[idx] = h;
    wpb[(size_t)total + idx] = m;
    wpb[2 * (size_t)total + idx] = l;
}
static __global__ void prep_conv_w_bf16x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ wpb, int cin, int cout) {
    prep_conv_w_bf16x3_elem(w, wpb, cin, cout, blockIdx.x * blockDim.x + threadIdx.x);
}

// data-gradient weights as bf16x3 planes: roles of the channel axes swapped, taps flipped (cf. prep_conv_wT_kernel);
// output channels co' = forward INPUT channels, 16-groups over the forward OUTPUT channels
__device__ __forceinline__ void prep_conv_wT_bf16x3_elem(const float* __restrict__ w, unsigned short* __restrict__ wpb, int cin_f, int cout_f, int idx) {
    const int total = cin_f * 9 * cout_f;
    if (idx >= total) return;
    const int e = idx & 15, cop = (idx >> 4) % cin_f, g = (idx >> 4) / cin_f;
    const int co_f = (g / 9) * 16 + e, tap = 8 - (g % 9);
    unsigned short h, m, l;
    split3(w[((size_t)co_f * cin_f + cop) * 9 + tap], h, m, l);
    wpb[idx] = h;
    wpb[(size_t)total + idx] = m;
    wpb[2 * (size_t)total + idx] = l;
}
static __global__ void prep_conv_wT_bf16x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ wpb, int cin_f, int cout_f) {
    prep_conv_wT_bf16x3_elem(w, wpb, cin_f, cout_f, blockIdx.x * blockDim.x + threadIdx.x);
}

// ------------------------------------------------------------------------------------------
// Second generation of the bf16x6 convolution: output channels split ACROSS the waves.
//   conv3x3_bf16x6_kernel gives every wave all COUT channels of its own pixel patches, so the four
//   waves of a workgroup each stream the full weight set of a tap from L2 (12 KiB per wave and tap
//   for 64 -> 128 channels: ~1.8 GB of L2 reads per launch at batch 256).  Here a wave owns ONE
//   32-channel slice (wn = wave % WN, WN = COUT / 32) and MT = PR * PC / (4 / WN) pixel patches: its
//   weight fragments are private (3 x 16 B per lane and tap, prefetched one tap ahead) and the pixel
//   fragments, which every wave needs, come from the shared LDS tile.  Same tile, LDS image, MFMA
//   order per accumulator and epilogues as the first kernel: results are bit-identical.
//   KNOCK (devtools/kernel_ab/bench_conv.hip timing experiments; 0 in the product): bit 0 = weights loaded once,
//   bit 1 = input tile staged once.
// ------------------------------------------------------------------------------------------
// MINB = workgroups per CU the register allocation must allow (3 where the accumulators leave room)
template <int CIN, int COUT, int PR, int PC, int OUT_MODE, int KNOCK = 0, int MINB = 2, int PIPE = 1, int DB = 0>
__global__ __launch_bounds__(256, MINB) void conv3x3_bf16x6_ns_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ wpb, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ out, int H, int W, int Hp, int Wp, float2* __restrict__ stats) {
    constexpr int WN = COUT / 32, WM = 4 / WN, MT = PR * PC / WM, CK = 16, PSB = 16;   // PSB: bytes between neighbouring pixels of a half-row
    constexpr int GS = MT <= 5 ? MT : 4;                    // patches per MFMA group (independent accumulators in flight)
    constexpr int TR = 8 * PR, TC = 4 * PC, TROWS = TR + 2, TCOLS = TC + 2;
    constexpr int RSB = conv_ns_row_bytes(PC), HSB = TCOLS * 16;   // row stride; offset of the second 16-byte halves inside a row
    constexpr int PLANE = TROWS * RSB;
    constexpr int G = (CIN / 16) * 9;
    static_assert(COUT % 32 == 0 && WN <= 4 && 4 % WN == 0 && (PR * PC) % WM == 0 && MT % GS == 0 && CIN % CK == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int b = blockIdx.z, ty0 = blockIdx.y * TR, tx0 = blockIdx.x * TC;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wn = wv % WN, wm = wv / WN;
    const int m = lane & 31, h = lane >> 5;
    const int pxl = (m & 1) + 2 * ((m >> 2) & 1);
    const int pyl = ((m >> 1) & 1) + 2 * ((m >> 3) & 1) + 4 * ((m >> 4) & 1);
    // patch pi = wm * MT + mt: row block pi % PR, column block pi / PR (column blocks ascend with mt)
    const int lane_off = pyl * RSB + pxl * PSB + h * HSB;
    int nvalid = 0;                                         // patches of this wave that start inside the image (a prefix)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) nvalid += (tx0 + 4 * ((wm * MT + mt) / PR) < W) ? 1 : 0;
    nvalid = __builtin_amdgcn_readfirstlane(nvalid);
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
    const uint4* wp4 = reinterpret_cast<const uint4*>(wpb) + (size_t)(wn * 32 + m) * 2 + h;   // + ((p*G + g)*COUT)*2
    const float* xb = x + (size_t)b * H * W * CIN;
    auto load_w = [&](int g, uint4 (&wf)[3]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) wf[p] = wp4[((size_t)p * G + g) * COUT * 2];
    };

    // DB: the LDS image is double-buffered and chunk cc + 1 is staged WHILE the taps of chunk cc run -- its global loads are
    // issued before tap 0 (into NIT float4 registers per thread), split and written to the other buffer after tap 4 -- with
    // one barrier per chunk.  Without it every workgroup of a CU (they start together and run in lockstep) sits in its
    // load -> split -> ds_write phase at the same time and the matrix pipe idles (timing knock-out "tile staged once").
    constexpr int NCH = CIN / CK, NITEMS = TROWS * TCOLS * 4, NIT = (NITEMS + 255) / 256, TILEB = 3 * PLANE;
    int goff[DB ? NIT : 1], doff[DB ? NIT : 1];             // per staging item: element offset in the image (-1: padding), byte offset in the tile (-1: none)
    float4 pre[DB ? NIT : 1];
    if constexpr (DB) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int idx = tid + 256 * k;
            const int hsel = idx / (2 * TROWS * TCOLS), rem = idx - hsel * (2 * TROWS * TCOLS);
            const int pix = rem >> 1, part = 2 * hsel + (rem & 1);
            const int tyy = pix / TCOLS, txx = pix - tyy * TCOLS;
            const int gy = ty0 - 1 + tyy, gx = tx0 - 1 + txx;
            const bool item = idx < NITEMS;
            goff[k] = (item && gy >= 0 && gy < H && gx >= 0 && gx < W) ? (gy * W + gx) * CIN + part * 4 : -1;
            doff[k] = item ? tyy * RSB + txx * PSB + (part >> 1) * HSB + (part & 1) * 8 : -1;
        }
    }
    auto stage_load = [&](int cc) {
#pragma unroll
        for (int k = 0; k < (DB ? NIT : 0); ++k) {
            pre[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (goff[k] >= 0) pre[k] = *reinterpret_cast<const float4*>(xb + goff[k] + cc * CK);
        }
    };
    auto stage_store_item = [&](unsigned char* buf, int k) {
        uint2 hh, mm, ll;
        split3_quad(pre[k], hh, mm, ll);
        if (doff[k] >= 0) {
            unsigned char* d = buf + doff[k];
            *reinterpret_cast<uint2*>(d) = hh;
            *reinterpret_cast<uint2*>(d + PLANE) = mm;
            *reinterpret_cast<uint2*>(d + 2 * PLANE) = ll;
        }
    };
    auto stage_store = [&](unsigned char* buf) {
#pragma unroll
        for (int k = 0; k < (DB ? NIT : 0); ++k) stage_store_item(buf, k);
    };
    static_assert(!DB || NIT <= 8, "one staging item per tap");

    uint4 wcur[3], wnext[3];
    load_w(0, wcur);
    if constexpr (DB) {
        stage_load(0);
        stage_store(ldsb);
        __syncthreads();
    }
    for (int cc = 0; cc < NCH; ++cc) {
        const unsigned char* tile = ldsb + (DB ? (cc & 1) * TILEB : 0);
        unsigned char* tile_next = ldsb + ((cc + 1) & 1) * TILEB;
        const bool stage_next = DB && cc + 1 < NCH && !(KNOCK & 2);
        if (!DB && (!(KNOCK & 2) || cc == 0)) {
        if (cc) __syncthreads();
        for (int idx = tid; idx < TROWS * TCOLS * 4; idx += 256) {
            // 16 consecutive lanes = 8 neighbouring pixels x the two 8-byte halves of ONE 16-byte slot class (hsel): their
            // ds_write_b64 group covers 128 contiguous bytes (lanes that mixed both classes met on 8 of the 32 banks)
            const int hsel = idx / (2 * TROWS * TCOLS), rem = idx - hsel * (2 * TROWS * TCOLS);
            const int pix = rem >> 1, part = 2 * hsel + (rem & 1);
            const int tyy = pix / TCOLS, txx = pix - tyy * TCOLS;
            const int gy = ty0 - 1 + tyy, gx = tx0 - 1 + txx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const float4*>(xb + ((size_t)gy * W + gx) * CIN + cc * CK + part * 4);
            uint2 hh, mm, ll;
            split3_quad(v, hh, mm, ll);
            unsigned char* d = ldsb + tyy * RSB + txx * PSB + (part >> 1) * HSB + (part & 1) * 8;
            *reinterpret_cast<uint2*>(d) = hh;
            *reinterpret_cast<uint2*>(d + PLANE) = mm;
            *reinterpret_cast<uint2*>(d + 2 * PLANE) = ll;
        }
        __syncthreads();
        }
        if constexpr (PIPE && MT == GS) {
            // One patch group per tap: the pixel fragments are software-pipelined across the taps.  The lo and hi planes of
            // tap t + 1 are read while the MFMAs of tap t run (lo is free after the first term, hi needs a second register
            // set), the mid plane at the top of its own tap, two terms before its first use: every ds_read_b128 has >= 4
            // MFMAs of this wave between issue and use instead of none.  Same MFMA order per accumulator as the plain loop.
            // NP = patches this wave computes: all MT, or only the first half where the rest starts right of the image
            // (the last tile column of a 50-wide map: column blocks ascend with the patch index)
            auto taps = [&](auto npc) {
                constexpr int NP = decltype(npc)::value;
                auto rd = [&](int tap, int p, bf16x8 (&a)[NP]) {
                    const unsigned char* tb = tile + lane_off + (tap / 3) * RSB + (tap % 3) * PSB + p * PLANE;
#pragma unroll
                    for (int i = 0; i < NP; ++i) {
                        const int pi = wm * MT + i, poff = 8 * (pi % PR) * RSB + 4 * (pi / PR) * PSB;
                        a[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(tb + poff));
                    }
                };
                bf16x8 alo[NP], ahi[NP], amid[NP], ahin[NP];
                rd(0, 2, alo);
                rd(0, 0, ahi);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int g = cc * 9 + tap;
                    if (tap == 0 && stage_next) stage_load(cc + 1);
                    if (!(KNOCK & 1) && g + 1 < G) load_w(g + 1, wnext);
                    bf16x8 bfr[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) bfr[p] = __builtin_bit_cast(bf16x8, wcur[p]);
                    rd(tap, 1, amid);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < NP; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[i], bfr[0], acc[i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (tap < 8) rd(tap + 1, 2, alo);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < NP; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[i], bfr[2], acc[i], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < NP; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(amid[i], bfr[1], acc[i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (tap < 8) rd(tap + 1, 0, ahin);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < NP; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(amid[i], bfr[0], acc[i], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < NP; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[i], bfr[1], acc[i], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < NP; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[i], bfr[0], acc[i], 0, 0, 0);
                    if (DB && tap >= 1 && tap <= NIT && stage_next) stage_store_item(tile_next, tap - 1);   // one item per tap: the split's VALU work rides between the MFMAs
                    if (tap < 8) {
#pragma unroll
                        for (int i = 0; i < NP; ++i) ahi[i] = ahin[i];
                    }
                    if (!(KNOCK & 1) && g + 1 < G) {
#pragma unroll
                        for (int p = 0; p < 3; ++p) wcur[p] = wnext[p];
                    }
                }
            };
            if (2 * nvalid > MT || MT == 1) taps(std::integral_constant<int, MT>{});
            else if (nvalid > 0) taps(std::integral_constant<int, (MT + 1) / 2>{});
            else if (stage_next) { stage_load(cc + 1); stage_store(tile_next); }    // a wave without pixels still stages its share
        } else {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3, g = cc * 9 + tap;
                if (tap == 0 && stage_next) stage_load(cc + 1);
                if (!(KNOCK & 1) && g + 1 < G) load_w(g + 1, wnext);
                bf16x8 bfr[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) bfr[p] = __builtin_bit_cast(bf16x8, wcur[p]);
                const unsigned char* tapbase = tile + lane_off + ky * RSB + kx * PSB;
#pragma unroll
                for (int g0 = 0; g0 < MT; g0 += GS) {
                    if (g0 >= nvalid) break;
                    bf16x8 afr[GS][3];
#pragma unroll
                    for (int i = 0; i < GS; ++i) {
                        const int pi = wm * MT + g0 + i, poff = 8 * (pi % PR) * RSB + 4 * (pi / PR) * PSB;
#pragma unroll
                        for (int p = 0; p < 3; ++p)
                            afr[i][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(tapbase + p * PLANE + poff));
                    }
                    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // small terms first
#pragma unroll
                    for (int t = 0; t < 6; ++t)
#pragma unroll
                        for (int i = 0; i < GS; ++i)
                            acc[g0 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[i][PA[t]], bfr[PB[t]], acc[g0 + i], 0, 0, 0);
                }
                if (DB && tap >= 1 && tap <= NIT && stage_next) stage_store_item(tile_next, tap - 1);
                if (!(KNOCK & 1) && g + 1 < G) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) wcur[p] = wnext[p];
                }
            }
        }
        if (DB && cc + 1 < NCH) __syncthreads();             // the other buffer is complete; this one may be overwritten
    }
    const int co = wn * 32 + m;
    if (OUT_MODE == 2) {
        // raw epilogue (training forward / data gradient) + per-workgroup channel statistics
        float ssum = 0.0f, ssq = 0.0f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int pi = wm * MT + mt, pr = pi % PR, pc = pi / PR;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int xl = (r & 1) + 2 * h, yl = ((r >> 1) & 1) + 2 * ((r >> 2) & 1) + 4 * ((r >> 3) & 1);
                const int gy = ty0 + 8 * pr + yl, gx = tx0 + 4 * pc + xl;
                if (gy < H && gx < W) {
                    const float v = acc[mt][r];
                    out[(((size_t)b * H + gy) * W + gx) * COUT + co] = v;
                    ssum += v;
                    ssq = fmaf(v, v, ssq);
                }
            }
        }
        if (stats) {
            float* lds = reinterpret_cast<float*>(ldsb);
            ssum += __shfl_xor(ssum, 32);
            ssq += __shfl_xor(ssq, 32);
            __syncthreads();
            if (h == 0) {
                lds[(wm * COUT + co) * 2] = ssum;
                lds[(wm * COUT + co) * 2 + 1] = ssq;
            }
            __syncthreads();
            const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            for (int c = tid; c < COUT; c += 256) {
                float s = 0.0f, q = 0.0f;
#pragma unroll
                for (int w4 = 0; w4 < WM; ++w4) { s += lds[(w4 * COUT + c) * 2]; q += lds[(w4 * COUT + c) * 2 + 1]; }
                stats[blk * COUT + c] = make_float2(s, q);
            }
        }
        return;
    }
    // epilogue: BN (folded) -> ReLU -> 2x2 max over registers 4q..4q+3 -> store
    const float s = scale[co], t = shift[co];
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
            for (int r = 0; r < 4; ++r) v = fmaxf(v, fmaf(acc[mt][4 * q + r], s, t));
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
                    // OUT_MODE 1 only: `stats` carries the f16x2 plane buffer [2][B * Wp][COUT * Hp] (f16_split.h) of the following
                    // GEMM's A operand, written here so that no separate split pass has to re-read the activations
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
