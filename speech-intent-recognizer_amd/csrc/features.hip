// Fused feature kernel: waveform batch -> normalised, zero-padded log-mel [B][n_mels][t_pad], ONE launch.
//
// Replaces (per clip, on CPU, one at a time in the reference):
//   torchaudio MelSpectrogram + AmplitudeToDB + z-norm   scripts/precompute_features.py:59-73
//   pad/trim to 200 frames                                scripts/dataset.py:109-113
//   SpecAugment bands / time_shift / add_noise            scripts/dataset.py:160-176, scripts/augment.py:6-28, :82-96
//
// feat_utt_kernel: one workgroup of 16 waves owns ONE utterance and walks its frames 16 at a time ("rounds"):
//   * FFT: wave w transforms frame 16 r + w.  1024 real samples (reflect-padded, Hann-windowed on load, the next
//     round's samples already in flight) are packed into 512 complex points, 8 per lane, and run through three
//     radix-8 passes (register butterflies, two exchanges through the wave's private LDS slab -- no workgroup
//     barrier inside the transform), then untangled to the 513-bin power spectrum, written to row w of P.
//   * mel: after one barrier all 16 spectra of the round are in LDS and the 1024 threads split the 16 x 64
//     (frame, filter) dot products one each: a thread walks only ITS filter's taps (2 ... 41 of them; the
//     filterbank is stored compact), filters are sorted by length so the four filters a wave handles are equally
//     long, and lanes that read P differ in the frame (row stride 513 words: a different bank each).  This is
//     1000 multiply-adds per frame instead of the 64 x 41 padded ones of a lane-per-filter loop.
//   * log-compress: the thread keeps its dB value of every round IN A REGISTER (10 rounds = 160 frames; 5 s clips are
//     157 frames), so the utterance's whole [64 x T] dB tile lives in the register file.  After the last round the
//     workgroup reduces mean and unbiased variance over it (two passes, double-precision combine) and every thread
//     stores (x - mean) / (std + 1e-5) of its own values with the SpecAugment bands and the zero padding (64-byte row
//     segments): no second kernel, no statistics round trip through L2, no tile in LDS -- which leaves room to
//     double-buffer P, so a round costs ONE workgroup barrier.
//     Clips longer than 160 frames (only the single-file predict surface feeds those) park their dB values in the
//     output rows instead and the same workgroup re-reads them for the two passes (same CU, same L1: no
//     inter-workgroup visibility involved).
// LDS traffic is what bounds the transform (PMC: the LDS pipe was busy 45 % of the first fused version, 29 % of that
// bank conflicts), hence: every exchange read is a single ds_read_b64 (`volatile`: hipcc otherwise pairs them into
// ds_read2_b64, which moves HALF the bytes per LDS cycle and banks mod 32 in 16-lane groups); the second exchange uses a
// stride-68 XOR-swizzled layout that is conflict-free for its 16-lane ds_write_b64 groups AND its 32-lane ds_read_b64
// groups; the untangle partner Z[512-k] comes from lane (64 - lane) by ds_bpermute instead of a store + load.
// HBM traffic per utterance (algorithmic): L*4 B read (L*2 for PCM16) + n_mels*t_pad*4 B written.
#include <stdint.h>
#include "sir_internal.h"

namespace {

constexpr int NW = 16;           // waves per workgroup = frames per round
constexpr int THREADS = NW * 64;
constexpr int XS = 72;           // first exchange: row stride (complex) -- conflict-free 64-bank ds_read_b64 in pass 2
constexpr int XS2 = 68;          // second exchange: row stride (complex), with the XOR swizzle of ex2_index
constexpr int XBUF = 8 * XS;     // complex slots per wave
constexpr int PROW = 514;        // words per power-spectrum row: the 16 frames of a filter sit on 16 different banks
constexpr int TW2S = 10;         // row stride (complex) of the pass-2 twiddle table in LDS
constexpr int RMAX = 10;         // rounds whose dB values a thread keeps in registers
constexpr int TILE_T = RMAX * NW;   // = 160 frames
constexpr float AMIN = 1e-10f;
constexpr float NORM_EPS = 1e-5f;

struct FeatTables {
    const float2* tw512;
    const float2* tw1024;
    const float* window;
    const float* melw;           // compact filter weights, filter after filter, taps ascending in frequency
    const int4* mel_desc;        // [64] per SLOT (filters sorted by tap count): {filter, first bin, taps, offset into melw}
    int mel_nnz;
    int n_mels;
};

struct AugArgs {
    const int32_t* shift;
    const float* sigma;
    unsigned long long seed;
};

// Complex values are 2-vectors so that every complex add / sub / scale is ONE packed instruction (v_pk_add_f32,
// v_pk_mul_f32, v_pk_fma_f32 with op_sel / neg modifiers for the swaps and sign flips): a VALU instruction costs the same
// ~4 issue cycles whether it carries one float or two per lane, and this kernel is issue-bound (PMC: SQ_ACTIVE_INST_ANY
// = the kernel's duration at 4.3 cycles per instruction).
typedef float cf32 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf32 CF(float2 a) { return cf32{a.x, a.y}; }
__device__ __forceinline__ cf32 swp(cf32 a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ cf32 mul_mi(cf32 a) { return cf32{a.y, -a.x}; }                       // a * (-i)
__device__ __forceinline__ cf32 cmul(cf32 a, cf32 b) {                                           // a * b
    return cf32{a.x, a.x} * b + cf32{a.y, a.y} * cf32{-b.y, b.x};
}

// slot of element (row k2, j, m1) in the second exchange: (j, m1) -> (j ^ (m1 >> 1)) + 8 m1 is injective, the 16 lanes
// (k2 in {2g, 2g+1}, m1) of a ds_write_b64 group land on 16 different bank pairs mod 32, and the 32 lanes (k2, j2 in
// {4g .. 4g+3}) of a ds_read_b64 group on 32 different bank pairs mod 64 (row stride 68: 136 words = 8 mod 64)
__device__ __forceinline__ int ex2_index(int k2, int j, int m1) { return k2 * XS2 + ((j ^ (m1 >> 1)) + 8 * m1); }

// Eight single ds_read_b64 + their wait in ONE asm statement (cdna_hip_programming.md 5.7 form (i)): hipcc pairs plain --
// and volatile -- adjacent LDS loads into ds_read2_b64 / ds_read2st64_b64, which move half the bytes per LDS cycle.
typedef cf32 f32x2;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)p; }   // LDS byte offset of a __shared__ pointer

// v[m] = *(base + m * STRIDE_BYTES), m = 0..7
template <int STRIDE_BYTES>
__device__ __forceinline__ void lds_read8(unsigned base, cf32 (&v)[8]) {
    f32x2 r0, r1, r2, r3, r4, r5, r6, r7;
    asm volatile(
        "ds_read_b64 %0, %8 offset:%9\n\tds_read_b64 %1, %8 offset:%10\n\tds_read_b64 %2, %8 offset:%11\n\t"
        "ds_read_b64 %3, %8 offset:%12\n\tds_read_b64 %4, %8 offset:%13\n\tds_read_b64 %5, %8 offset:%14\n\t"
        "ds_read_b64 %6, %8 offset:%15\n\tds_read_b64 %7, %8 offset:%16\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
        : "v"(base), "i"(0 * STRIDE_BYTES), "i"(1 * STRIDE_BYTES), "i"(2 * STRIDE_BYTES), "i"(3 * STRIDE_BYTES),
          "i"(4 * STRIDE_BYTES), "i"(5 * STRIDE_BYTES), "i"(6 * STRIDE_BYTES), "i"(7 * STRIDE_BYTES)
        : "memory");
    v[0] = r0; v[1] = r1; v[2] = r2; v[3] = r3; v[4] = r4; v[5] = r5; v[6] = r6; v[7] = r7;
}
// v[m] = *(base[m >> 1] + m * STRIDE_BYTES): the XOR-swizzled second exchange (one base per value of m >> 1)
template <int STRIDE_BYTES>
__device__ __forceinline__ void lds_read8x4(unsigned b0, unsigned b1, unsigned b2, unsigned b3, cf32 (&v)[8]) {
    f32x2 r0, r1, r2, r3, r4, r5, r6, r7;
    asm volatile(
        "ds_read_b64 %0, %8 offset:%12\n\tds_read_b64 %1, %8 offset:%13\n\tds_read_b64 %2, %9 offset:%14\n\t"
        "ds_read_b64 %3, %9 offset:%15\n\tds_read_b64 %4, %10 offset:%16\n\tds_read_b64 %5, %10 offset:%17\n\t"
        "ds_read_b64 %6, %11 offset:%18\n\tds_read_b64 %7, %11 offset:%19\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
        : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "i"(0 * STRIDE_BYTES), "i"(1 * STRIDE_BYTES), "i"(2 * STRIDE_BYTES),
          "i"(3 * STRIDE_BYTES), "i"(4 * STRIDE_BYTES), "i"(5 * STRIDE_BYTES), "i"(6 * STRIDE_BYTES), "i"(7 * STRIDE_BYTES)
        : "memory");
    v[0] = r0; v[1] = r1; v[2] = r2; v[3] = r3; v[4] = r4; v[5] = r5; v[6] = r6; v[7] = r7;
}

// forward 8-point DFT, natural order in and out (decimation in frequency)
__device__ __forceinline__ void dft8(cf32 (&v)[8]) {
    const float R = 0.70710678118654752440f;
    cf32 a0 = v[0] + v[4], a4 = v[0] - v[4];
    cf32 a1 = v[1] + v[5], a5 = v[1] - v[5];
    cf32 a2 = v[2] + v[6], a6 = v[2] - v[6];
    cf32 a3 = v[3] + v[7], a7 = v[3] - v[7];
    a5 = (a5 + mul_mi(a5)) * R;                                  // * W8^1 = (x + y, y - x) / sqrt 2
    a6 = mul_mi(a6);                                             // * W8^2
    a7 = (mul_mi(a7) - a7) * R;                                  // * W8^3 = (y - x, -x - y) / sqrt 2
    cf32 b0 = a0 + a2, b2 = a0 - a2, b1 = a1 + a3, b3 = mul_mi(a1 - a3);
    cf32 c0 = a4 + a6, c2 = a4 - a6, c1 = a5 + a7, c3 = mul_mi(a5 - a7);
    v[0] = b0 + b1; v[4] = b0 - b1; v[2] = b2 + b3; v[6] = b2 - b3;
    v[1] = c0 + c1; v[5] = c0 - c1; v[3] = c2 + c3; v[7] = c2 - c3;
}

__device__ __forceinline__ unsigned fmix32(unsigned x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
// Two independent standard normals, a pure function of (seed, utterance, sample PAIR index p): samples 2p and 2p+1 of the
// clip take .x and .y, so every frame that touches a sample sees the same noise value (Box-Muller on two 24-bit
// uniforms from a counter hash; hardware log2 / sqrt / sin / cos: ~1e-6 absolute, far below the sigma <= 1e-2 it scales).
// Restated on the host in tests/host_rng.py.
__device__ __forceinline__ float2 gauss_pair(unsigned long long seed, int b, int p) {
    const unsigned k = fmix32((unsigned)seed ^ ((unsigned)p * 0x9E3779B1u) ^ ((unsigned)b * 0x85EBCA77u));
    const unsigned a = fmix32(k ^ (unsigned)(seed >> 32));
    const unsigned c = fmix32(a + 0x632BE5ABu + (unsigned)p);
    const float u1 = (float)((a >> 8) + 1u) * (1.0f / 16777216.0f);      // (0, 1]
    const float u2 = (float)(c >> 8) * (1.0f / 16777216.0f);             // [0, 1)  (revolutions)
    const float r = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1), logf = log2
    return make_float2(r * __builtin_amdgcn_cosf(u2), r * __builtin_amdgcn_sinf(u2));
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<short>(short v) { return (float)v * (1.0f / 32768.0f); }

// sample i of the (augmented) clip after reflect padding; needs L > 512
template <typename T, bool AUG>
__device__ __forceinline__ float fetch(const T* __restrict__ x, int L, int i, int shift, float sigma,
                                       unsigned long long seed, int b) {
    if (i < 0) i = -i;
    else if (i >= L) i = 2 * L - 2 - i;
    if (AUG) {
        int s = i - shift;
        float v = (s >= 0 && s < L) ? to_f32<T>(x[s]) : 0.0f;
        if (sigma > 0.0f) {
            const float2 g = gauss_pair(seed, b, i >> 1);
            v += sigma * ((i & 1) ? g.y : g.x);
        }
        return v;
    }
    return to_f32<T>(x[i]);
}

// the 16 samples of one lane for frame t (pairs (i0, i0 + 1), i0 = base + 2 (lane + 64 j)), un-windowed
template <typename T, bool AUG>
__device__ __forceinline__ void load_frame(const T* __restrict__ x, int L, int t, int lane, int shift, float sigma,
                                           unsigned long long seed, int b, cf32 (&s)[8]) {
    const int base = t * SIR_HOP - SIR_HOP;         // first padded sample of the frame, in clip coordinates
    const bool interior = base >= 0 && base + SIR_NFFT <= L;          // wave-uniform: no reflection in this frame
    const bool paired = (reinterpret_cast<uintptr_t>(x) & (2 * sizeof(T) - 1)) == 0;   // row starts on a sample-pair boundary
    if (interior && !AUG && !paired) {              // odd row stride / offset view: two scalar loads per pair
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const T* q = x + base + 2 * (lane + 64 * j);
            s[j] = cf32{to_f32<T>(q[0]), to_f32<T>(q[1])};
        }
        return;
    }
    if (interior && !AUG) {
        if (sizeof(T) == 4) {
            const cf32* p = reinterpret_cast<const cf32*>(x + base) + lane;           // base is even: 8-byte aligned rows
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = p[64 * j];
        } else {
            const unsigned* p = reinterpret_cast<const unsigned*>(x + base) + lane;   // two PCM16 samples per word
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned w = p[64 * j];
                s[j] = cf32{(float)(short)(w & 0xFFFFu), (float)(short)(w >> 16)} * (1.0f / 32768.0f);
            }
        }
        return;
    }
    if (interior && AUG) {                          // shifted reads + ONE noise pair per lane and j (i0 is even)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i0 = base + 2 * (lane + 64 * j), s0 = i0 - shift;
            cf32 v;
            v.x = (s0 >= 0 && s0 < L) ? to_f32<T>(x[s0]) : 0.0f;
            v.y = (s0 + 1 >= 0 && s0 + 1 < L) ? to_f32<T>(x[s0 + 1]) : 0.0f;
            if (sigma > 0.0f) {
                const float2 g = gauss_pair(seed, b, i0 >> 1);
                v += sigma * cf32{g.x, g.y};
            }
            s[j] = v;
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {                   // first / last frames: reflect padding, element by element
        const int i0 = base + 2 * (lane + 64 * j);
        s[j].x = fetch<T, AUG>(x, L, i0, shift, sigma, seed, b);
        s[j].y = fetch<T, AUG>(x, L, i0 + 1, shift, sigma, seed, b);
    }
}

__device__ __forceinline__ void wave_fence() {
    // per-wave LDS slab: LDS ops of one wave execute in order, only the compiler must not reorder
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// workgroup sum of one float per thread, combined in double (deterministic order); every thread gets the result
__device__ __forceinline__ double block_sum(float v, double* red, int lane, int wv) {
    v = wave_sum(v);
    __syncthreads();                                 // red is reused across calls
    if (lane == 0) red[wv] = (double)v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += red[i];
    return s;
}

template <typename WT, bool AUG>
__global__ __launch_bounds__(THREADS) void feat_utt_kernel(
    const WT* __restrict__ wave, long long wave_stride, const int32_t* __restrict__ lengths, int max_len,
    float* __restrict__ out, float* __restrict__ db_out, int t_pad, FeatTables tb, AugArgs aug,
    const int32_t* __restrict__ time_mask, const int32_t* __restrict__ freq_mask) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf32* xall = reinterpret_cast<cf32*>(smem);                        // [NW][XBUF] exchange slabs
    double* red = reinterpret_cast<double*>(xall + NW * XBUF);         // [NW] reduction scratch
    float* Pbuf = reinterpret_cast<float*>(red + NW);                  // [2][NW][PROW] power spectra, by round parity
    float* melw = Pbuf + 2 * NW * PROW;                                // [mel_nnz]
    cf32* winl = reinterpret_cast<cf32*>(melw + ((tb.mel_nnz + 3) & ~3));       // [512] Hann window as sample pairs
    cf32* twul = winl + 512;                                           // [512] -i/2 * W1024^k, the untangle twiddles
    cf32* tw2l = twul + 512;                                           // [8][TW2S] W64^(m1 * j2), row m1 (stride 10: the 8 rows' 16-byte reads fall on disjoint banks)

    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n_mels = tb.n_mels;
    int L = lengths[b];
    if (L > max_len) L = max_len;
    const int T = (L > SIR_HOP) ? 1 + L / SIR_HOP : 0;                 // L <= 512: the reference fails (reflect pad) -> zero row
    const int tv = T < t_pad ? T : t_pad;
    float* orow = out + (size_t)b * n_mels * t_pad;
    float* drow = db_out ? db_out + (size_t)b * n_mels * t_pad : nullptr;
    const bool in_regs = T <= TILE_T;                                  // block-uniform
    const int nrounds = (T + NW - 1) / NW;
    // mel role of this thread: (frame of the round, filter slot); the four slots of a wave are equally long filters
    const int mf = tid & 15;
    const int4 md = tb.mel_desc[tid >> 4];                             // {filter, first bin, taps, offset}
    float dbv[RMAX];                                                   // this thread's dB values: frame 16 r + mf of filter md.x
#pragma unroll
    for (int r = 0; r < RMAX; ++r) dbv[r] = 0.0f;

    if (nrounds > 0) {
        for (int i = tid; i < tb.mel_nnz; i += THREADS) melw[i] = tb.melw[i];
        if (tid < 512) {
            winl[tid] = reinterpret_cast<const cf32*>(tb.window)[tid];
            // X[k] = (z + conj zp)/2 + (-i/2 W1024^k)(z - conj zp)
            twul[tid] = 0.5f * mul_mi(CF(tb.tw1024[tid]));
        }
        if (tid < 64) tw2l[(tid >> 3) * TW2S + (tid & 7)] = CF(tb.tw512[(8 * (tid >> 3) * (tid & 7)) & 511]);

        // per-lane constants of the transform, reused for every frame of this wave (the window, the pass-2 twiddles -- which only
        // depend on lane & 7 -- and the untangle twiddles live in LDS: 46 more registers per lane would spill at four waves per SIMD)
        cf32 tw1[8];
        const int m1p = lane & 7;
#pragma unroll
        for (int k = 0; k < 8; ++k) tw1[k] = CF(tb.tw512[(lane * k) & 511]);        // W512^(n1*k2)
        const WT* x = wave + (size_t)b * wave_stride;
        int shift = 0;
        float sigma = 0.0f;
        if (AUG) {
            if (aug.shift) shift = aug.shift[b];
            if (aug.sigma) sigma = aug.sigma[b];
        }
        cf32* xb = xall + wv * XBUF;
        const int mirror = ((64 - lane) & 63) * 4;          // ds_bpermute address of the lane that holds Z[512 - k]
        // LDS byte addresses of this lane's read columns (loop-invariant)
        const unsigned a_win = lds_addr(winl + lane);
        const unsigned a_p2 = lds_addr(xb + (lane >> 3) * XS + m1p);
        unsigned a_p3[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a_p3[q] = lds_addr(xb + (lane & 7) * XS2 + ((lane >> 3) ^ q));

        cf32 nxt[8];
        if (wv < T) load_frame<WT, AUG>(x, L, wv, lane, shift, sigma, aug.seed, b, nxt);
        __syncthreads();                                    // mel weights and window are staged

        for (int r = 0; r < nrounds; ++r) {
            const int t = r * NW + wv;
            float* P = Pbuf + (r & 1) * (NW * PROW);
            if (t < T) {                                    // wave-uniform
                cf32 v[8];
                lds_read8<64 * 8>(a_win, v);                // window pairs lane + 64 j
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = nxt[j] * v[j];       // (a product only: nothing to fuse)
                if (t + NW < T) load_frame<WT, AUG>(x, L, t + NW, lane, shift, sigma, aug.seed, b, nxt);   // next round's samples
                // pass 1: DFT over n2 (stride 64), twiddle W512^(n1*k2)
                dft8(v);
#pragma unroll
                for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], tw1[k]);
                wave_fence();
#pragma unroll
                for (int k = 0; k < 8; ++k) xb[k * XS + lane] = v[k];
                wave_fence();
                {   // pass 2: lane = (k2, m1): DFT over m2, twiddle W64^(m1*j2)
                    const int k2 = lane >> 3;
                    lds_read8<8 * 8>(a_p2, v);              // xb[k2 * XS + m1p + 8 m]
                    dft8(v);
#pragma unroll
                    for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], tw2l[TW2S * m1p + k]);
                    wave_fence();
#pragma unroll
                    for (int j = 0; j < 8; ++j) xb[ex2_index(k2, j, m1p)] = v[j];
                    wave_fence();
                }
                {   // pass 3: lane = k2 + 8*j2: DFT over m1 -> Z[lane + 64*j1]
                    lds_read8x4<8 * 8>(a_p3[0], a_p3[1], a_p3[2], a_p3[3], v);   // xb[ex2_index(lane & 7, lane >> 3, m)]
                    dft8(v);
                }
                // untangle the packed real transform: X[k] = (z + conj zp)/2 + (-i/2 W1024^k)(z - conj zp), power = |X|^2,
                // k = lane + 64 j.  The partner zp = Z[512 - k] is register 7 - j of lane 64 - lane (lane 0: its own
                // register 8 - j, and Z[512] = Z[0]).
                float* prow = P + wv * PROW;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const cf32 z = v[j];
                    cf32 zc;
                    zc.x = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(v[7 - j].x)));
                    zc.y = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(v[7 - j].y)));
                    if (lane == 0) zc = v[(8 - j) & 7];
                    zc.y = -zc.y;                           // conj(zp)
                    const cf32 xk = 0.5f * (z + zc) + cmul(z - zc, twul[lane + 64 * j]);
                    const cf32 sq = xk * xk;
                    prow[lane + 64 * j] = sq.x + sq.y;
                }
                if (lane == 0) { const float n = v[0].x - v[0].y; prow[512] = n * n; }       // X[512] = Re Z0 - Im Z0
            }
            __syncthreads();                                // all spectra of this round are in P (the other parity is free again)
            // sparse HTK mel filterbank: thread = (frame mf, filter md.x), taps ascending in frequency
            const int tm = r * NW + mf;
            if (md.x >= 0 && tm < T) {
                const float* pr = P + mf * PROW + md.y;
                const float* wr = melw + md.w;
                float acc = 0.0f;
                for (int i = 0; i < md.z; ++i) acc = fmaf(wr[i], pr[i], acc);
                // the 1e-10 clamp is exact in the reference (silence -> exactly -100 dB)
                const float db = (acc <= AMIN) ? -100.0f : 10.0f * log10f(acc);
                if (in_regs) {
#pragma unroll
                    for (int q = 0; q < RMAX; ++q) dbv[q] = (q == r) ? db : dbv[q];
                } else if (tm < t_pad) {
                    orow[(size_t)md.x * t_pad + tm] = db;   // long clip: dB parked in the output row
                }
            }
        }
        __syncthreads();
    }

    // ---- whole-utterance statistics (two passes) and the normalised, masked, zero-padded store ----------------------
    // the statistics are over ALL frames of the clip (precompute_features.py:73 normalises before any trim); a clip
    // longer than 160 frames has every frame parked in its output row (the host checks t_pad >= T for those)
    float mean = 0.0f, denom = 1.0f;
    const bool mine = md.x >= 0;
    if (T > 0) {
        float s = 0.0f;
        if (in_regs) {
            if (mine) {
#pragma unroll
                for (int q = 0; q < RMAX; ++q) if (q * NW + mf < T) s += dbv[q];
            }
        } else {
            for (int idx = tid; idx < n_mels * tv; idx += THREADS) s += orow[(size_t)(idx / tv) * t_pad + idx % tv];
        }
        const double cnt = (double)n_mels * (in_regs ? T : tv);
        mean = (float)(block_sum(s, red, lane, wv) / cnt);
        float q2 = 0.0f;
        if (in_regs) {
            if (mine) {
#pragma unroll
                for (int q = 0; q < RMAX; ++q) if (q * NW + mf < T) { const float d = dbv[q] - mean; q2 += d * d; }
            }
        } else {
            for (int idx = tid; idx < n_mels * tv; idx += THREADS) { const float d = orow[(size_t)(idx / tv) * t_pad + idx % tv] - mean; q2 += d * d; }
        }
        const double m2 = block_sum(q2, red, lane, wv);
        denom = (cnt > 1.0 ? (float)sqrt(m2 / (cnt - 1.0)) : 0.0f) + NORM_EPS;
    }
    int tm0 = 0, tmw = 0, fm0 = 0, fmw = 0;
    if (time_mask) { tm0 = time_mask[2 * b]; tmw = time_mask[2 * b + 1]; }
    if (freq_mask) { fm0 = freq_mask[2 * b]; fmw = freq_mask[2 * b + 1]; }
    if (in_regs) {
        // every thread stores its own values: 16 lanes = 16 consecutive frames of one mel row (64-byte segments)
        if (mine) {
            const bool fmasked = md.x >= fm0 && md.x < fm0 + fmw;
            float* o = orow + (size_t)md.x * t_pad;
            float* dbo = drow ? drow + (size_t)md.x * t_pad : nullptr;
#pragma unroll
            for (int q = 0; q < RMAX; ++q) {
                const int t = q * NW + mf;
                if (t < t_pad) {
                    float v = 0.0f, d = 0.0f;
                    if (t < tv) {
                        d = dbv[q];
                        v = (d - mean) / denom;
                        if ((t >= tm0 && t < tm0 + tmw) || fmasked) v = 0.0f;
                    }
                    o[t] = v;
                    if (dbo) dbo[t] = d;
                }
            }
            for (int t = RMAX * NW + mf; t < t_pad; t += NW) {       // padding beyond the register tile
                o[t] = 0.0f;
                if (dbo) dbo[t] = 0.0f;
            }
        }
    } else {
        for (int idx = tid; idx < n_mels * t_pad; idx += THREADS) {
            const int mel = idx / t_pad, t = idx - mel * t_pad;
            float v = 0.0f, d = 0.0f;
            if (t < tv) {
                d = orow[idx];
                v = (d - mean) / denom;
                if ((t >= tm0 && t < tm0 + tmw) || (mel >= fm0 && mel < fm0 + fmw)) v = 0.0f;
            }
            orow[idx] = v;
            if (drow) drow[idx] = d;
        }
    }
}

}  // namespace

extern "C" size_t sir_features_workspace_bytes(const sir_handle* h, int batch, int max_len) {
    (void)h;
    if (batch <= 0 || max_len <= 0) return 0;
    return 256;          // the fused kernel keeps its statistics on chip; a token size keeps the (workspace, bytes) contract
}

int sir_features_launch(sir_handle* h, const void* wave, int wave_dtype, int64_t wave_stride,
                        const int32_t* lengths, int batch, int max_len, float* out, int t_pad,
                        float* db_out, void* workspace, size_t workspace_bytes, const sir_augment* aug,
                        hipStream_t stream) {
    if (!h || !wave || !lengths || !out || !workspace) { sir_set_error("sir_features_fwd: NULL argument"); return SIR_EINVAL; }
    if (batch <= 0 || max_len <= 0 || t_pad <= 0 || wave_stride < max_len) {
        sir_set_error("sir_features_fwd: bad shape batch=%d max_len=%d t_pad=%d stride=%lld", batch, max_len, t_pad,
                      (long long)wave_stride);
        return SIR_EINVAL;
    }
    if (wave_dtype != SIR_WAVE_F32 && wave_dtype != SIR_WAVE_I16) { sir_set_error("sir_features_fwd: wave_dtype %d", wave_dtype); return SIR_EINVAL; }
    if (workspace_bytes < sir_features_workspace_bytes(h, batch, max_len)) { sir_set_error("sir_features_fwd: workspace too small"); return SIR_ENOMEM; }
    const size_t esz = wave_dtype == SIR_WAVE_F32 ? 4 : 2;
    if (((uintptr_t)wave % esz) != 0) { sir_set_error("sir_features_fwd: waveform pointer is not aligned to its sample type"); return SIR_EINVAL; }
    const int max_t = 1 + max_len / SIR_HOP;
    if (max_t > TILE_T && max_t > t_pad) {
        // a clip longer than the LDS tile parks its dB values in its output rows; the statistics of the reference are
        // over ALL frames (precompute_features.py:73 normalises before any trim), so every frame needs a slot there
        sir_set_error("sir_features_fwd: clips of up to %d frames need t_pad >= %d (or max_len <= %d samples)", max_t, max_t,
                      TILE_T * SIR_HOP + SIR_HOP - 1);
        return SIR_EUNSUPPORTED;
    }
    FeatTables tb{h->tw512, h->tw1024, h->window, h->melw, h->mel_desc, h->mel_nnz, h->cfg.n_mels};
    AugArgs ag{nullptr, nullptr, 0ull};
    bool wave_aug = false;
    const int32_t *tmask = nullptr, *fmask = nullptr;
    if (aug) {
        ag.shift = aug->shift; ag.sigma = aug->noise_sigma; ag.seed = aug->noise_seed;
        wave_aug = aug->shift || aug->noise_sigma;
        tmask = aug->time_mask; fmask = aug->freq_mask;
    }
    const size_t lds = (size_t)NW * XBUF * sizeof(float2) + NW * sizeof(double) + (size_t)2 * NW * PROW * sizeof(float) +
                       (size_t)((h->mel_nnz + 3) & ~3) * sizeof(float) + (1024 + 8 * TW2S) * sizeof(float2);
    if (!h->feat_attr_set) {          // > 64 KB of dynamic LDS needs the opt-in, once per handle (= per device)
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)feat_utt_kernel<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)feat_utt_kernel<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)feat_utt_kernel<short, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)feat_utt_kernel<short, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        h->feat_attr_set = true;
    }
    dim3 grid(batch), block(THREADS);
#define SIR_LAUNCH_FEAT(TY, AUGF)                                                                            \
    hipLaunchKernelGGL((feat_utt_kernel<TY, AUGF>), grid, block, lds, stream, (const TY*)wave,              \
                       (long long)wave_stride, lengths, max_len, out, db_out, t_pad, tb, ag, tmask, fmask)
    {
    SirProfScope prof(h, SIR_K_FEAT_FRAMES, stream);
    if (wave_dtype == SIR_WAVE_F32) { if (wave_aug) SIR_LAUNCH_FEAT(float, true); else SIR_LAUNCH_FEAT(float, false); }
    else { if (wave_aug) SIR_LAUNCH_FEAT(short, true); else SIR_LAUNCH_FEAT(short, false); }
    }
#undef SIR_LAUNCH_FEAT
    SIR_HIP_TRY(hipGetLastError());
    return SIR_OK;
}
