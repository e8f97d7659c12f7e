// Fused feature kernels: waveform batch -> normalised, zero-padded log-mel [B][n_mels][t_pad].
//
// Replaces (per clip, on CPU, one at a time in the reference):
//   torchaudio MelSpectrogram + AmplitudeToDB + z-norm   scripts/precompute_features.py:59-73
//   pad/trim to 200 frames                                scripts/dataset.py:109-113
//
// Kernel 1 (feat_frames): one workgroup = 16 consecutive frames of one utterance, 4 waves, each
//   wave transforms one frame at a time: 1024 real samples (reflect-padded, Hann-windowed on load)
//   are packed into 512 complex points, 8 per lane, and run through three radix-8 passes
//   (register butterflies, two exchanges through a private per-wave LDS slab -- no workgroup barrier
//   inside the frame loop), then untangled to the 513-bin power spectrum, reduced by the sparse
//   HTK mel filterbank (tap-major table staged in LDS), log-compressed and collected in an LDS tile
//   that is written out in 64-byte row segments together with (count, mean, M2) of the chunk.
//   Twiddles / window are per-lane constants and live in registers for the whole workgroup.
// Kernel 2 (feat_normalise): merges the chunk statistics (Chan), applies the whole-utterance
//   (x-mean)/(std_unbiased+1e-5), SpecAugment masks if given, and zero-fills the padding frames.
//
// HBM traffic per utterance (algorithmic): L*4 B read (L*2 for PCM16) + n_mels*t_pad*4 B written;
// the un-normalised dB tile makes one extra round trip that stays in L2/MALL (24 KB per 3 s clip).
#include <stdint.h>
#include "sir_internal.h"

namespace {

constexpr int FPC = SIR_FRAMES_PER_CHUNK;
constexpr int WAVES = 4;
constexpr int XS = 72;           // padded stride (complex) of the radix-8 exchange slabs: conflict-free
constexpr int XBUF = 8 * XS;     // complex slots per wave
constexpr float AMIN = 1e-10f;
constexpr float NORM_EPS = 1e-5f;

struct FeatTables {
    const float2* tw512;
    const float2* tw1024;
    const float* window;
    const float* melw;
    const int* mel_start;
    int max_taps;
    int n_mels;
};

struct AugArgs {
    const int32_t* shift;
    const float* sigma;
    unsigned long long seed;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)

// forward 8-point DFT, natural order in and out (decimation in frequency)
__device__ __forceinline__ void dft8(float2 (&v)[8]) {
    const float R = 0.70710678118654752440f;
    float2 a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
    float2 a1 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
    float2 a2 = cadd(v[2], v[6]), a6 = csub(v[2], v[6]);
    float2 a3 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
    a5 = make_float2((a5.x + a5.y) * R, (a5.y - a5.x) * R);      // * W8^1
    a6 = mul_mi(a6);                                             // * W8^2
    a7 = make_float2((a7.y - a7.x) * R, -(a7.x + a7.y) * R);     // * W8^3
    float2 b0 = cadd(a0, a2), b2 = csub(a0, a2), b1 = cadd(a1, a3), b3 = mul_mi(csub(a1, a3));
    float2 c0 = cadd(a4, a6), c2 = csub(a4, a6), c1 = cadd(a5, a7), c3 = mul_mi(csub(a5, a7));
    v[0] = cadd(b0, b1); v[4] = csub(b0, b1); v[2] = cadd(b2, b3); v[6] = csub(b2, b3);
    v[1] = cadd(c0, c1); v[5] = csub(c0, c1); v[3] = cadd(c2, c3); v[7] = csub(c2, c3);
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// standard normal, a pure function of (seed, utterance, sample index): every frame that touches a
// sample sees the same noise value
__device__ __forceinline__ float gauss_at(unsigned long long seed, int b, int i) {
    unsigned long long r = splitmix64(seed ^ splitmix64(((unsigned long long)(unsigned)b << 32) | (unsigned)i));
    float u1 = (float)((unsigned)(r >> 40) + 1u) * (1.0f / 16777216.0f);
    float u2 = (float)((unsigned)r & 0xFFFFFFu) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
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
        if (sigma > 0.0f) v += sigma * gauss_at(seed, b, i);
        return v;
    }
    return to_f32<T>(x[i]);
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

template <typename WT, bool AUG>
__global__ __launch_bounds__(256) void feat_frames_kernel(
    const WT* __restrict__ wave, long long wave_stride, const int32_t* __restrict__ lengths, int max_len,
    float* __restrict__ out, int t_pad, float4* __restrict__ stats, int nchunks, FeatTables tb, AugArgs aug) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* xall = reinterpret_cast<float2*>(smem);                    // [WAVES][XBUF]
    float* melw = reinterpret_cast<float*>(xall + WAVES * XBUF);       // [max_taps][64]
    float* tile = melw + tb.max_taps * 64;                             // [64][FPC+1]
    float* red = tile + 64 * (FPC + 1);                                // [16]

    const int b = blockIdx.y, chunk = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int L = lengths[b];
    if (L > max_len) L = max_len;
    const int T = (L > SIR_HOP) ? 1 + L / SIR_HOP : 0;
    const int t0 = chunk * FPC;
    if (t0 >= T) {                                  // block-uniform: nothing to do for this chunk
        if (tid == 0) stats[(size_t)b * nchunks + chunk] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    for (int i = tid; i < tb.max_taps * 64; i += 256) melw[i] = tb.melw[i];

    // per-lane constants, reused for every frame of this wave
    float2 tw1[8], tw2[8], twu[8];
    float win[16];
    const int m1p = lane & 7;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        tw1[k] = tb.tw512[(lane * k) & 511];            // W512^(n1*k2)
        tw2[k] = tb.tw512[(8 * m1p * k) & 511];         // W64^(m1*j2)
        twu[k] = tb.tw1024[lane + 64 * k];              // W1024^k, k = lane + 64*j1
        win[2 * k] = tb.window[2 * (lane + 64 * k)];
        win[2 * k + 1] = tb.window[2 * (lane + 64 * k) + 1];
    }
    const int mel_start = (lane < tb.n_mels) ? tb.mel_start[lane] : 0;
    const WT* x = wave + (size_t)b * wave_stride;
    int shift = 0;
    float sigma = 0.0f;
    if (AUG) {
        if (aug.shift) shift = aug.shift[b];
        if (aug.sigma) sigma = aug.sigma[b];
    }
    __syncthreads();

    float2* xb = xall + wv * XBUF;
    float* pb = reinterpret_cast<float*>(xb);           // power spectrum aliases the slab
    for (int fl = wv; fl < FPC; fl += WAVES) {
        const int t = t0 + fl;
        if (t >= T) {                                   // wave-uniform
            tile[lane * (FPC + 1) + fl] = 0.0f;
            continue;
        }
        float2 v[8];
        const int base = t * SIR_HOP - SIR_HOP;         // first padded sample of the frame, in clip coords
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i0 = base + 2 * (lane + 64 * j);
            // __fmul_rn: keep the window product un-fused so every template variant rounds alike
            v[j].x = __fmul_rn(fetch<WT, AUG>(x, L, i0, shift, sigma, aug.seed, b), win[2 * j]);
            v[j].y = __fmul_rn(fetch<WT, AUG>(x, L, i0 + 1, shift, sigma, aug.seed, b), win[2 * j + 1]);
        }
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
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = xb[k2 * XS + m1p + 8 * m];
            dft8(v);
#pragma unroll
            for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], tw2[k]);
            wave_fence();
#pragma unroll
            for (int j = 0; j < 8; ++j) xb[k2 * XS + j + 9 * m1p] = v[j];
            wave_fence();
        }
        {   // pass 3: lane = k2 + 8*j2: DFT over m1 -> Z[lane + 64*j1]
            const int k2 = lane & 7, j2 = lane >> 3;
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = xb[k2 * XS + j2 + 9 * m];
            dft8(v);
            wave_fence();
#pragma unroll
            for (int j = 0; j < 8; ++j) xb[lane + 64 * j] = v[j];
            wave_fence();
        }
        // untangle the packed real transform: X[k] = E[k] + W1024^k * O[k], power = |X|^2
        float p[8];
        float p512 = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = lane + 64 * j;
            const float2 z = v[j];
            const float2 zp = xb[(512 - k) & 511];
            const float2 e = make_float2(0.5f * (z.x + zp.x), 0.5f * (z.y - zp.y));
            const float2 d = make_float2(z.x - zp.x, z.y + zp.y);            // z - conj(zp)
            const float2 o = make_float2(0.5f * d.y, -0.5f * d.x);            // d / (2i)
            const float2 w = cmul(twu[j], o);
            const float xr = e.x + w.x, xi = e.y + w.y;
            p[j] = xr * xr + xi * xi;
            if (k == 0) { const float n = z.x - z.y; p512 = n * n; }          // X[512] = Re Z0 - Im Z0
        }
        wave_fence();
#pragma unroll
        for (int j = 0; j < 8; ++j) pb[lane + 64 * j] = p[j];
        if (lane == 0) pb[512] = p512;
        wave_fence();
        // sparse HTK mel filterbank: lane = filter, taps ascending in frequency
        float acc = 0.0f;
        for (int i = 0; i < tb.max_taps; ++i) {
            int k = mel_start + i;
            k = k > 512 ? 512 : k;
            acc = fmaf(melw[i * 64 + lane], pb[k], acc);
        }
        wave_fence();
        // the 1e-10 clamp is exact in the reference (silence -> exactly -100 dB)
        const float db = (acc <= AMIN) ? -100.0f : 10.0f * log10f(acc);
        tile[lane * (FPC + 1) + fl] = db;
    }
    __syncthreads();

    // chunk statistics over the valid (mel, frame) entries, two-pass inside the chunk
    const int nv = (T - t0) < FPC ? (T - t0) : FPC;
    const float cnt = (float)(nv * tb.n_mels);
    float s = 0.0f;
    for (int idx = tid; idx < 64 * FPC; idx += 256) {
        const int mel = idx / FPC, f = idx % FPC;
        if (mel < tb.n_mels && f < nv) s += tile[mel * (FPC + 1) + f];
    }
    s = wave_sum(s);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / cnt;
    float q = 0.0f;
    for (int idx = tid; idx < 64 * FPC; idx += 256) {
        const int mel = idx / FPC, f = idx % FPC;
        if (mel < tb.n_mels && f < nv) {
            const float v = tile[mel * (FPC + 1) + f];
            q += (v - mean) * (v - mean);
            if (t0 + f < t_pad) out[((size_t)b * tb.n_mels + mel) * t_pad + t0 + f] = v;
        }
    }
    q = wave_sum(q);
    if (lane == 0) red[8 + wv] = q;
    __syncthreads();
    if (tid == 0)
        stats[(size_t)b * nchunks + chunk] = make_float4(cnt, mean, red[8] + red[9] + red[10] + red[11], 0.f);
}

constexpr int NORM_ROWS = 16;

__global__ __launch_bounds__(256) void feat_normalise_kernel(
    float* __restrict__ out, float* __restrict__ db_out, int t_pad, int n_mels,
    const int32_t* __restrict__ lengths, int max_len, const float4* __restrict__ stats, int nchunks, const int32_t* __restrict__ time_mask,
    const int32_t* __restrict__ freq_mask) {
    const int b = blockIdx.y;
    int L = lengths[b];
    if (L > max_len) L = max_len;
    const int T = (L > SIR_HOP) ? 1 + L / SIR_HOP : 0;
    const int tv = T < t_pad ? T : t_pad;
    // merge chunk statistics (Chan et al.) ONCE per workgroup: the double-precision divisions of the merge, repeated by
    // every thread, were most of this kernel (a dozen elements per thread of actual work)
    __shared__ float s_mean, s_denom;
    if (threadIdx.x == 0) {
        double n = 0.0, mean = 0.0, m2 = 0.0;
        for (int c = 0; c < nchunks; ++c) {
            const float4 st = stats[(size_t)b * nchunks + c];
            if (st.x > 0.0f) {
                const double nc = st.x, delta = (double)st.y - mean, tot = n + nc;
                mean += delta * nc / tot;
                m2 += (double)st.z + delta * delta * n * nc / tot;
                n = tot;
            }
        }
        s_mean = (float)mean;
        s_denom = (n > 1.0 ? (float)sqrt(m2 / (n - 1.0)) : 0.0f) + NORM_EPS;
    }
    __syncthreads();
    const float meanf = s_mean, denom = s_denom;
    int tm0 = 0, tmw = 0, fm0 = 0, fmw = 0;
    if (time_mask) { tm0 = time_mask[2 * b]; tmw = time_mask[2 * b + 1]; }
    if (freq_mask) { fm0 = freq_mask[2 * b]; fmw = freq_mask[2 * b + 1]; }
    const int row0 = blockIdx.x * NORM_ROWS;
    float* o = out + ((size_t)b * n_mels + row0) * t_pad;
    float* dbo = db_out ? db_out + ((size_t)b * n_mels + row0) * t_pad : nullptr;
    const int rows = (n_mels - row0) < NORM_ROWS ? (n_mels - row0) : NORM_ROWS;
    if ((t_pad & 3) == 0 && ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(db_out)) & 15) == 0) {
        // four frames per thread and iteration (rows are 16-byte aligned)
        const int q4 = t_pad >> 2;
        for (int i4 = threadIdx.x; i4 < rows * q4; i4 += 256) {
            const int r = i4 / q4, t0 = 4 * (i4 - r * q4), mel = row0 + r;
            const bool fmasked = mel >= fm0 && mel < fm0 + fmw;
            float4* op = reinterpret_cast<float4*>(o + (size_t)r * t_pad + t0);
            float4 dbv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t0 < tv) dbv = *op;
            float dbs[4] = {dbv.x, dbv.y, dbv.z, dbv.w}, vs[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = t0 + e;
                float v = 0.0f;
                if (t < tv) {
                    v = (dbs[e] - meanf) / denom;
                    if ((t >= tm0 && t < tm0 + tmw) || fmasked) v = 0.0f;
                } else dbs[e] = 0.0f;
                vs[e] = v;
            }
            *op = make_float4(vs[0], vs[1], vs[2], vs[3]);
            if (dbo) *reinterpret_cast<float4*>(dbo + (size_t)r * t_pad + t0) = make_float4(dbs[0], dbs[1], dbs[2], dbs[3]);
        }
        return;
    }
    for (int idx = threadIdx.x; idx < rows * t_pad; idx += 256) {
        const int r = idx / t_pad, t = idx - r * t_pad;
        float v = 0.0f, db = 0.0f;
        if (t < tv) {
            db = o[idx];
            v = (db - meanf) / denom;
            const int mel = row0 + r;
            if ((t >= tm0 && t < tm0 + tmw) || (mel >= fm0 && mel < fm0 + fmw)) v = 0.0f;
        }
        o[idx] = v;
        if (dbo) dbo[idx] = db;
    }
}

}  // namespace

static inline int feat_nchunks(int max_len) {
    const int max_t = 1 + max_len / SIR_HOP;
    return (max_t + FPC - 1) / FPC;
}

extern "C" size_t sir_features_workspace_bytes(const sir_handle* h, int batch, int max_len) {
    (void)h;
    if (batch <= 0 || max_len <= 0) return 0;
    return sir_align_up((size_t)batch * feat_nchunks(max_len) * sizeof(float4), 256);
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
    if (batch > 65535) { sir_set_error("sir_features_fwd: batch > 65535"); return SIR_EINVAL; }
    const int nchunks = feat_nchunks(max_len);
    FeatTables tb{h->tw512, h->tw1024, h->window, h->melw, h->mel_start, h->max_taps, h->cfg.n_mels};
    AugArgs ag{nullptr, nullptr, 0ull};
    bool wave_aug = false;
    const int32_t *tmask = nullptr, *fmask = nullptr;
    if (aug) {
        ag.shift = aug->shift; ag.sigma = aug->noise_sigma; ag.seed = aug->noise_seed;
        wave_aug = aug->shift || aug->noise_sigma;
        tmask = aug->time_mask; fmask = aug->freq_mask;
    }
    float4* stats = reinterpret_cast<float4*>(workspace);
    const size_t lds = (size_t)WAVES * XBUF * sizeof(float2) + (size_t)h->max_taps * 64 * sizeof(float) +
                       64 * (FPC + 1) * sizeof(float) + 16 * sizeof(float);
    dim3 grid(nchunks, batch), block(256);
#define SIR_LAUNCH_FRAMES(TY, AUGF)                                                                          \
    hipLaunchKernelGGL((feat_frames_kernel<TY, AUGF>), grid, block, lds, stream, (const TY*)wave,           \
                       (long long)wave_stride, lengths, max_len, out, t_pad, stats, nchunks, tb, ag)
    {
    SirProfScope prof(h, SIR_K_FEAT_FRAMES, stream);
    if (wave_dtype == SIR_WAVE_F32) { if (wave_aug) SIR_LAUNCH_FRAMES(float, true); else SIR_LAUNCH_FRAMES(float, false); }
    else { if (wave_aug) SIR_LAUNCH_FRAMES(short, true); else SIR_LAUNCH_FRAMES(short, false); }
    }
#undef SIR_LAUNCH_FRAMES
    SIR_HIP_TRY(hipGetLastError());
    SirProfScope prof2(h, SIR_K_FEAT_NORM, stream);
    dim3 grid2((h->cfg.n_mels + NORM_ROWS - 1) / NORM_ROWS, batch);
    hipLaunchKernelGGL(feat_normalise_kernel, grid2, block, 0, stream, out, db_out, t_pad, h->cfg.n_mels, lengths, max_len,
                       (const float4*)stats, nchunks, tmask, fmask);
    SIR_HIP_TRY(hipGetLastError());
    return SIR_OK;
}
