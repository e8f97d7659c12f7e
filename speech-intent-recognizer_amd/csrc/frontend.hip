// Waveform front-end ahead of the feature path: channel mix-down and sample-rate conversion on the GPU
// (reference: torch.mean(waveform, dim=0) at scripts/precompute_features.py:50-51 and
// torchaudio.transforms.Resample(sr, 16000) at :54-56, scripts/dataset.py:132-135, scripts/test_model.py:68-72).
//
// Resample = torchaudio's "sinc_interp_hann" polyphase FIR (lowpass_filter_width 6, rolloff 0.99).  With the
// rates reduced by their gcd (orig, new), output sample j = n * new + p is
//     y[j] = sum_k kernel[p][k] * x[n * orig + k - width],   k in [0, 2 * width + orig)
// where kernel[p][k] is nonzero only for the ~2 * 6 * orig / (0.99 * min(orig, new)) taps inside the Hann
// window.  The host builds the kernel in float64 exactly as torchaudio does (including its float32 phase
// term -p / new), casts to float32 and keeps, per phase, only the in-window taps: table [new][L] + first
// tap index.  The GPU kernel is a gather-FMA over L taps: HBM-bound (4 B in + 4 B out per sample), no LDS.
#include <math.h>
#include <algorithm>
#include <vector>
#include "sir_internal.h"

namespace {

constexpr int kLowpassWidth = 6;
constexpr double kRolloff = 0.99;

int gcd_int(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

struct HostTable { int orig, nw, width, L; std::vector<float> taps; std::vector<int> first; };

HostTable build_table(int orig_freq, int new_freq) {
    HostTable t;
    const int g = gcd_int(orig_freq, new_freq);
    t.orig = orig_freq / g;
    t.nw = new_freq / g;
    const double base = std::min(t.orig, t.nw) * kRolloff;
    t.width = (int)ceil(kLowpassWidth * (double)t.orig / base);
    const int K = 2 * t.width + t.orig;
    const double scale = base / t.orig;
    std::vector<double> row(K);
    std::vector<int> lo(t.nw), hi(t.nw);
    std::vector<std::vector<float>> rows(t.nw);
    t.L = 1;
    for (int p = 0; p < t.nw; ++p) {
        const float phase32 = (float)(-p) / (float)t.nw;             // int64 tensor / int -> float32 in torch
        int first = K, last = -1;
        rows[p].assign(K, 0.0f);
        for (int k = 0; k < K; ++k) {
            double tt = ((double)phase32 + (double)(k - t.width) / (double)t.orig) * base;
            const bool inside = tt > -kLowpassWidth && tt < kLowpassWidth;
            tt = std::min((double)kLowpassWidth, std::max(-(double)kLowpassWidth, tt));
            const double c = cos(tt * M_PI / kLowpassWidth / 2);
            const double window = c * c;
            tt *= M_PI;
            const double sinc = tt == 0.0 ? 1.0 : sin(tt) / tt;
            rows[p][k] = (float)(sinc * (window * scale));
            if (inside) { first = std::min(first, k); last = std::max(last, k); }   // clamped taps are ~1e-33: dropped
        }
        if (last < first) { first = 0; last = 0; }
        lo[p] = first; hi[p] = last;
        t.L = std::max(t.L, last - first + 1);
    }
    t.taps.assign((size_t)t.nw * t.L, 0.0f);
    t.first = lo;
    for (int p = 0; p < t.nw; ++p)
        for (int k = lo[p]; k <= hi[p]; ++k) t.taps[(size_t)p * t.L + (k - lo[p])] = rows[p][k];
    return t;
}

template <typename WT>
__global__ __launch_bounds__(256) void resample_kernel(const WT* __restrict__ wave, long long wave_stride, const int* __restrict__ lengths,
                                                       int max_len, const float* __restrict__ taps, const int* __restrict__ first, int orig,
                                                       int nw, int width, int L, float* __restrict__ out, long long out_stride,
                                                       int max_out_len, int* __restrict__ out_lengths) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int len = min(lengths ? lengths[b] : max_len, max_len);
    const long long target64 = ((long long)nw * len + orig - 1) / orig;                 // ceil(new * len / orig)
    const int target = (int)min(target64, (long long)max_out_len);
    if (j == 0 && out_lengths) out_lengths[b] = target;
    if (j >= max_out_len) return;
    float acc = 0.0f;
    if (j < target) {
        const int n = j / nw, p = j - n * nw;
        const int base = n * orig + first[p] - width;
        const WT* x = wave + (size_t)b * wave_stride;
        const float* tp = taps + (size_t)p * L;
        for (int l = 0; l < L; ++l) {
            const int i = base + l;
            float v = 0.0f;
            if (i >= 0 && i < len) {
                if constexpr (sizeof(WT) == 2) v = (float)x[i] * (1.0f / 32768.0f);
                else v = x[i];
            }
            acc = fmaf(tp[l], v, acc);
        }
    }
    out[(size_t)b * out_stride + j] = acc;
}

// interleaved [frames][channels] (i16: dequantised by 1/32768 as torchaudio.load does) -> mean over channels
template <typename WT>
__global__ __launch_bounds__(256) void mono_kernel(const WT* __restrict__ pcm, int channels, long long clip_stride, const int* __restrict__ frames,
                                                   int max_frames, float* __restrict__ out, long long out_stride) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= max_frames) return;
    const int n = min(frames ? frames[b] : max_frames, max_frames);
    float s = 0.0f;
    if (i < n) {
        const WT* x = pcm + (size_t)b * clip_stride + (size_t)i * channels;
        for (int c = 0; c < channels; ++c) {
            if constexpr (sizeof(WT) == 2) s += (float)x[c] * (1.0f / 32768.0f);
            else s += x[c];
        }
        if (channels > 1) s = s / (float)channels;
    }
    out[(size_t)b * out_stride + i] = s;
}

}  // namespace

static bool get_resample_table(sir_handle* h, int orig_freq, int new_freq, sir_resample_table* out) {
    for (auto& t : h->resample_tables)
        if (t.orig_freq == orig_freq && t.new_freq == new_freq) { *out = t; return true; }
    HostTable ht = build_table(orig_freq, new_freq);
    sir_resample_table t{};
    t.orig_freq = orig_freq; t.new_freq = new_freq;
    t.orig = ht.orig; t.nw = ht.nw; t.width = ht.width; t.L = ht.L;
    if (hipMalloc(&t.taps, ht.taps.size() * sizeof(float)) != hipSuccess) return false;
    if (hipMalloc(&t.first, ht.first.size() * sizeof(int)) != hipSuccess) { (void)hipFree(t.taps); return false; }
    if (hipMemcpy(t.taps, ht.taps.data(), ht.taps.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(t.first, ht.first.data(), ht.first.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(t.taps); (void)hipFree(t.first);
        return false;
    }
    h->resample_tables.push_back(t);
    *out = t;
    return true;
}

extern "C" int sir_resample_out_len(int length, int orig_freq, int new_freq) {
    if (length < 0 || orig_freq <= 0 || new_freq <= 0) return -1;
    const int g = gcd_int(orig_freq, new_freq);
    const long long orig = orig_freq / g, nw = new_freq / g;
    return (int)((nw * length + orig - 1) / orig);
}

extern "C" int sir_resample(sir_handle* h, const void* wave, int wave_dtype, int64_t wave_stride, const int32_t* lengths, int batch,
                            int max_len, int orig_freq, int new_freq, float* out, int64_t out_stride, int max_out_len,
                            int32_t* out_lengths, void* stream) {
    if (!h || !wave || !out) { sir_set_error("sir_resample: NULL argument"); return SIR_EINVAL; }
    if (batch <= 0 || max_len <= 0 || max_out_len <= 0 || orig_freq <= 0 || new_freq <= 0 || orig_freq == new_freq) {
        sir_set_error("sir_resample: bad sizes or rates (batch %d, max_len %d, %d -> %d Hz)", batch, max_len, orig_freq, new_freq);
        return SIR_EINVAL;
    }
    if (wave_dtype != SIR_WAVE_F32 && wave_dtype != SIR_WAVE_I16) { sir_set_error("sir_resample: unknown wave dtype %d", wave_dtype); return SIR_EINVAL; }
    sir_resample_table tab;
    sir_resample_table* t = &tab;
    if (!get_resample_table(h, orig_freq, new_freq, t)) { sir_set_error("sir_resample: could not build the %d -> %d Hz filter table", orig_freq, new_freq); return SIR_EHIP; }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((max_out_len + 255) / 256, batch);
    if (wave_dtype == SIR_WAVE_I16)
        hipLaunchKernelGGL(resample_kernel<short>, grid, dim3(256), 0, st, (const short*)wave, (long long)wave_stride, lengths, max_len,
                           (const float*)t->taps, (const int*)t->first, t->orig, t->nw, t->width, t->L, out, (long long)out_stride,
                           max_out_len, out_lengths);
    else
        hipLaunchKernelGGL(resample_kernel<float>, grid, dim3(256), 0, st, (const float*)wave, (long long)wave_stride, lengths, max_len,
                           (const float*)t->taps, (const int*)t->first, t->orig, t->nw, t->width, t->L, out, (long long)out_stride,
                           max_out_len, out_lengths);
    return sir_check_hip(hipGetLastError(), "resample_kernel");
}

extern "C" int sir_mix_to_mono(sir_handle* h, const void* pcm, int dtype, int channels, int64_t clip_stride, const int32_t* frames,
                               int batch, int max_frames, float* out, int64_t out_stride, void* stream) {
    if (!h || !pcm || !out) { sir_set_error("sir_mix_to_mono: NULL argument"); return SIR_EINVAL; }
    if (batch <= 0 || max_frames <= 0 || channels < 1 || channels > 64) {
        sir_set_error("sir_mix_to_mono: bad sizes (batch %d, max_frames %d, channels %d)", batch, max_frames, channels);
        return SIR_EINVAL;
    }
    if (dtype != SIR_WAVE_F32 && dtype != SIR_WAVE_I16) { sir_set_error("sir_mix_to_mono: unknown dtype %d", dtype); return SIR_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((max_frames + 255) / 256, batch);
    if (dtype == SIR_WAVE_I16)
        hipLaunchKernelGGL(mono_kernel<short>, grid, dim3(256), 0, st, (const short*)pcm, channels, (long long)clip_stride, frames, max_frames, out,
                           (long long)out_stride);
    else
        hipLaunchKernelGGL(mono_kernel<float>, grid, dim3(256), 0, st, (const float*)pcm, channels, (long long)clip_stride, frames, max_frames, out,
                           (long long)out_stride);
    return sir_check_hip(hipGetLastError(), "mono_kernel");
}

// ---- batch assembly from an HBM-resident feature store ------------------------------------------------------------------
// out[b] = store[index[b]] with the SpecAugment bands of dataset.py:160-176 zeroed on the way: one pass, float4 per thread
// (rows are t_pad floats, t_pad % 4 == 0).  A band {start, width} with width 0 is no band.
namespace {
__global__ __launch_bounds__(256) void gather_features_kernel(const float* __restrict__ store, const long long* __restrict__ index,
                                                              long long n_store, int n_mels, int t_pad, const int* __restrict__ time_mask,
                                                              const int* __restrict__ freq_mask, float* __restrict__ out,
                                                              unsigned int* __restrict__ status) {
    const int b = blockIdx.y;
    long long src = index[b];
    const bool bad = src < 0 || src >= n_store;
    if (bad) {                                              // an index outside the store: zeros + the handle's status word (SIR_EINVAL at the next check)
        if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_fetch_or(status, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        src = 0;
    }
    const int per = n_mels * t_pad / 4;
    const float4* in4 = reinterpret_cast<const float4*>(store + (size_t)src * n_mels * t_pad);
    float4* out4 = reinterpret_cast<float4*>(out + (size_t)b * n_mels * t_pad);
    int t0 = 0, tw = 0, f0 = 0, fw = 0;
    if (time_mask) { t0 = time_mask[2 * b]; tw = time_mask[2 * b + 1]; }
    if (freq_mask) { f0 = freq_mask[2 * b]; fw = freq_mask[2 * b + 1]; }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < per; i += gridDim.x * 256) {
        const int mel = (i * 4) / t_pad, t = (i * 4) - mel * t_pad;
        float4 v = bad ? make_float4(0.f, 0.f, 0.f, 0.f) : in4[i];
        if (mel >= f0 && mel < f0 + fw) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tw > 0) {
            if (t >= t0 && t < t0 + tw) v.x = 0.0f;
            if (t + 1 >= t0 && t + 1 < t0 + tw) v.y = 0.0f;
            if (t + 2 >= t0 && t + 2 < t0 + tw) v.z = 0.0f;
            if (t + 3 >= t0 && t + 3 < t0 + tw) v.w = 0.0f;
        }
        out4[i] = v;
    }
}
}  // namespace

extern "C" int sir_gather_features(sir_handle* h, const float* store, int64_t n_store, const int64_t* index, int batch, int n_mels,
                                   int t_pad, const int32_t* time_mask, const int32_t* freq_mask, float* out, void* stream) {
    if (!h || !store || !index || !out) { sir_set_error("sir_gather_features: NULL argument"); return SIR_EINVAL; }
    if (batch <= 0 || batch > 65535 || n_store <= 0 || n_mels <= 0 || t_pad <= 0 || (t_pad & 3) != 0) {
        sir_set_error("sir_gather_features: bad sizes (batch %d, store %lld, n_mels %d, t_pad %d: t_pad must be a multiple of 4)", batch,
                      (long long)n_store, n_mels, t_pad);
        return SIR_EINVAL;
    }
    const int per = n_mels * t_pad / 4;
    int gx = (per + 255) / 256;
    gx = gx > 16 ? 16 : gx;
    hipLaunchKernelGGL(gather_features_kernel, dim3(gx, batch), dim3(256), 0, (hipStream_t)stream, store, (const long long*)index,
                       (long long)n_store, n_mels, t_pad, (const int*)time_mask, (const int*)freq_mask, out, h->status);
    return sir_check_hip(hipGetLastError(), "gather_features_kernel");
}
