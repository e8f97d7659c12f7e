// Training-only device kernels (forward pieces that differ from inference, backward, optimiser).
// Included by model_train.hip.  Layouts as in model_kernels.h (NHWC activations).
#pragma once
#include "model_kernels.h"
#include "bf16x6_kernels.h"
#include "conv_wino_bf16x6_kernel.h"
#include "gru_frag_prep.h"

// ------------------------------------------------------------------------------------------
// BatchNorm with batch statistics
// ------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------
// conv1 BatchNorm statistics WITHOUT computing conv1: z_c = sum_t w_c[t] x_t with x_t the nine shifted copies of the
// (zero-padded) feature image, hence
//     sum z_c   = sum_t w_c[t] S[t]                       S[t]     = sum_pixels x_t
//     sum z_c^2 = sum_{t,u} w_c[t] w_c[u] R[t][u]         R[t][u]  = sum_pixels x_t x_u
// -- 9 + 45 moments of the INPUT, the same for all 32 channels (one pass over 13 MB instead of recomputing 105 M conv
// outputs x 32 channels).  The backward reuses them for the mean terms of the weight gradient (conv1_bwd_finalize_kernel).
// Moment layout M[54]: S[0..8], then R packed by rows, t <= u: index 9 + t*9 - t*(t-1)/2 + (u - t).
// ------------------------------------------------------------------------------------------
constexpr int C1_NMOM = 54;
__host__ __device__ constexpr int c1_r_index(int t, int u) { return 9 + t * 9 - t * (t - 1) / 2 + (u - t); }   // t <= u

// wave WV of a block accumulates the moments [14 WV, 14 WV + 14) -- over ALL pixels of a tile -- so that each of the 54 sums
// is reduced across 64 lanes exactly once per block (with 54 accumulators in every thread the wave reductions cost more than
// the accumulation itself)
constexpr int C1_MPW = 14;
template <int WV>
__device__ __forceinline__ void c1_moments_accum(const float (&v)[9], float (&m)[C1_MPW]) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (t / C1_MPW == WV) m[t - C1_MPW * WV] += v[t];
#pragma unroll
        for (int u = t; u < 9; ++u)
            if (c1_r_index(t, u) / C1_MPW == WV) m[c1_r_index(t, u) - C1_MPW * WV] = fmaf(v[t], v[u], m[c1_r_index(t, u) - C1_MPW * WV]);
    }
}

// block (g, b) walks the 8 x 64-pixel tiles g, g + gridDim.x, ... of image b; part[54][gridDim.y * gridDim.x] (moment-major: the
// reduce kernel then reads contiguous rows -- block-major rows of 54 floats made it a 216-byte-stride gather, 37 us for 442 KB)
static __global__ __launch_bounds__(256) void conv1_moments_kernel(const float* __restrict__ x, float* __restrict__ part, int H, int W,
                                                                    int tiles_x, int tiles_y) {
    __shared__ float tile[C1_TR * C1_TC];
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* xb = x + (size_t)b * H * W;
    float m[C1_MPW];
#pragma unroll
    for (int i = 0; i < C1_MPW; ++i) m[i] = 0.0f;
    for (int tl = blockIdx.x; tl < tiles_x * tiles_y; tl += gridDim.x) {
        const int y0 = 2 * (tl / tiles_x) * C1_PROWS, x0 = 2 * (tl % tiles_x) * C1_PCOLS;
        __syncthreads();                                   // previous tile consumed
        for (int i = tid; i < C1_TR * C1_TC; i += 256) {
            const int ty = i / C1_TC, tx = i - ty * C1_TC;
            const int gy = y0 - 1 + ty, gx = x0 - 1 + tx;
            tile[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? xb[(size_t)gy * W + gx] : 0.0f;
        }
        __syncthreads();
        for (int k = 0; k < (4 * C1_PROWS * C1_PCOLS) / 64; ++k) {          // 512 pixels / 64 lanes
            const int pix = lane + 64 * k, ly = pix / (2 * C1_PCOLS), lx = pix % (2 * C1_PCOLS);
            if (y0 + ly >= H || x0 + lx >= W) continue;
            float v[9];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) v[ky * 3 + kx] = tile[(ly + ky) * C1_TC + lx + kx];
            switch (wv) {                                   // wave-uniform
                case 0: c1_moments_accum<0>(v, m); break;
                case 1: c1_moments_accum<1>(v, m); break;
                case 2: c1_moments_accum<2>(v, m); break;
                default: c1_moments_accum<3>(v, m); break;
            }
        }
    }
    const size_t blk = (size_t)b * gridDim.x + blockIdx.x, nblk = (size_t)gridDim.x * gridDim.y;
#pragma unroll
    for (int i = 0; i < C1_MPW; ++i) {
        float a = m[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if (lane == 0 && C1_MPW * wv + i < C1_NMOM) part[(size_t)(C1_MPW * wv + i) * nblk + blk] = a;
    }
}

// M[i] = sum over blocks (double); one block per moment
static __global__ __launch_bounds__(256) void conv1_moments_reduce_kernel(const float* __restrict__ part, int nblk, double* __restrict__ M) {
    __shared__ double rs[256];
    const int i = blockIdx.x, tid = threadIdx.x;
    double s = 0.0;
#pragma unroll 8
    for (int r = tid; r < nblk; r += 256) s += part[(size_t)i * nblk + r];
    rs[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) rs[tid] += rs[tid + o];
        __syncthreads();
    }
    if (tid == 0) M[i] = rs[0];
}

__device__ __forceinline__ void bn_finalize_channel(double sum, double sumsq, double count, int c, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float* __restrict__ run_mean,
                                                    float* __restrict__ run_var, float momentum, float* __restrict__ scale,
                                                    float* __restrict__ shift, float* __restrict__ save_mean,
                                                    float* __restrict__ save_invstd) {
    const double mean = sum / count;
    double var = sumsq / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)SIR_BN_EPS));
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    save_mean[c] = (float)mean;
    save_invstd[c] = invstd;
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * (float)mean;
    run_var[c] = (1.0f - momentum) * run_var[c] + momentum * (float)unbiased;
}

// per channel: (sum z, sum z^2) from the moments and the nine weights, then the usual BatchNorm finalisation; 32 threads
static __global__ void conv1_bn_from_moments_kernel(const double* __restrict__ M, const float* __restrict__ w, double count,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float* __restrict__ run_mean, float* __restrict__ run_var, float momentum,
                                                    float* __restrict__ scale, float* __restrict__ shift,
                                                    float* __restrict__ save_mean, float* __restrict__ save_invstd) {
    const int c = threadIdx.x;
    if (c >= 32) return;
    double wk[9];
    for (int t = 0; t < 9; ++t) wk[t] = (double)w[c * 9 + t];
    double sum = 0.0, sumsq = 0.0;
    for (int t = 0; t < 9; ++t) {
        sum += wk[t] * M[t];
        sumsq += wk[t] * wk[t] * M[c1_r_index(t, t)];
        for (int u = t + 1; u < 9; ++u) sumsq += 2.0 * wk[t] * wk[u] * M[c1_r_index(t, u)];
    }
    bn_finalize_channel(sum, sumsq, count, c, gamma, beta, run_mean, run_var, momentum, scale, shift, save_mean, save_invstd);
}

// partial (sum, sumsq) [nblk][C] -> batch mean / biased var -> folded scale/shift for the forward,
// saved mean / invstd for the backward, running statistics updated in place
// (momentum 0.1, unbiased variance: torch.nn.BatchNorm2d training semantics).  One block per channel.
static __global__ __launch_bounds__(256) void bn_finalize_kernel(const float2* __restrict__ stats, int nblk, int C, double count,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ run_mean, float* __restrict__ run_var,
                                                           float momentum, float* __restrict__ scale,
                                                           float* __restrict__ shift, float* __restrict__ save_mean,
                                                           float* __restrict__ save_invstd) {
    __shared__ double rs[256], rq[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    double s = 0.0, q = 0.0;
#pragma unroll 4
    for (int i = tid; i < nblk; i += 256) {
        const float2 v = stats[(size_t)i * C + c];
        s += v.x;
        q += v.y;
    }
    rs[tid] = s; rq[tid] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { rs[tid] += rs[tid + o]; rq[tid] += rq[tid + o]; }
        __syncthreads();
    }
    if (tid == 0)
        bn_finalize_channel(rs[0], rq[0], count, c, gamma, beta, run_mean, run_var, momentum, scale, shift, save_mean, save_invstd);
}

// z (raw conv output, NHWC [B][H][W][C]) -> relu(bn(z)) -> 2x2 max-pool.
// GRU_OUT = false: NHWC [B][Hp][Wp][C];  true: [B][Wp][C*Hp] with feature = c*Hp + py (models.py:55-57)
template <bool GRU_OUT>
__global__ __launch_bounds__(256) void bn_relu_pool_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* __restrict__ out,
                                                            int B, int H, int W, int C, int Hp, int Wp) {
    const int c4n = C / 4;
    const size_t total = (size_t)B * Hp * Wp * c4n;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        int c4, px, py, b;
        if (!GRU_OUT) {
            c4 = idx % c4n;
            size_t rest = idx / c4n;
            px = rest % Wp; rest /= Wp;
            py = rest % Hp;
            b = rest / Hp;
        } else {                                      // py fastest: the lanes of a wave then fill whole 32-byte sectors of
            py = idx % Hp;                            // the [c*Hp + py] feature rows (c4 fastest scattered single floats)
            size_t rest = idx / Hp;
            c4 = rest % c4n; rest /= c4n;
            px = rest % Wp;
            b = rest / Wp;
        }
        const float4 s = *reinterpret_cast<const float4*>(scale + c4 * 4);
        const float4 t = *reinterpret_cast<const float4*>(shift + c4 * 4);
        float4 best = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const float4 v = *reinterpret_cast<const float4*>(
                    z + (((size_t)b * H + 2 * py + dy) * W + 2 * px + dx) * C + c4 * 4);
                best.x = fmaxf(best.x, fmaf(v.x, s.x, t.x));
                best.y = fmaxf(best.y, fmaf(v.y, s.y, t.y));
                best.z = fmaxf(best.z, fmaf(v.z, s.z, t.z));
                best.w = fmaxf(best.w, fmaf(v.w, s.w, t.w));
            }
        if (!GRU_OUT) {
            *reinterpret_cast<float4*>(out + (((size_t)b * Hp + py) * Wp + px) * C + c4 * 4) = best;
        } else {
            float* o = out + ((size_t)b * Wp + px) * ((size_t)C * Hp) + (size_t)(c4 * 4) * Hp + py;
            o[0] = best.x; o[Hp] = best.y; o[2 * Hp] = best.z; o[3 * Hp] = best.w;
        }
    }
}

// ------------------------------------------------------------------------------------------
// inter-layer GRU dropout (models.py:32, p = 0.5 in train()): counter-based keep mask, a pure
// function of (seed, element index) so the backward pass regenerates it.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool dropout_keep(unsigned long long seed, size_t idx, float p) {
    unsigned long long x = seed ^ (idx * 0x9E3779B97F4A7C15ull);
    x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull; x ^= x >> 33;
    return (float)(unsigned)(x >> 40) * (1.0f / 16777216.0f) >= p;
}

// out = dropout(in) AND the f16x2 planes [2][n] of out (the A operand of the next layer's input projection) in one pass;
// one thread = 8 consecutive elements (n is a multiple of 8: rows of 512)
static __global__ __launch_bounds__(256) void dropout_split2h_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                      unsigned short* __restrict__ planes, size_t n, float p,
                                                                      unsigned long long seed) {
    const float sc = 1.0f / (1.0f - p);
    for (size_t i8 = (size_t)blockIdx.x * 256 + threadIdx.x; i8 < n / 8; i8 += (size_t)gridDim.x * 256) {
        const size_t i = i8 * 8;
        float v[8];
        *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(in + i);
        *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(in + i + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = dropout_keep(seed, i + e, p) ? v[e] * sc : 0.0f;
        const float4 v0 = make_float4(v[0], v[1], v[2], v[3]), v1 = make_float4(v[4], v[5], v[6], v[7]);
        *reinterpret_cast<float4*>(out + i) = v0;
        *reinterpret_cast<float4*>(out + i + 4) = v1;
        uint2 h0, l0, h1, l1;
        split2h_quad(v0, h0, l0);
        split2h_quad(v1, h1, l1);
        *reinterpret_cast<uint4*>(planes + i) = make_uint4(h0.x, h0.y, h1.x, h1.y);
        *reinterpret_cast<uint4*>(planes + n + i) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}

// ------------------------------------------------------------------------------------------
// cross-entropy (mean) forward + gradient wrt logits:  loss = -mean_b log softmax(l_b)[y_b],
// dlogits = (softmax - onehot) * grad_scale / (rows whose label is not ignore_index)      (nn.CrossEntropyLoss(), train.py:242)
// single workgroup, deterministic reduction
// ------------------------------------------------------------------------------------------
template <int CMAX>
static __global__ __launch_bounds__(256) void ce_loss_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                       int B, int C, float* __restrict__ loss, float* __restrict__ dlogits,
                                                       float grad_scale, unsigned int* status) {
    __shared__ float red[256];
    __shared__ int cnt[256];
    // nn.CrossEntropyLoss() has ignore_index = -100 by default: such a row contributes neither loss nor gradient and the mean is
    // taken over the remaining rows (all rows ignored: 0 / 0 = NaN, as torch)
    int nv = 0;
    for (int b = threadIdx.x; b < B; b += 256) nv += labels[b] != -100ll ? 1 : 0;
    cnt[threadIdx.x] = nv;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) cnt[threadIdx.x] += cnt[threadIdx.x + o];
        __syncthreads();
    }
    const float nvalid = (float)cnt[0];
    float acc = 0.0f;
    for (int b = threadIdx.x; b < B; b += 256) {
        // the row goes into registers with ALL its loads in flight (C <= CMAX, dispatched by the host): the per-class loops of
        // the first version waited for one dependent load after the other, 15 us for 256 x 31 logits
        const float* r = logits + (size_t)b * C;
        // any OTHER label outside [0, C) (nn.CrossEntropyLoss raises on it; train.py:242): flag the handle's status word (bit 1 ->
        // SIR_EINVAL at the next sir_check_status) and make the loss NaN instead of reading out of bounds
        const long long yl = labels[b];
        const bool ignored = yl == -100ll;
        const bool bad = !ignored && (yl < 0 || yl >= (long long)C);
        if (bad) __hip_atomic_fetch_or(status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int y = (bad || ignored) ? 0 : (int)yl;
        float v[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) v[c] = c < C ? r[c] : 0.0f;
        const float ry = r[y];
        float mx = v[0];
#pragma unroll
        for (int c = 1; c < CMAX; ++c) if (c < C) mx = fmaxf(mx, v[c]);
        float den = 0.0f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) if (c < C) { v[c] = expf(v[c] - mx); den += v[c]; }
        const float lse = mx + logf(den);
        acc += bad ? __builtin_nanf("") : (ignored ? 0.0f : lse - ry);
        if (dlogits) {
            const float inv = 1.0f / den, gs = ignored ? 0.0f : grad_scale / nvalid;
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c < C) dlogits[(size_t)b * C + c] = ignored ? 0.0f : (v[c] * inv - (c == y ? 1.0f : 0.0f)) * gs;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = red[0] / nvalid;
}
// ------------------------------------------------------------------------------------------
// head backward: fc + attention pooling (models.py:63-67)
//   dctx = dlogits fc_w;  w = softmax_t(y a + b);  dw_t = <dctx, y_t>;  ds_t = w_t (dw_t - sum w dw)
//   dy_t = w_t dctx + ds_t a;  per-utterance partials of d attention.weight / d attention.bias
// one workgroup per utterance
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void head_bwd_utt(int b, const float* __restrict__ dlogits, const float* __restrict__ fcw,
                                             const float* __restrict__ y, const float* __restrict__ aw,
                                             const float* __restrict__ ab, float* __restrict__ dy,
                                             float* __restrict__ daw_part, float* __restrict__ dab_part,
                                             int S, int C, float gscale) {
    __shared__ float dctx[512];
    __shared__ float sc[ATT_MAX_S], ds[ATT_MAX_S];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* yb = y + (size_t)b * S * 512;
    for (int c = tid; c < 512; c += 256) {
        float a = 0.0f;
        for (int j = 0; j < C; ++j) a = fmaf(dlogits[(size_t)b * C + j], fcw[(size_t)j * 512 + c], a);
        dctx[c] = a;
    }
    float a8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a8[i] = aw[lane + 64 * i];
    __syncthreads();
    float dc8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) dc8[i] = dctx[lane + 64 * i];
    // four time steps per wave and round, all 32 loads issued before the first reduction (cf. attention_pool_kernel)
    for (int t0 = 4 * wv; t0 < S; t0 += 16) {
        float v[4][8];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[k][i] = (t0 + k < S) ? yb[(size_t)(t0 + k) * 512 + lane + 64 * i] : 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float d = 0.0f, g = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                d = fmaf(v[k][i], a8[i], d);
                g = fmaf(v[k][i], dc8[i], g);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { d += __shfl_xor(d, o); g += __shfl_xor(g, o); }
            if (lane == 0 && t0 + k < S) { sc[t0 + k] = d + ab[0]; ds[t0 + k] = g; }   // ds holds dw_t for now
        }
    }
    __syncthreads();
    // softmax weights once per time step (thread t)
    __shared__ float ex[ATT_MAX_S];
    float mx = -INFINITY;
    for (int t = 0; t < S; ++t) mx = fmaxf(mx, sc[t]);
    for (int t = tid; t < S; t += 256) ex[t] = expf(sc[t] - mx);
    __syncthreads();
    float den = 0.0f;
    for (int t = 0; t < S; ++t) den += ex[t];
    float wdw = 0.0f;
    for (int t = 0; t < S; ++t) wdw = fmaf(ex[t] / den, ds[t], wdw);
    __syncthreads();
    for (int t = tid; t < S; t += 256) {
        const float w = ex[t] / den;
        const float dst = w * (ds[t] - wdw);
        sc[t] = w;                 // now the attention weight
        ds[t] = dst;               // now d score
    }
    __syncthreads();
    float dsum = 0.0f;
    for (int t = 0; t < S; ++t) dsum += ds[t];
    for (int c = tid; c < 512; c += 256) {
        const float dc = dctx[c], ac = aw[c];
        float da = 0.0f;
        for (int t = 0; t < S; ++t) {
            dy[((size_t)b * S + t) * 512 + c] = gscale * fmaf(sc[t], dc, ds[t] * ac);     // (the backward's loss scale: a power of two, exact)
            da = fmaf(ds[t], yb[(size_t)t * 512 + c], da);
        }
        daw_part[(size_t)b * 512 + c] = da;
    }
    if (tid == 0) dab_part[b] = dsum;
}

// out[n] = sum_r in[r][n]   (deterministic column sums: bias / attention gradients)
static __global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ in, int rows, int ld, int n_cols,
                                                      float* __restrict__ out) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    float a = 0.0f;
    if (col < n_cols) {
#pragma unroll 8
        for (int r = part; r < rows; r += 4) a += in[(size_t)r * ld + col];
    }
    red[part][threadIdx.x & 63] = a;
    __syncthreads();
    if (part == 0 && col < n_cols) out[col] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// attention.weight (512 column sums of daw_part [rows][512]) and attention.bias (the sum of dab_part [rows]) in ONE launch:
// blocks 0..7 take 64 columns each, block 8 the bias
static __global__ __launch_bounds__(256) void head_colsum_kernel(const float* __restrict__ daw_part, const float* __restrict__ dab_part,
                                                                  int rows, float* __restrict__ out_w, float* __restrict__ out_b) {
    __shared__ float red[4][64];
    const bool bias = blockIdx.x == 8;
    const int l = threadIdx.x & 63, col = blockIdx.x * 64 + l, part = threadIdx.x >> 6;
    float a = 0.0f;
    if (!bias) {
#pragma unroll 8
        for (int r = part; r < rows; r += 4) a += daw_part[(size_t)r * 512 + col];     // 8 loads in flight, added in row order
    }
    else { for (int r = threadIdx.x; r < rows; r += 256) a += dab_part[r]; }
    if (bias) {                                   // 256 partial sums -> one value, fixed order
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    }
    red[part][l] = a;
    __syncthreads();
    if (part == 0 && (!bias || l == 0)) {
        const float v = red[0][l] + red[1][l] + red[2][l] + red[3][l];
        if (bias) out_b[0] = v; else out_w[col] = v;
    }
}

// the four GRU bias gradients of one layer in ONE launch: column sums of bsum_i / bsum_h [rows][1536] (blockIdx.y selects
// the array), columns [0, 768) -> direction 0, [768, 1536) -> direction 1
static __global__ __launch_bounds__(256) void gru_bias_colsum_kernel(const float* __restrict__ bsum_i, const float* __restrict__ bsum_h, int rows,
                                                                      float* __restrict__ bi0, float* __restrict__ bi1,
                                                                      float* __restrict__ bh0, float* __restrict__ bh1, float unscale) {
    __shared__ float red[4][64];
    const float* in = blockIdx.y ? bsum_h : bsum_i;
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;     // col < 1536 (grid.x = 24)
    float a = 0.0f;
#pragma unroll 8
    for (int r = part; r < rows; r += 4) a += in[(size_t)r * 1536 + col];
    red[part][threadIdx.x & 63] = a;
    __syncthreads();
    if (part == 0) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        float* out = blockIdx.y ? (col < 768 ? bh0 : bh1) : (col < 768 ? bi0 : bi1);
        out[col < 768 ? col : col - 768] = v * unscale;
    }
}

// first stage for tall inputs: out[chunk][n] = sum over the rows of chunk `blockIdx.y` (then colsum_kernel)
static __global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ in, int rows, int ld, int n_cols,
                                                                     float* __restrict__ out) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    const int per = (rows + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
    float a = 0.0f;
    if (col < n_cols) {
#pragma unroll 8
        for (int r = r0 + part; r < r1; r += 4) a += in[(size_t)r * ld + col];
    }
    red[part][threadIdx.x & 63] = a;
    __syncthreads();
    if (part == 0 && col < n_cols)
        out[(size_t)blockIdx.y * n_cols + col] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// dfc_w[j][c] = sum_b dlogits[b][j] ctx[b][c];  dfc_b[j] = sum_b dlogits[b][j]
//   grid (C, 2): block (j, half) owns 256 columns c; the dlogits column is staged in LDS once and the loop over
//   the batch is unrolled so that the ctx loads pipeline (the first version ran one dependent load per iteration)
__device__ __forceinline__ void fc_wgrad_block(int j, int half, const float* __restrict__ dlogits, const float* __restrict__ ctx,
                                               float* __restrict__ dw, float* __restrict__ db, int B, int C) {
    __shared__ float dl[1024];
    const int c = half * 256 + threadIdx.x;
    float a = 0.0f, sb = 0.0f;
    for (int b0 = 0; b0 < B; b0 += 1024) {
        const int nb = min(1024, B - b0);
        __syncthreads();
        for (int i = threadIdx.x; i < nb; i += 256) dl[i] = dlogits[(size_t)(b0 + i) * C + j];
        __syncthreads();
        int b = 0;
        for (; b + 8 <= nb; b += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ctx[(size_t)(b0 + b + u) * 512 + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) { a = fmaf(dl[b + u], v[u], a); sb += dl[b + u]; }
        }
        for (; b < nb; ++b) { a = fmaf(dl[b], ctx[(size_t)(b0 + b) * 512 + c], a); sb += dl[b]; }
    }
    dw[(size_t)j * 512 + c] = a;
    if (c == 0) db[j] = sb;
}

// head backward, ONE launch for two independent jobs (both only need dlogits): workgroups [0, B) = one utterance each
// (head_bwd_utt: fc + attention pooling backward, per-utterance attention partials; its dy output -- and with it everything the
// rest of the backward computes -- carries the loss scale `gscale`, see sir_bwd_loss_scale); workgroups [B, B + 2 C) = the fc weight /
// bias gradients (fc_wgrad_block).  head_colsum_kernel adds the attention partials afterwards.
static __global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dlogits, const float* __restrict__ fcw,
                                                              const float* __restrict__ y, const float* __restrict__ aw,
                                                              const float* __restrict__ ab, const float* __restrict__ ctx,
                                                              float* __restrict__ dy, float* __restrict__ daw_part,
                                                              float* __restrict__ dab_part, float* __restrict__ d_fc_w,
                                                              float* __restrict__ d_fc_b, int B, int S, int C, float gscale) {
    if ((int)blockIdx.x >= B) {
        const int i = blockIdx.x - B;
        fc_wgrad_block(i >> 1, i & 1, dlogits, ctx, d_fc_w, d_fc_b, B, C);
        return;
    }
    head_bwd_utt(blockIdx.x, dlogits, fcw, y, aw, ab, dy, daw_part, dab_part, S, C, gscale);
}

// ------------------------------------------------------------------------------------------
// GRU back-propagation through time, one layer, both directions (mirror of gru_recurrence_kernel).
//   dy    [B][S][512]  gradient wrt the layer output (both directions)
//   gates [B][S][2][4][256] saved (r, z, n, W_hn h + b_hn);  y = layer output (h_prev = neighbour step)
//   wr4   [2][192][256][4]  W_hh regrouped by 4 gate rows (prep_whh_bwd_kernel): lane = k
//   dgi / dgh [B*S][1536]   gradients wrt the input-side / hidden-side gate pre-activations
// Thread (u, b): gate math for hidden unit u of utterance b; thread (k, rs): partial of W_hh^T dgh
// over gate rows [192 rs, 192 rs + 192).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void prep_whh_bwd_elem(const float* __restrict__ w, float* __restrict__ wr4, int idx) {
    if (idx >= 768 * 256) return;                                 // over 768*256, layout [row/4][k][4]
    const int e = idx & 3, k = (idx >> 2) & 255, r4 = idx >> 10;
    wr4[idx] = w[(size_t)(r4 * 4 + e) * 256 + k];
}

// Every per-step re-layout of the weights (they change with each optimizer step) in ONE launch: a dozen ~5 us launches
// otherwise.  kind 0: split2h_rows (a = ld_in = K, b = rows), 1: prep_conv_w_bf16x3 (a = cin, b = cout),
// 2: prep_conv_wT_bf16x3 (a = cin_f, b = cout_f), 3: prep_whh_bwd, 4: prep_conv_w_wino_bf16x3 (a = cin, b = cout),
// 5: prep_conv_wT_wino_bf16x3 (a = cin_f, b = cout_f), 6 / 7: the f16x3 forms of 4 / 5 (conv_wino_bf16x6_kernel.h), 8 / 9: W_hh as
// the resident fragments of the forward / backward cluster recurrence (gru_frag_prep.h).  Job j owns blocks [block0[j], block0[j+1]).
constexpr int PREP_MAX_JOBS = 20;
struct PrepJobs {
    const float* src[PREP_MAX_JOBS];
    void* dst[PREP_MAX_JOBS];
    int kind[PREP_MAX_JOBS], a[PREP_MAX_JOBS], b[PREP_MAX_JOBS];
    int block0[PREP_MAX_JOBS + 1];
    int njobs;
    unsigned int* status;        // the handle's status word (kinds 6 / 7 flag weights outside the f16x3 range)
};
static __global__ __launch_bounds__(256) void train_prep_kernel(PrepJobs jobs) {
    int j = 0;
    while (j + 1 < jobs.njobs && (int)blockIdx.x >= jobs.block0[j + 1]) ++j;
    const int lb = blockIdx.x - jobs.block0[j], nb = jobs.block0[j + 1] - jobs.block0[j];
    const int idx = lb * 256 + threadIdx.x;
    const float* __restrict__ src = jobs.src[j];
    switch (jobs.kind[j]) {
        case 0: split2h_rows(src, jobs.a[j], (unsigned short*)jobs.dst[j], (size_t)jobs.b[j], jobs.a[j], (size_t)idx, (size_t)nb * 256); break;
        case 1: prep_conv_w_bf16x3_elem(src, (unsigned short*)jobs.dst[j], jobs.a[j], jobs.b[j], idx); break;
        case 2: prep_conv_wT_bf16x3_elem(src, (unsigned short*)jobs.dst[j], jobs.a[j], jobs.b[j], idx); break;
        case 4: prep_conv_w_wino_bf16x3_elem(src, (unsigned short*)jobs.dst[j], jobs.a[j], jobs.b[j], idx); break;
        case 5: prep_conv_wT_wino_bf16x3_elem(src, (unsigned short*)jobs.dst[j], jobs.a[j], jobs.b[j], idx); break;
        case 6: prep_conv_w_wino_f16x3_elem(src, (unsigned short*)jobs.dst[j], jobs.a[j], jobs.b[j], idx, jobs.status); break;
        case 7: prep_conv_wT_wino_f16x3_elem(src, (unsigned short*)jobs.dst[j], jobs.a[j], jobs.b[j], idx, jobs.status); break;
        case 8: prep_whh_quad_elem(src, (uint4*)jobs.dst[j], idx); break;
        case 9: prep_whh_bwd_quad_elem(src, (uint4*)jobs.dst[j], idx); break;
        default: prep_whh_bwd_elem(src, (float*)jobs.dst[j], idx); break;
    }
}

// Reads two buffers and keeps nothing: a software prefetch into the last-level cache, launched on the backward's side stream (model_train.hip:
// layer 0's saved gates and outputs ahead of its BPTT)
static __global__ __launch_bounds__(256) void cache_touch_kernel(const float4* __restrict__ a, size_t na, const float4* __restrict__ b, size_t nb,
                                                                  float* __restrict__ sink) {
    float acc = 0.0f;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < na; i += stride) { const float4 v = a[i]; acc += v.x + v.y + v.z + v.w; }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += stride) { const float4 v = b[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123456.789f) sink[0] = acc;               // (never true in practice; keeps the loads)
}

// out_j[i] = sum_z slabs_j[z][i] for up to four jobs (blockIdx.y) with a common slab count; slab stride = n_j
struct SlabJobs { const float* src[4]; float* out[4]; size_t n[4]; };
static __global__ void slab_reduce_jobs_kernel(SlabJobs jobs, int nslab, float unscale) {
    const int j = blockIdx.y;
    const size_t n = jobs.n[j];
    const float* __restrict__ s = jobs.src[j];
    float* __restrict__ o = jobs.out[j];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float a = 0.0f;
#pragma unroll 8
        for (int z = 0; z < nslab; ++z) a += s[(size_t)z * n + i];
        o[i] = a * unscale;
    }
}

constexpr int GRU_BBW = 4;      // utterances per workgroup of the backward recurrence

// ------------------------------------------------------------------------------------------
// backward through max-pool -> ReLU -> BatchNorm (batch statistics).
//   da: gradient wrt the pooled output (NHWC [B][Hp][Wp][C], or the GRU layout when GRU_IN)
//   z : raw conv output [B][H][W][C];  y = z*scale+shift
//   dy(pixel) = da if the pixel is the first maximum of its 2x2 window and y > 0, else 0
// pass 1 (bn_bwd_reduce): per-block partial sums of dy and dy*xhat per channel (-> dbeta, dgamma)
// pass 2 (bn_bwd_dz): dz = gamma*invstd*(dy - mean(dy) - xhat*mean(dy*xhat)) at full resolution
// ------------------------------------------------------------------------------------------
template <bool GRU_IN>
__device__ __forceinline__ float4 load_da4(const float* __restrict__ da, int b, int py, int px, int c4, int Hp, int Wp,
                                           int C) {
    if (!GRU_IN) return *reinterpret_cast<const float4*>(da + (((size_t)b * Hp + py) * Wp + px) * C + c4 * 4);
    const float* o = da + ((size_t)b * Wp + px) * ((size_t)C * Hp) + (size_t)(c4 * 4) * Hp + py;
    return make_float4(o[0], o[Hp], o[2 * Hp], o[3 * Hp]);
}

// gradient reaching pixel (dy,dx) of the window, per channel component
__device__ __forceinline__ float route1(float y00, float y01, float y10, float y11, int pos, float g) {
    float best = y00; int arg = 0;
    if (y01 > best) { best = y01; arg = 1; }
    if (y10 > best) { best = y10; arg = 2; }
    if (y11 > best) { best = y11; arg = 3; }
    return (arg == pos && best > 0.0f) ? g : 0.0f;
}

// The same two sums from the POOLED activations a = pool(relu(bn(z))): dy = da wherever a > 0 (the window's maximum passed
// the ReLU; a zero gradient otherwise), and at that maximum y = a, hence xhat = (y - beta) / gamma = (a - beta) / gamma.
// Reads a and da (2 x 52 MB for conv2) instead of z and da (210 + 52 MB).  A channel whose gamma is too small for the
// division (|gamma| < 1e-2 (1 + |beta|): rounding of a would be amplified) takes the z path for its windows.
__device__ __forceinline__ bool bn_gamma_ok(float gamma, float beta) { return fabsf(gamma) >= 1e-2f * (1.0f + fabsf(beta)); }

// NHWC activations [B][Hp][Wp][C]; same block / thread mapping and partial layout as bn_bwd_reduce_kernel<false>
static __global__ __launch_bounds__(256) void bn_bwd_reduce_pooled_kernel(
    const float* __restrict__ a, const float* __restrict__ da, const float* __restrict__ z, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, float2* __restrict__ part, int B, int H, int W, int C, int Hp, int Wp, int pix_per_block) {
    __shared__ float rs[256], rq[256];
    const int c4n = C / 4;
    const int lanes_c = c4n < 64 ? c4n : 64;
    const int pl = 256 / lanes_c;
    const int c4 = threadIdx.x % lanes_c, pslot = threadIdx.x / lanes_c;
    const size_t npix = (size_t)B * Hp * Wp;
    const size_t p0 = (size_t)blockIdx.x * pix_per_block;
    for (int cc = c4; cc < c4n; cc += lanes_c) {
        const float4 gm = *reinterpret_cast<const float4*>(gamma + cc * 4), bt = *reinterpret_cast<const float4*>(beta + cc * 4);
        const bool fast = bn_gamma_ok(gm.x, bt.x) && bn_gamma_ok(gm.y, bt.y) && bn_gamma_ok(gm.z, bt.z) && bn_gamma_ok(gm.w, bt.w);
        float4 sdy = make_float4(0.f, 0.f, 0.f, 0.f), sdx = sdy;
        if (fast) {
            const float4 rg = make_float4(1.0f / gm.x, 1.0f / gm.y, 1.0f / gm.z, 1.0f / gm.w);
            for (int i = pslot; i < pix_per_block; i += pl) {
                const size_t p = p0 + i;
                if (p >= npix) break;
                const float4 av = *reinterpret_cast<const float4*>(a + p * C + cc * 4);
                const float4 g = *reinterpret_cast<const float4*>(da + p * C + cc * 4);
                const float dx_ = av.x > 0.0f ? g.x : 0.0f, dy_ = av.y > 0.0f ? g.y : 0.0f;
                const float dz_ = av.z > 0.0f ? g.z : 0.0f, dw_ = av.w > 0.0f ? g.w : 0.0f;
                sdy.x += dx_; sdy.y += dy_; sdy.z += dz_; sdy.w += dw_;
                sdx.x = fmaf(dx_, (av.x - bt.x) * rg.x, sdx.x); sdx.y = fmaf(dy_, (av.y - bt.y) * rg.y, sdx.y);
                sdx.z = fmaf(dz_, (av.z - bt.z) * rg.z, sdx.z); sdx.w = fmaf(dw_, (av.w - bt.w) * rg.w, sdx.w);
            }
        } else {
            const float4 s = *reinterpret_cast<const float4*>(scale + cc * 4), t = *reinterpret_cast<const float4*>(shift + cc * 4);
            const float4 mu = *reinterpret_cast<const float4*>(mean + cc * 4), is = *reinterpret_cast<const float4*>(invstd + cc * 4);
            for (int i = pslot; i < pix_per_block; i += pl) {
                const size_t p = p0 + i;
                if (p >= npix) break;
                const int px = p % Wp, py = (p / Wp) % Hp, b = p / ((size_t)Wp * Hp);
                const float4 g = *reinterpret_cast<const float4*>(da + p * C + cc * 4);
                float4 zz[4], yy[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    zz[q] = *reinterpret_cast<const float4*>(z + (((size_t)b * H + 2 * py + (q >> 1)) * W + 2 * px + (q & 1)) * C + cc * 4);
                    yy[q] = make_float4(fmaf(zz[q].x, s.x, t.x), fmaf(zz[q].y, s.y, t.y), fmaf(zz[q].z, s.z, t.z), fmaf(zz[q].w, s.w, t.w));
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float dx_ = route1(yy[0].x, yy[1].x, yy[2].x, yy[3].x, q, g.x);
                    const float dy_ = route1(yy[0].y, yy[1].y, yy[2].y, yy[3].y, q, g.y);
                    const float dz_ = route1(yy[0].z, yy[1].z, yy[2].z, yy[3].z, q, g.z);
                    const float dw_ = route1(yy[0].w, yy[1].w, yy[2].w, yy[3].w, q, g.w);
                    sdy.x += dx_; sdy.y += dy_; sdy.z += dz_; sdy.w += dw_;
                    sdx.x = fmaf(dx_, (zz[q].x - mu.x) * is.x, sdx.x); sdx.y = fmaf(dy_, (zz[q].y - mu.y) * is.y, sdx.y);
                    sdx.z = fmaf(dz_, (zz[q].z - mu.z) * is.z, sdx.z); sdx.w = fmaf(dw_, (zz[q].w - mu.w) * is.w, sdx.w);
                }
            }
        }
        const float vs[4] = {sdy.x, sdy.y, sdy.z, sdy.w}, vq[4] = {sdx.x, sdx.y, sdx.z, sdx.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __syncthreads();
            rs[threadIdx.x] = vs[e]; rq[threadIdx.x] = vq[e];
            __syncthreads();
            if (pslot == 0) {
                float acc = 0.0f, q2 = 0.0f;
                for (int k = 0; k < pl; ++k) { acc += rs[k * lanes_c + c4]; q2 += rq[k * lanes_c + c4]; }
                part[(size_t)blockIdx.x * C + cc * 4 + e] = make_float2(acc, q2);
            }
        }
    }
}

// GRU-layout activations [B * Wp][C * Hp] with feature = c * Hp + py and C * Hp = 1024, Hp = 8: thread = one float4 of a
// row (four py of ONE channel), block = rows_per_block rows; lanes 2c, 2c + 1 hold the two halves of channel c.
static __global__ __launch_bounds__(256) void bn_bwd_reduce_pooled_gru_kernel(
    const float* __restrict__ a, const float* __restrict__ da, const float* __restrict__ z, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, float2* __restrict__ part, int rows, int H, int W, int Wp, int rows_per_block) {
    constexpr int C = 128, Hp = 8;
    const int tid = threadIdx.x, c = tid >> 1, py0 = (tid & 1) * 4;
    const float gm = gamma[c], bt = beta[c];
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float sdy = 0.0f, sdx = 0.0f;
    if (bn_gamma_ok(gm, bt)) {
        const float rg = 1.0f / gm;
        for (int r = r0; r < r1; ++r) {
            const float4 av = *reinterpret_cast<const float4*>(a + (size_t)r * 1024 + tid * 4);
            const float4 g = *reinterpret_cast<const float4*>(da + (size_t)r * 1024 + tid * 4);
            const float d0 = av.x > 0.0f ? g.x : 0.0f, d1 = av.y > 0.0f ? g.y : 0.0f;
            const float d2 = av.z > 0.0f ? g.z : 0.0f, d3 = av.w > 0.0f ? g.w : 0.0f;
            sdy += (d0 + d1) + (d2 + d3);
            sdx = fmaf(d0, (av.x - bt) * rg, sdx); sdx = fmaf(d1, (av.y - bt) * rg, sdx);
            sdx = fmaf(d2, (av.z - bt) * rg, sdx); sdx = fmaf(d3, (av.w - bt) * rg, sdx);
        }
    } else {
        const float s = scale[c], t = shift[c], mu = mean[c], is = invstd[c];
        for (int r = r0; r < r1; ++r) {
            const int b = r / Wp, px = r - b * Wp;
            const float4 g4 = *reinterpret_cast<const float4*>(da + (size_t)r * 1024 + tid * 4);
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int py = py0 + e;
                float zz[4], yy[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    zz[q] = z[(((size_t)b * H + 2 * py + (q >> 1)) * W + 2 * px + (q & 1)) * C + c];
                    yy[q] = fmaf(zz[q], s, t);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float d_ = route1(yy[0], yy[1], yy[2], yy[3], q, gv[e]);
                    sdy += d_;
                    sdx = fmaf(d_, (zz[q] - mu) * is, sdx);
                }
            }
        }
    }
    static_assert(C * Hp == 1024, "one 1024-float row per 256 threads");
    sdy += __shfl_xor(sdy, 1);
    sdx += __shfl_xor(sdx, 1);
    if ((tid & 1) == 0) part[(size_t)blockIdx.x * C + c] = make_float2(sdy, sdx);
}

// sums the (sum dy, sum dy*xhat) partials -> dbeta, dgamma and the two per-channel means used by dz
static __global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float2* __restrict__ part, int nblk, int C, double count,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ mdy, float* __restrict__ mdyx, float unscale) {
    __shared__ double rs[256], rq[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    double s = 0.0, q = 0.0;
#pragma unroll 4
    for (int i = tid; i < nblk; i += 256) { const float2 v = part[(size_t)i * C + c]; s += v.x; q += v.y; }
    rs[tid] = s; rq[tid] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { rs[tid] += rs[tid + o]; rq[tid] += rq[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        dbeta[c] = (float)rs[0] * unscale;              // parameter gradients leave the loss scale; the means below feed dz and keep it
        dgamma[c] = (float)rq[0] * unscale;
        mdy[c] = (float)(rs[0] / count);
        mdyx[c] = (float)(rq[0] / count);
    }
}

// first maximum of the 2x2 window (PyTorch's max-pool tie rule: row-major scan, strict >), -1 if ReLU blocks it
__device__ __forceinline__ int route_arg(float y00, float y01, float y10, float y11) {
    float best = y00; int arg = 0;
    if (y01 > best) { best = y01; arg = 1; }
    if (y10 > best) { best = y10; arg = 2; }
    if (y11 > best) { best = y11; arg = 3; }
    return best > 0.0f ? arg : -1;
}

// thread = one 2x2 window x 4 channels: the four z values are read once (a thread per PIXEL re-read its three window
// neighbours for the routing: 5 loads per output against 1.25 here), the pooled gradient once per window
template <bool GRU_IN>
__global__ __launch_bounds__(256) void bn_bwd_dz_kernel(const float* __restrict__ z, const float* __restrict__ da,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ mdy, const float* __restrict__ mdyx,
                                                         float* __restrict__ dz, int B, int H, int W, int C, int Hp, int Wp) {
    const int c4n = C / 4, Hc = (H + 1) / 2, Wc = (W + 1) / 2;           // cells cover an odd last row / column too
    const size_t total = (size_t)B * Hc * Wc * c4n;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        int cc, cx, cy, b;
        if (GRU_IN) {
            // the pooled gradient is [b][px][c][py] here: with the cell ROW as the fastest thread index eight lanes read one 32-byte
            // row of it (channel-fastest threads picked one float out of every row: 373 MB of traffic for 236 MB of work)
            cy = idx % Hc;
            size_t rest = idx / Hc;
            cc = rest % c4n; rest /= c4n;
            cx = rest % Wc;
            b = rest / Wc;
        } else {
            cc = idx % c4n;
            size_t rest = idx / c4n;
            cx = rest % Wc; rest /= Wc;
            cy = rest % Hc;
            b = rest / Hc;
        }
        const float4 s = *reinterpret_cast<const float4*>(scale + cc * 4), t = *reinterpret_cast<const float4*>(shift + cc * 4);
        const float4 mu = *reinterpret_cast<const float4*>(mean + cc * 4), is = *reinterpret_cast<const float4*>(invstd + cc * 4);
        const float4 m1 = *reinterpret_cast<const float4*>(mdy + cc * 4), m2 = *reinterpret_cast<const float4*>(mdyx + cc * 4);
        float4 zq[4];
        bool ok[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int y = 2 * cy + (q >> 1), x = 2 * cx + (q & 1);
            ok[q] = y < H && x < W;
            zq[q] = ok[q] ? *reinterpret_cast<const float4*>(z + (((size_t)b * H + y) * W + x) * C + cc * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        int ax = -1, ay = -1, az = -1, aw = -1;
        if (cy < Hp && cx < Wp) {                                        // pooled window (all four pixels exist)
            g = load_da4<GRU_IN>(da, b, cy, cx, cc, Hp, Wp, C);
            ax = route_arg(fmaf(zq[0].x, s.x, t.x), fmaf(zq[1].x, s.x, t.x), fmaf(zq[2].x, s.x, t.x), fmaf(zq[3].x, s.x, t.x));
            ay = route_arg(fmaf(zq[0].y, s.y, t.y), fmaf(zq[1].y, s.y, t.y), fmaf(zq[2].y, s.y, t.y), fmaf(zq[3].y, s.y, t.y));
            az = route_arg(fmaf(zq[0].z, s.z, t.z), fmaf(zq[1].z, s.z, t.z), fmaf(zq[2].z, s.z, t.z), fmaf(zq[3].z, s.z, t.z));
            aw = route_arg(fmaf(zq[0].w, s.w, t.w), fmaf(zq[1].w, s.w, t.w), fmaf(zq[2].w, s.w, t.w), fmaf(zq[3].w, s.w, t.w));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!ok[q]) continue;
            const int y = 2 * cy + (q >> 1), x = 2 * cx + (q & 1);
            float4 o;
            o.x = s.x * ((ax == q ? g.x : 0.0f) - m1.x - (zq[q].x - mu.x) * is.x * m2.x);
            o.y = s.y * ((ay == q ? g.y : 0.0f) - m1.y - (zq[q].y - mu.y) * is.y * m2.y);
            o.z = s.z * ((az == q ? g.z : 0.0f) - m1.z - (zq[q].z - mu.z) * is.z * m2.z);
            o.w = s.w * ((aw == q ? g.w : 0.0f) - m1.w - (zq[q].w - mu.w) * is.w * m2.w);
            *reinterpret_cast<float4*>(dz + (((size_t)b * H + y) * W + x) * C + cc * 4) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------
// conv1 block backward (conv1 output is recomputed from the features, never stored).
//   pass 1: partial (sum dy, sum dy*xhat) per channel      -> bn_bwd_finalize_kernel
//   pass 2: dz = gamma*invstd*(dy - mean dy - xhat*mean(dy xhat)), dW1[c][tap] partials per block
// Same tiling as conv1_bn_relu_pool_kernel: block = 4 x 32 pooled pixels, lane&31 = channel.
// ------------------------------------------------------------------------------------------
// MODE 0: pass 1;  MODE 1: pass 2 (needs mdy / mdyx);  MODE 2: both in ONE pass -- (sum dy, sum dy*xhat, A[9]) with
// A[tap] = sum dy * x_tap, the only data-dependent part of the weight gradient: the mean terms of dz are closed forms in the
// input moments of conv1_moments_kernel (conv1_bwd_finalize_kernel).
template <int MODE>
__global__ __launch_bounds__(256) void conv1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ da, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, const float* __restrict__ mdy,
                                                         const float* __restrict__ mdyx, float* __restrict__ part,
                                                         int H, int W, int Hp, int Wp) {
    constexpr bool WGRAD = MODE == 1;
    constexpr int NV = MODE == 0 ? 2 : (MODE == 1 ? 9 : 11);
    __shared__ float tile[C1_TR * C1_TC];
    __shared__ float red[8 * 32 * NV];
    const int b = blockIdx.z, py0 = blockIdx.y * C1_PROWS, px0 = blockIdx.x * C1_PCOLS;
    const int tid = threadIdx.x, c = tid & 31, slot = tid >> 5;
    const float* xb = x + (size_t)b * H * W;
    for (int i = tid; i < C1_TR * C1_TC; i += 256) {
        const int ty = i / C1_TC, tx = i - ty * C1_TC;
        const int gy = 2 * py0 - 1 + ty, gx = 2 * px0 - 1 + tx;
        tile[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? xb[(size_t)gy * W + gx] : 0.0f;
    }
    float wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wk[i] = w[c * 9 + i];
    const float s = scale[c], t = shift[c], mu = mean[c], is = invstd[c];
    const float m1 = WGRAD ? mdy[c] : 0.0f, m2 = WGRAD ? mdyx[c] : 0.0f;
    float accv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) accv[i] = 0.0f;
    __syncthreads();
    for (int i = 0; i < (C1_PROWS * C1_PCOLS) / 8; ++i) {
        const int pp = slot + 8 * i, pyl = pp / C1_PCOLS, pxl = pp % C1_PCOLS;
        const int py = py0 + pyl, px = px0 + pxl;
        if (2 * py >= H || 2 * px >= W) continue;
        float in[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) in[r][k] = tile[(2 * pyl + r) * C1_TC + 2 * pxl + k];
        float a[4], yv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = 0.0f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) v = fmaf(in[(q >> 1) + ky][(q & 1) + kx], wk[ky * 3 + kx], v);
            a[q] = v;
            yv[q] = fmaf(v, s, t);
        }
        const bool pooled = (py < Hp && px < Wp);
        const float g = pooled ? da[(((size_t)b * Hp + py) * Wp + px) * 32 + c] : 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gy = 2 * py + (q >> 1), gx = 2 * px + (q & 1);
            if (gy >= H || gx >= W) continue;
            const float dyq = pooled ? route1(yv[0], yv[1], yv[2], yv[3], q, g) : 0.0f;
            const float xh = (a[q] - mu) * is;
            if (MODE != 1) {
                accv[0] += dyq;
                accv[1] = fmaf(dyq, xh, accv[1]);
                if (MODE == 2) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            accv[2 + ky * 3 + kx] = fmaf(dyq, in[(q >> 1) + ky][(q & 1) + kx], accv[2 + ky * 3 + kx]);
                            // keeps the SLP vectoriser from pairing these into v_pk_fma_f32: the shifted windows are not
                            // register-pair aligned, and the pairing cost 160 v_mov per pixel (166 us instead of ~100)
                            asm volatile("" : "+v"(accv[2 + ky * 3 + kx]));
                        }
                }
            } else {
                const float dzq = s * (dyq - m1 - xh * m2);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        accv[ky * 3 + kx] = fmaf(dzq, in[(q >> 1) + ky][(q & 1) + kx], accv[ky * 3 + kx]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) red[(slot * 32 + c) * NV + i] = accv[i];
    __syncthreads();
    const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    for (int o = tid; o < 32 * NV; o += 256) {
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += red[k * 32 * NV + o];
        part[blk * 32 * NV + o] = v;           // !WGRAD: float2 (sum dy, sum dy xhat) per channel
    }
}

// totals[c][11] = (sum dy, sum dy*xhat, A[9]) summed over the blocks;  M = input moments (forward).  Thread (c, tap):
//   dW[c][tap] = s_c (A[tap] - m1 S[tap] - m2 invstd_c (sum_t w_c[t] R[t][tap] - mean_c S[tap])),   m1 = sum dy / N, m2 = sum dy*xhat / N
// (dz = s (dy - m1 - xhat m2), xhat = (z - mean) invstd, z = sum_t w[t] x_t), evaluated in double; thread (c, 0) also writes
// dbeta = sum dy and dgamma = sum dy*xhat.
static __global__ void conv1_bwd_finalize_kernel(const float* __restrict__ totals, const double* __restrict__ M, const float* __restrict__ w,
                                                 const float* __restrict__ scale, const float* __restrict__ mean,
                                                 const float* __restrict__ invstd, double count, float* __restrict__ dgamma,
                                                 float* __restrict__ dbeta, float* __restrict__ dw, float unscale) {
    const int idx = threadIdx.x;
    if (idx >= 288) return;
    const int c = idx / 9, tap = idx % 9;
    const double sdy = totals[c * 11], sdx = totals[c * 11 + 1], A = totals[c * 11 + 2 + tap];
    const double m1 = sdy / count, m2 = sdx / count;
    double zx = 0.0;                                             // sum z * x_tap
    for (int t = 0; t < 9; ++t) zx += (double)w[c * 9 + t] * M[t <= tap ? c1_r_index(t, tap) : c1_r_index(tap, t)];
    const double xhx = (double)invstd[c] * (zx - (double)mean[c] * M[tap]);
    dw[c * 9 + tap] = (float)((double)scale[c] * (A - m1 * M[tap] - m2 * xhx)) * unscale;
    if (tap == 0) { dbeta[c] = (float)sdy * unscale; dgamma[c] = (float)sdx * unscale; }
}

// dW[co][ci][tap] = sum_blk slab[blk][tap][co][ci], in two ordered (deterministic) passes: WGR_PARTS partial sums over
// interleaved-free contiguous slab ranges (grid.y), then their sum + the transpose to the torch layout
constexpr int WGR_PARTS = 16;
static __global__ __launch_bounds__(256) void wgrad_reduce_partial_kernel(const float* __restrict__ slab, int nblk, int total4,
                                                                     float* __restrict__ part) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // float4 index into one slab
    if (idx >= total4) return;
    const int per = (nblk + WGR_PARTS - 1) / WGR_PARTS, k0 = blockIdx.y * per, k1 = min(nblk, k0 + per);
    const float4* s4 = reinterpret_cast<const float4*>(slab);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = k0;
    for (; k + 4 <= k1; k += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = s4[(size_t)(k + u) * total4 + idx];
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; k < k1; ++k) {
        const float4 v = s4[(size_t)k * total4 + idx];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4*>(part)[(size_t)blockIdx.y * total4 + idx] = acc;
}
static __global__ void wgrad_reduce_kernel(const float* __restrict__ part, int nparts, int cin, int cout, float* __restrict__ dw, float unscale) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // (tap, co, ci), ci fastest
    const int total = 9 * cout * cin;
    if (idx >= total) return;
    const int ci = idx % cin, co = (idx / cin) % cout, tap = idx / (cin * cout);
    float s = 0.0f;
#pragma unroll 8
    for (int k = 0; k < nparts; ++k) s += part[(size_t)k * total + idx];
    dw[((size_t)co * cin + ci) * 9 + tap] = s * unscale;
}

// ------------------------------------------------------------------------------------------
// multi-tensor Adam, coupled L2 weight decay (torch.optim.Adam, train.py:246-250): one launch
// updates every parameter tensor.  Tensors are described by value in the kernel argument.
// ------------------------------------------------------------------------------------------
#define SIR_ADAM_MAX_TENSORS 32
#define SIR_ADAM_CHUNK 4096

struct AdamTensors {
    float* p[SIR_ADAM_MAX_TENSORS];
    const float* g[SIR_ADAM_MAX_TENSORS];
    float* m[SIR_ADAM_MAX_TENSORS];
    float* v[SIR_ADAM_MAX_TENSORS];
    long long n[SIR_ADAM_MAX_TENSORS];
    int first_block[SIR_ADAM_MAX_TENSORS + 1];
    int count;
};

static __global__ __launch_bounds__(256) void adam_multi_kernel(AdamTensors ts, float lr, float beta1, float beta2, float eps,
                                                          float weight_decay, float bc1, float bc2_sqrt) {
    int ti = 0;
    while (ti + 1 < ts.count && (int)blockIdx.x >= ts.first_block[ti + 1]) ++ti;
    const long long base = (long long)((int)blockIdx.x - ts.first_block[ti]) * SIR_ADAM_CHUNK;
    float* __restrict__ p = ts.p[ti];
    const float* __restrict__ g = ts.g[ti];
    float* __restrict__ m = ts.m[ti];
    float* __restrict__ v = ts.v[ti];
    const long long n = ts.n[ti];
    const float step_size = lr / bc1;
#pragma unroll 4
    for (int k = 0; k < SIR_ADAM_CHUNK / 256; ++k) {
        const long long i = base + threadIdx.x + 256 * k;
        if (i >= n) break;
        float gi = g[i];
        const float pi = p[i];
        if (weight_decay != 0.0f) gi = fmaf(weight_decay, pi, gi);
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}
