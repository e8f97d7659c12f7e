// The two-way fp16 split of an fp32 value ("f16x3" contractions, f16x3_kernels.h):
//     hi = fp16(x)  (round to nearest, 11 significant bits)         lo = fp16((x - hi) * 2^11)
// x - hi is exact in fp32; the residual is scaled by 2^11 so that it sits in fp16's normal range wherever hi does.
// hi + lo * 2^-11 carries 22-23 significant bits of x for 2^-14 <= |x| <= 65504; smaller values keep an ABSOLUTE error of
// 2^-36.  Values beyond fp16's range are clamped to +-65504 first (BatchNorm / ReLU outputs, GRU states and weights are
// nowhere near; an overflow to infinity would turn 0 * inf products into NaNs).
#pragma once

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sir_f16x2 __attribute__((ext_vector_type(2)));

constexpr float H3_LO_SCALE = 2048.0f;                      // 2^11
constexpr float H3_LO_INV = 1.0f / 2048.0f;

// two floats -> packed hi halves, packed scaled-residual halves (a in the low 16 bits)
__device__ __forceinline__ void split2h_pair(float a, float b, unsigned& h, unsigned& l) {
    sir_f32x2 v = {__builtin_fminf(__builtin_fmaxf(a, -65504.0f), 65504.0f), __builtin_fminf(__builtin_fmaxf(b, -65504.0f), 65504.0f)};
    const sir_f16x2 hi = __builtin_convertvector(v, sir_f16x2);
    v -= __builtin_convertvector(hi, sir_f32x2);            // exact
    v *= H3_LO_SCALE;
    const sir_f16x2 lo = __builtin_convertvector(v, sir_f16x2);
    h = __builtin_bit_cast(unsigned, hi);
    l = __builtin_bit_cast(unsigned, lo);
}
// the same without the clamps, for waves that run with MODE.FP16_OVFL = 1 (an overflowing conversion then saturates at +-65504 by itself)
__device__ __forceinline__ void split2h_pair_ovfl(float a, float b, unsigned& h, unsigned& l) {
    // (lo' as one v_fma_mixlo / mixhi_f16 per value -- 4 instead of 6 instructions per pair, the same rounding -- was measured: no faster, devtools/gpu_r4ae.sh)
    sir_f32x2 v = {a, b};
    const sir_f16x2 hi = __builtin_convertvector(v, sir_f16x2);
    v -= __builtin_convertvector(hi, sir_f32x2);
    v *= H3_LO_SCALE;
    const sir_f16x2 lo = __builtin_convertvector(v, sir_f16x2);
    h = __builtin_bit_cast(unsigned, hi);
    l = __builtin_bit_cast(unsigned, lo);
}
__device__ __forceinline__ void split2h_quad_ovfl(const float4& v, uint2& h, uint2& l) {
    split2h_pair_ovfl(v.x, v.y, h.x, l.x);
    split2h_pair_ovfl(v.z, v.w, h.y, l.y);
}
// MODE.FP16_OVFL = 1 for the calling wave (hwreg id 1 = MODE, bit 23): fp16 results that overflow saturate at +-65504 instead of becoming infinity
__device__ __forceinline__ void sir_fp16_ovfl_on() { __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1); }
__device__ __forceinline__ void split2h_quad(const float4& v, uint2& h, uint2& l) {
    split2h_pair(v.x, v.y, h.x, l.x);
    split2h_pair(v.z, v.w, h.y, l.y);
}

// in [rows][K] fp32 (row stride ld_in) -> planes [2][rows][K] fp16; one thread = 8 consecutive k
__device__ __forceinline__ void split2h_rows(const float* __restrict__ in, int ld_in, unsigned short* __restrict__ out, size_t rows, int K,
                                             size_t gidx, size_t nthreads) {
    const int k8n = K / 8;
    const size_t total = rows * k8n, plane = rows * (size_t)K;
    for (size_t idx = gidx; idx < total; idx += nthreads) {
        const size_t row = idx / k8n;
        const int k8 = idx % k8n;
        const float4 v0 = *reinterpret_cast<const float4*>(in + row * ld_in + k8 * 8);
        const float4 v1 = *reinterpret_cast<const float4*>(in + row * ld_in + k8 * 8 + 4);
        uint2 h0, l0, h1, l1;
        split2h_quad(v0, h0, l0);
        split2h_quad(v1, h1, l1);
        const size_t o = row * K + (size_t)k8 * 8;
        *reinterpret_cast<uint4*>(out + o) = make_uint4(h0.x, h0.y, h1.x, h1.y);
        *reinterpret_cast<uint4*>(out + plane + o) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}
static __global__ __launch_bounds__(256) void split2h_kernel(const float* __restrict__ in, int ld_in, unsigned short* __restrict__ out,
                                                              size_t rows, int K) {
    split2h_rows(in, ld_in, out, rows, K, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
}
