// Standalone timing/consistency harness of the bf16x6 GEMM kernels (developer tool, not part of the
// library): random fp32 operands -> bf16x3 planes -> every kernel generation, HIP-event timing,
// max difference against generation 1 and against an fp64 host dot product on sampled entries.
//   build: make -C speech-intent-recognizer_amd/csrc tools     run (GPU box): lib/bench_gemm [M] [K] [A.f32 B.f32 [bias.f32]]
// (A.f32 = [M][K], B.f32 = [2 * 768][K] raw float32 files: the REAL projection operands dumped by devtools/dump_gemm_operands.py)
// Round 4: the two-way fp16 split ("f16x3": 3 products, 2 planes; f16x3_kernels.h) beside bf16x6, with a float64 product of the
// same fp32 operands (computed on the GPU) as the error reference for both.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../speech-intent-recognizer_amd/csrc/bf16x6_kernels.h"
#include "../../speech-intent-recognizer_amd/csrc/f16x3_kernels.h"
#include "legacy_kernels.h"
#include "gemm_f16x3_w4_kernel.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

// matrix-pipe ceiling: 8 waves per CU-resident workgroup, 30 independent-accumulator bf16 MFMAs per iteration, no memory
template <int WAVES>
static __global__ __launch_bounds__(WAVES * 64) void mfma_peak_kernel(float* out, int iters, unsigned seed) {
    f32x16 acc[5];
    for (int i = 0; i < 5; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    union { bf16x8 v; unsigned u[4]; } a[3], b[3];
    unsigned x = seed + threadIdx.x * 2654435761u + blockIdx.x;
    for (int p = 0; p < 3; ++p) for (int j = 0; j < 4; ++j) {
        x = x * 1664525u + 1013904223u; a[p].u[j] = (x & 0x807F807Fu) | 0x3F003F00u;
        x = x * 1664525u + 1013904223u; b[p].u[j] = (x & 0x807F807Fu) | 0x3F003F00u;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t % 3].v, b[(t + mt) % 3].v, acc[mt], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 5; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int WAVES>
static void run_peak(hipStream_t st, float* out) {
    const int iters = 2000, blocks = 256 * (8 / WAVES) * 2;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_peak_kernel<WAVES>, dim3(blocks), dim3(WAVES * 64), 0, st, out, 100, 1u);
    hipEventRecord(e0, st);
    hipLaunchKernelGGL(mfma_peak_kernel<WAVES>, dim3(blocks), dim3(WAVES * 64), 0, st, out, iters, 1u);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * WAVES * iters * 30 * 2.0 * 32 * 32 * 16;
    printf("mfma peak probe (%d waves/WG, %d WGs): %.1f us, %.0f TF bf16 executed = %.1f TF bf16x6-algorithmic\n", WAVES, blocks, ms * 1e3,
           flops / (ms * 1e-3) * 1e-12, flops / (ms * 1e-3) * 1e-12 / 6);
}

template <int KNOCK>
static float time_v3(hipStream_t st, const unsigned short* Ap, const unsigned short* B0, const unsigned short* B1, const float* bias,
                     float* C, int M, int N, int K, int reps) {
    hipFuncSetAttribute((const void*)gemm_nt_bf16x6_v3_kernel<KNOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES);
    const int nwg = ((M + G3_BM - 1) / G3_BM) * 2 * (N / G3_BN);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL(gemm_nt_bf16x6_v3_kernel<KNOCK>, dim3(nwg), dim3(512), G3_LDS_BYTES, st, Ap, B0, B1, bias, bias + N, C, 2 * N, M, N, K);
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL(gemm_nt_bf16x6_v3_kernel<KNOCK>, dim3(nwg), dim3(512), G3_LDS_BYTES, st, Ap, B0, B1, bias, bias + N, C, 2 * N, M, N, K);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f / reps;
}


// float64 product of the fp32 operands: the error reference (naive, one output per thread)
static __global__ void ref_gemm_f64_kernel(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
                                           double* __restrict__ C, int M, int N2, int K) {
    const int n = blockIdx.x * 16 + (threadIdx.x & 15), mm = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (mm >= M || n >= N2) return;
    double acc = 0.0;
    const float4* a = reinterpret_cast<const float4*>(A + (size_t)mm * K);
    const float4* b = reinterpret_cast<const float4*>(B + (size_t)n * K);
    for (int k = 0; k < K / 4; ++k) {
        const float4 x = a[k], y = b[k];
        acc += (double)x.x * y.x + (double)x.y * y.y + (double)x.z * y.z + (double)x.w * y.w;
    }
    C[(size_t)mm * N2 + n] = acc + (double)bias[n];
}
// the same product as one fp32 fmaf chain in k order (what a scalar fp32 loop computes): context for the error table
static __global__ void ref_gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
                                           float* __restrict__ C, int M, int N2, int K) {
    const int n = blockIdx.x * 16 + (threadIdx.x & 15), mm = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (mm >= M || n >= N2) return;
    float acc = 0.0f;
    for (int k = 0; k < K; ++k) acc = fmaf(A[(size_t)mm * K + k], B[(size_t)n * K + k], acc);
    C[(size_t)mm * N2 + n] = acc + bias[n];
}
// fp16 subnormal probe: does v_mfma_f32_32x32x16_f16 keep subnormal inputs?  a = 2^-20 (subnormal), b = 1 -> 16 * 2^-20 expected
static __global__ void f16_denorm_probe_kernel(float* out) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)9.5367431640625e-07f; b[i] = (_Float16)1.0f; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.0f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}

template <int NST, int KNOCK>
static float time_h3(hipStream_t st, const unsigned short* Ap, const unsigned short* B0, const unsigned short* B1, const float* bias,
                     float* C, int M, int N, int K, int reps) {
    hipFuncSetAttribute((const void*)gemm_nt_f16x3_kernel<NST, KNOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, h3_lds_bytes(NST));
    const int nwg = ((M + H3_BM - 1) / H3_BM) * 2 * (N / H3_BN);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((gemm_nt_f16x3_kernel<NST, KNOCK>), dim3(nwg), dim3(512), h3_lds_bytes(NST), st, Ap, B0, B1, bias, bias + N, C, 2 * N, M, N, K);
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((gemm_nt_f16x3_kernel<NST, KNOCK>), dim3(nwg), dim3(512), h3_lds_bytes(NST), st, Ap, B0, B1, bias, bias + N, C, 2 * N, M, N, K);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f / reps;
}

template <int KNOCK, bool XT = true>
static float time_h3w4(hipStream_t st, const unsigned short* Ap, const unsigned short* B0, const unsigned short* B1, const float* bias,
                       float* C, int M, int N, int K, int reps) {
    hipFuncSetAttribute((const void*)gemm_nt_f16x3_w4_kernel<KNOCK, XT>, hipFuncAttributeMaxDynamicSharedMemorySize, h3_lds_bytes(3));
    const int nwg = ((M + H3_BM - 1) / H3_BM) * 2 * (N / H3_BN);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((gemm_nt_f16x3_w4_kernel<KNOCK, XT>), dim3(nwg), dim3(256), h3_lds_bytes(3), st, Ap, B0, B1, bias, bias + N, C, 2 * N, M, N, K);
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((gemm_nt_f16x3_w4_kernel<KNOCK, XT>), dim3(nwg), dim3(256), h3_lds_bytes(3), st, Ap, B0, B1, bias, bias + N, C, 2 * N, M, N, K);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f / reps;
}

static bool read_f32(const char* path, std::vector<float>& v) {
    FILE* f = fopen(path, "rb");
    if (!f) { printf("cannot open %s\n", path); return false; }
    const size_t n = fread(v.data(), 4, v.size(), f);
    fclose(f);
    if (n != v.size()) { printf("%s: %zu floats, expected %zu\n", path, n, v.size()); return false; }
    return true;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 6400, K = argc > 2 ? atoi(argv[2]) : 1024, N = 768, reps = 20;
    std::vector<float> hA((size_t)M * K), hB((size_t)2 * N * K), hbias(2 * N);
    srand(1);
    for (auto& v : hA) v = (rand() / (float)RAND_MAX - 0.5f) * 2.0f;
    for (auto& v : hB) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    for (auto& v : hbias) v = rand() / (float)RAND_MAX;
    const bool real_ops = argc > 4;
    if (real_ops) {
        if (!read_f32(argv[3], hA) || !read_f32(argv[4], hB)) return 1;
        if (argc > 5 && !read_f32(argv[5], hbias)) return 1;
        printf("operands from %s, %s\n", argv[3], argv[4]);
    }
    float *dA, *dB, *dbias, *C1, *C2;
    unsigned short *pA, *pB;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dbias, hbias.size() * 4));
    CK(hipMalloc(&C1, (size_t)M * 2 * N * 4)); CK(hipMalloc(&C2, (size_t)M * 2 * N * 4));
    CK(hipMalloc(&pA, hA.size() * 6)); CK(hipMalloc(&pB, hB.size() * 6));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, hbias.data(), hbias.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipLaunchKernelGGL(split3_kernel, dim3(2048), dim3(256), 0, st, (const float*)dA, K, pA, (size_t)M, K);
    for (int d = 0; d < 2; ++d)
        hipLaunchKernelGGL(split3_kernel, dim3(512), dim3(256), 0, st, (const float*)(dB + (size_t)d * N * K), K, pB + (size_t)d * 3 * N * K, (size_t)N, K);
    CK(hipStreamSynchronize(st));
    const unsigned short *B0 = pB, *B1 = pB + (size_t)3 * N * K;
    const double gf = 2.0 * M * K * 2 * N * 1e-9;
    run_peak<4>(st, C2);
    run_peak<8>(st, C2);
    // generation 1
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 g1((N + GB_N - 1) / GB_N, (M + GB_M - 1) / GB_M, 2);
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL(gemm_nt_bf16x6_kernel, g1, dim3(256), 0, st, (const unsigned short*)pA, B0, B1, (const float*)dbias, (const float*)(dbias + N), C1, 2 * N, M, N, K);
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL(gemm_nt_bf16x6_kernel, g1, dim3(256), 0, st, (const unsigned short*)pA, B0, B1, (const float*)dbias, (const float*)(dbias + N), C1, 2 * N, M, N, K);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("M=%d K=%d N=2x%d  %.2f GF\n", M, K, N, gf);
    printf("gen1                      %8.1f us  %7.1f TF\n", ms * 1000 / reps, gf / (ms / reps));
    std::vector<float> h1((size_t)M * 2 * N), h2((size_t)M * 2 * N);
    CK(hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost));
    double dref = 0;
    for (int s = 0; s < 2000; ++s) {
        const int r = rand() % M, c = rand() % (2 * N);
        double acc = hbias[c];
        for (int k = 0; k < K; ++k) acc += (double)hA[(size_t)r * K + k] * hB[(size_t)c * K + k];
        dref = fmax(dref, fabs(acc - h1[(size_t)r * 2 * N + c]));
    }
    printf("max |gen1 - fp64| over 2000 sampled entries = %.3e\n", dref);
    auto check = [&](const char* name, float t) {
        hipStreamSynchronize(st);
        hipMemcpy(h2.data(), C2, h2.size() * 4, hipMemcpyDeviceToHost);
        double d = 0;
        for (size_t i = 0; i < h1.size(); ++i) d = fmax(d, fabs((double)h1[i] - h2[i]));
        printf("%-25s %8.1f us  %7.1f TF   max |gen1 - this| = %.3e\n", name, t, gf * 1e3 / t, d);
        hipMemset(C2, 0, (size_t)M * 2 * N * 4);
    };
    CK(hipMemset(C2, 0, (size_t)M * 2 * N * 4));
    check("gen2 bulk DMA issue", time_v3<0>(st, pA, B0, B1, dbias, C2, M, N, K, reps));
    check("gen2 spread 1", time_v3<8>(st, pA, B0, B1, dbias, C2, M, N, K, reps));
    check("gen2 spread 2 (product)", time_v3<G3_DEFAULT>(st, pA, B0, B1, dbias, C2, M, N, K, reps));
    check("gen2 spread 2 + prefetch", time_v3<48>(st, pA, B0, B1, dbias, C2, M, N, K, reps));
    printf("timing-only knock-outs (results invalid):\n");
    printf("  no staging       %8.1f us\n", time_v3<1>(st, pA, B0, B1, dbias, C2, M, N, K, reps));
    printf("  no MFMA          %8.1f us\n", time_v3<4>(st, pA, B0, B1, dbias, C2, M, N, K, reps));
    // ---- round 4: fp16 two-way split against bf16x6, errors against a float64 product -------------------------------------
    {
        float* dp; CK(hipMalloc(&dp, 4));
        hipLaunchKernelGGL(f16_denorm_probe_kernel, dim3(1), dim3(64), 0, st, dp);
        float pv; CK(hipMemcpy(&pv, dp, 4, hipMemcpyDeviceToHost));
        printf("fp16 subnormal through v_mfma_f32_32x32x16_f16: 16 x 2^-20 x 1 = %.6e (expected %.6e)%s\n", pv, 16.0 * 9.5367431640625e-07,
               pv == 0.0f ? "  -- FLUSHED" : "");
        unsigned short *hA2, *hB2;
        CK(hipMalloc(&hA2, hA.size() * 4)); CK(hipMalloc(&hB2, hB.size() * 4));
        hipLaunchKernelGGL(split2h_kernel, dim3(2048), dim3(256), 0, st, (const float*)dA, K, hA2, (size_t)M, K);
        for (int d = 0; d < 2; ++d)
            hipLaunchKernelGGL(split2h_kernel, dim3(512), dim3(256), 0, st, (const float*)(dB + (size_t)d * N * K), K, hB2 + (size_t)d * 2 * N * K, (size_t)N, K);
        CK(hipStreamSynchronize(st));
        const unsigned short *H0 = hB2, *H1 = hB2 + (size_t)2 * N * K;
        double* Cref; float* Cf32;
        CK(hipMalloc(&Cref, (size_t)M * 2 * N * 8)); CK(hipMalloc(&Cf32, (size_t)M * 2 * N * 4));
        hipLaunchKernelGGL(ref_gemm_f64_kernel, dim3(2 * N / 16, (M + 15) / 16), dim3(256), 0, st, (const float*)dA, (const float*)dB, (const float*)dbias, Cref, M, 2 * N, K);
        hipLaunchKernelGGL(ref_gemm_f32_kernel, dim3(2 * N / 16, (M + 15) / 16), dim3(256), 0, st, (const float*)dA, (const float*)dB, (const float*)dbias, Cf32, M, 2 * N, K);
        CK(hipStreamSynchronize(st));
        std::vector<double> href((size_t)M * 2 * N);
        CK(hipMemcpy(href.data(), Cref, href.size() * 8, hipMemcpyDeviceToHost));
        double rms_ref = 0, max_ref = 0;
        for (double v : href) { rms_ref += v * v; max_ref = fmax(max_ref, fabs(v)); }
        rms_ref = sqrt(rms_ref / href.size());
        printf("\nerror table against the float64 product (%zu entries, rms |C| = %.4f, max |C| = %.4f)\n", href.size(), rms_ref, max_ref);
        printf("%-34s %9s %12s %12s %12s\n", "kernel", "us", "max |err|", "rms err", "rms / rms|C|");
        auto errs = [&](const char* name, const float* dC, float t) {
            hipStreamSynchronize(st);
            hipMemcpy(h2.data(), dC, h2.size() * 4, hipMemcpyDeviceToHost);
            double mx = 0, sq = 0;
            for (size_t i = 0; i < h2.size(); ++i) { const double e = (double)h2[i] - href[i]; mx = fmax(mx, fabs(e)); sq += e * e; }
            const double rms = sqrt(sq / h2.size());
            printf("%-34s %9.1f %12.3e %12.3e %12.3e\n", name, t, mx, rms, rms / rms_ref);
            return rms;
        };
        errs("fp32 fmaf chain (scalar loop)", Cf32, 0.0f);
        const float tb = time_v3<G3_DEFAULT>(st, pA, B0, B1, dbias, C2, M, N, K, reps);
        const double eb = errs("bf16x6 (product, 6 MFMA 3 planes)", C2, tb);
        CK(hipMemset(C2, 0, (size_t)M * 2 * N * 4));
        const float t2 = time_h3<2, 0>(st, hA2, H0, H1, dbias, C2, M, N, K, reps);
        errs("f16x3, 2 stages", C2, t2);
        CK(hipMemset(C2, 0, (size_t)M * 2 * N * 4));
        const float t3 = time_h3<3, 0>(st, hA2, H0, H1, dbias, C2, M, N, K, reps);
        const double eh = errs("f16x3, 3 stages", C2, t3);
        printf("f16x3 (3 stages) vs bf16x6: %.2fx faster, rms error %.2fx\n", tb / t3, eh / eb);
        // round 4 experiment: four fat waves (160 x 64 per wave, one 2^11-scaled accumulator per product, register double-buffered fragments)
        CK(hipMemset(C2, 0, (size_t)M * 2 * N * 4));
        const float t4 = time_h3w4<0>(st, hA2, H0, H1, dbias, C2, M, N, K, reps);
        const double e4 = errs("f16x3, four 160x64 waves", C2, t4);
        printf("four-wave form vs f16x3 (3 stages): %.2fx faster, rms error %.2fx;  knock-outs: no staging %.1f us, no MFMA %.1f us\n", t3 / t4, e4 / eh,
               time_h3w4<1>(st, hA2, H0, H1, dbias, C2, M, N, K, reps), time_h3w4<4>(st, hA2, H0, H1, dbias, C2, M, N, K, reps));
        CK(hipMemset(C2, 0, (size_t)M * 2 * N * 4));
        const float t5 = time_h3w4<0, false>(st, hA2, H0, H1, dbias, C2, M, N, K, reps);
        errs("f16x3, four waves, barrier at tile end", C2, t5);
        printf("  its knock-outs: no staging %.1f us, no MFMA %.1f us\n", time_h3w4<1, false>(st, hA2, H0, H1, dbias, C2, M, N, K, reps), time_h3w4<4, false>(st, hA2, H0, H1, dbias, C2, M, N, K, reps));
        printf("f16x3 timing-only knock-outs (results invalid):\n");
        printf("  3 stages, no staging   %8.1f us\n", time_h3<3, 1>(st, hA2, H0, H1, dbias, C2, M, N, K, reps));
        printf("  3 stages, no MFMA      %8.1f us\n", time_h3<3, 4>(st, hA2, H0, H1, dbias, C2, M, N, K, reps));
    }
    return 0;
}
