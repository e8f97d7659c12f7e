// Standalone timing/consistency harness of the bf16x6 GEMM kernels (developer tool, not part of the
// library): random fp32 operands -> bf16x3 planes -> every kernel generation, HIP-event timing,
// max difference against generation 1 and against an fp64 host dot product on sampled entries.
//   build: make -C speech-intent-recognizer_amd/csrc tools     run (GPU box): lib/bench_gemm [M] [K]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../speech-intent-recognizer_amd/csrc/bf16x6_kernels.h"
#include "legacy_kernels.h"

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

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 6400, K = argc > 2 ? atoi(argv[2]) : 1024, N = 768, reps = 20;
    std::vector<float> hA((size_t)M * K), hB((size_t)2 * N * K), hbias(2 * N);
    srand(1);
    for (auto& v : hA) v = (rand() / (float)RAND_MAX - 0.5f) * 2.0f;
    for (auto& v : hB) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    for (auto& v : hbias) v = rand() / (float)RAND_MAX;
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
    return 0;
}
