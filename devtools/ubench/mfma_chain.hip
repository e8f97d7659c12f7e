// How far apart must two v_mfma_f32_32x32x16_f16 on the SAME accumulator be?  One wave (or two, or four) per SIMD issues MFMAs round-robin over NACC
// independent accumulators; ticks (s_memtime) per MFMA and wave.  If the time per MFMA falls as NACC grows, the shorter chains were bound by the
// accumulator dependency (the result of MFMA k is an operand of MFMA k + NACC), not by the matrix pipe.
//   build: hipcc -O3 --offload-arch=gfx950 mfma_chain.hip -o mfma_chain ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int WPS>
__global__ __launch_bounds__(256 * WPS) void chain(float* out, int iters, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    f16x8 a, b;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(lane * 0.01f + k); b[k] = (_Float16)(0.25f * k); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 24 / NACC; ++r)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sink = 0.0f;
#pragma unroll
    for (int j = 0; j < NACC; ++j) sink += acc[j][0] + acc[j][7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, int WPS>
static void one(float* out, unsigned long long* cyc) {
    const int iters = 2000;
    unsigned long long h = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemset(cyc, 0, 8);
        hipLaunchKernelGGL((chain<NACC, WPS>), dim3(256), dim3(256 * WPS), 0, 0, out, iters, cyc);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    // the same launch under HIP events: what a tick is in wall time, and the chip's achieved matrix rate
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((chain<NACC, WPS>), dim3(256), dim3(256 * WPS), 0, 0, out, iters, cyc);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double tflops = 256.0 * 4 * WPS * iters * 24.0 * 32768.0 / (ms * 1e-3) * 1e-12;
    printf("   %d wave(s) per SIMD: %.1f ticks per MFMA and wave (SIMD: one per %.1f); %.3f ms -> %.0f MHz ticks, %.0f TFLOP/s\n", WPS, (double)h / (iters * 24.0),
           (double)h / (iters * 24.0) / WPS, ms, (double)h / (ms * 1e-3) * 1e-6, tflops);
}
template <int NACC>
static void run(float* out, unsigned long long* cyc) {
    printf("%2d accumulators:\n", NACC);
    one<NACC, 1>(out, cyc);
    one<NACC, 2>(out, cyc);
    if (NACC <= 6) one<NACC <= 6 ? NACC : 1, 4>(out, cyc);
    printf("\n");
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 8);
    run<1>(out, cyc); run<2>(out, cyc); run<4>(out, cyc); run<6>(out, cyc); run<8>(out, cyc); run<12>(out, cyc);
    if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
    return 0;
}
