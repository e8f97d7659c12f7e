// VALU issue-rate probe: how many cycles does one wave64 v_fma_f32 cost a SIMD, at 1 / 2 / 4 waves per SIMD?
// build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MIX>
__global__ void probe(float* out, int iters, unsigned long long* cyc) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    const float m = 1.0001f, c = 0.5f;
    int s = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MIX == 0) a[i] = fmaf(a[i], m, c);                               // compiler may pack pairs into v_pk_fma_f32
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));   // scalar fma, one per accumulator
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4 * 8); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int mix = 0; mix < 2; ++mix)
    for (int threads : {64, 256, 512, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mix == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
            else hipLaunchKernelGGL(probe<1>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double fmas_per_wave = (double)iters * 64;
            const int waves_per_simd = threads >= 256 ? threads / 256 : 1;
            if (rep == 1)
                printf("mix=%d threads=%4d waves/SIMD=%d: %.3f ms, %llu cycles (memtime @100MHz?) -> %.2f shader-cycles per wave-fma per SIMD (assuming 2.4 GHz: %.2f)\n",
                       mix, threads, waves_per_simd, ms, c, 0.0, ms * 1e-3 * 2.4e9 / (fmas_per_wave * waves_per_simd));
        }
    }
    return 0;
}
