// NOTE on reading the output: waves of one kind on a SIMD are served oldest-first (mfma_chain.hip), so the stamps are the rate of the FIRST-served wave of
// each role; the printed "its SIMD: one per ..." quotients are not throughput.  s_memtime ticks are core cycles.
// Issue-sharing probe: ONE wave per SIMD runs plain fp32 VALU work (a Winograd producer's kind of instruction stream) while TWO waves per SIMD
// run v_mfma_f32_32x32x16_f16 in the consumers' pattern (12 MFMAs per round on four accumulators).  How much does each side lose to the other?
//   build: hipcc -O3 --offload-arch=gfx950 mfma_valu_mix.hip -o mfma_valu_mix ; run on the GPU box
// Workgroup = 12 waves (waves 0-3 -> VALU role, one per SIMD; waves 4-11 -> MFMA role, two per SIMD), one workgroup per CU (64 KB of LDS asked for
// 3 per CU would also fit: the grid is 256 workgroups, so each lands on its own CU).  mode bit 0: VALU waves run, bit 1: MFMA waves run,
// bit 2: VALU waves at s_setprio 3, bit 3: the VALU stream is v_pk_fma_f32 instead of v_fma_f32, bit 4: VALU waves also do one ds_read_b128 per 16 FMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int mode>
__global__ __launch_bounds__(768, 1) void probe(float* out, int iters_v, int iters_m, unsigned long long* cyc) {
    extern __shared__ float lds[];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float sink = 0.0f;
    if (wv < 4) {
        if (!(mode & 1)) return;
        if (mode & 4) __builtin_amdgcn_s_setprio(3);
        float a[8];
        f32x2 p[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = lane * 0.001f + i; p[i] = f32x2{a[i], a[i] + 1.0f}; }
        const float m = 1.0001f, c = 0.5f;
        const f32x2 m2 = {m, m}, c2 = {c, c};
        float4 l = {0, 0, 0, 0};
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters_v; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (mode & 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                }
                if ((mode & 16) && (r & 1)) {
                    float4 t = *reinterpret_cast<const float4*>(&lds[((it * 8 + r) & 63) * 256 + lane * 4]);
                    l.x += t.x;
                }
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 8; ++i) sink += a[i] + p[i].x + p[i].y;
        sink += l.x;
        if (lane == 0 && blockIdx.x == 0 && wv == 0) cyc[0] = t1 - t0;
    } else {
        if (!(mode & 2)) return;
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        f16x8 av[2], bv[3];
#pragma unroll
        for (int k = 0; k < 8; ++k) { av[0][k] = (_Float16)(lane * 0.01f + k); av[1][k] = (_Float16)(k * 0.5f); bv[0][k] = (_Float16)(0.25f * k); bv[1][k] = (_Float16)1.0f; bv[2][k] = (_Float16)(lane & 3); }
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters_m; ++it) {
            // the consumers' order: two cross terms, then the main one, each over the four accumulators
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[t == 0 ? 1 : 0], bv[t], acc[j], 0, 0, 0);
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int j = 0; j < 4; ++j) sink += acc[j][0] + acc[j][5];
        if (lane == 0 && blockIdx.x == 0 && wv == 4) cyc[1] = t1 - t0;
    }
    out[blockIdx.x * 768 + threadIdx.x] = sink;
}

// Second probe: 16 waves per workgroup, eight VALU and eight MFMA waves.  SPLIT = 0: roles by wave >> 3 (every SIMD gets two of each, the Winograd
// kernels' mix); SPLIT = 1: roles by SIMD -- waves with (wave & 3) < 2 are VALU waves (four each on SIMDs 0 and 1), the others MFMA waves (four each
// on SIMDs 2 and 3), assuming wave i of a workgroup lands on SIMD i & 3 (checked: the SIMD id of every wave is recorded from HW_ID).
template <int SPLIT>
__global__ __launch_bounds__(1024, 1) void probe2(float* out, int iters_v, int iters_m, unsigned long long* cyc, int* simd_of_wave) {
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool valu_role = SPLIT ? (wv & 3) < 2 : wv < 8;
    if (blockIdx.x == 0 && lane == 0) simd_of_wave[wv] = (__builtin_amdgcn_s_getreg((4 << 0) | (4 << 6) | (1 << 11)) & 3);   // HW_ID bits 5:4 = SIMD_ID
    float sink = 0.0f;
    if (valu_role) {
        float a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = lane * 0.001f + i;
        const float m = 1.0001f, c = 0.5f;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters_v; ++it) {
#pragma unroll
            for (int r = 0; r < 64; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[r & 7]) : "v"(m), "v"(c));
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 8; ++i) sink += a[i];
        if (lane == 0 && blockIdx.x == 0 && wv == 0) cyc[0] = t1 - t0;
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        f16x8 av[2], bv[3];
#pragma unroll
        for (int k = 0; k < 8; ++k) { av[0][k] = (_Float16)(lane * 0.01f + k); av[1][k] = (_Float16)(k * 0.5f); bv[0][k] = (_Float16)(0.25f * k); bv[1][k] = (_Float16)1.0f; bv[2][k] = (_Float16)(lane & 3); }
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters_m; ++it) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[t == 0 ? 1 : 0], bv[t], acc[j], 0, 0, 0);
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int j = 0; j < 4; ++j) sink += acc[j][0] + acc[j][5];
        if (lane == 0 && blockIdx.x == 0 && wv == (SPLIT ? 2 : 8)) cyc[1] = t1 - t0;
    }
    out[blockIdx.x * 1024 + threadIdx.x] = sink;
}
template <int SPLIT>
static void run2(const char* name, float* out, unsigned long long* cyc, int* simd) {
    const int iters = 2000;
    double v = 0, m = 0;
    int hs[16];
    for (int pass = 0; pass < 2; ++pass) {
        // pass 0: the MFMA side runs ~3 x longer than the VALU side (its measurement is of VALU under full MFMA load); pass 1 the other way round
        const int iv = pass ? 8 * iters : iters, im = pass ? iters : 4 * iters;
        unsigned long long h[2] = {0, 0};
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(cyc, 0, 16);
            hipLaunchKernelGGL(probe2<SPLIT>, dim3(256), dim3(1024), 0, 0, out, iv, im, cyc, simd);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        if (pass == 0) v = (double)h[0] / (iv * 64.0); else m = (double)h[1] / (im * 12.0);
    }
    hipMemcpy(hs, simd, sizeof(hs), hipMemcpyDeviceToHost);
    printf("%-36s: VALU wave %.2f ticks per instruction (its SIMD: one per %.2f)   MFMA wave %.1f ticks per MFMA (its SIMD: one per %.1f)\n   SIMD of waves 0..15:", name, v,
           v / (SPLIT ? 4 : 2), m, m / (SPLIT ? 4 : 2));
    for (int i = 0; i < 16; ++i) printf(" %d", hs[i]);
    printf("\n");
}

template <int MODE>
static void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    // in the mixed modes each side is measured while the OTHER runs for longer (first pass: long MFMA side, second: long VALU side)
    double v_cpi = 0, m_cpm = 0;
    constexpr bool mixed = (MODE & 3) == 3;
    for (int pass = 0; pass < (mixed ? 2 : 1); ++pass) {
        const int iv = mixed && pass == 1 ? 8 * iters : iters, im = mixed && pass == 0 ? 2 * iters : iters;
        unsigned long long h[2] = {0, 0};
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(cyc, 0, 16);
            hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(768), 65536, 0, out, iv, im, cyc);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        if (!mixed || pass == 0) v_cpi = (double)h[0] / (iv * 64.0);
        if (!mixed || pass == 1) m_cpm = (double)h[1] / (im * 12.0);
    }
    printf("%-36s: ", name);
    if (MODE & 1) printf("VALU wave %.2f cycles per instruction   ", v_cpi);
    if (MODE & 2) printf("MFMA wave %.1f cycles per MFMA (two waves per SIMD: one MFMA per %.1f cycles)", m_cpm, m_cpm / 2);
    printf("\n");
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 768 * 4); hipMalloc(&cyc, 16);
    run<1>("VALU alone", out, cyc);
    run<2>("MFMA alone", out, cyc);
    run<3>("both", out, cyc);
    run<7>("both, VALU waves at s_setprio 3", out, cyc);
    run<9>("packed VALU alone", out, cyc);
    run<11>("packed VALU + MFMA", out, cyc);
    run<17>("VALU + ds_read_b128 alone", out, cyc);
    run<19>("VALU + ds_read_b128 + MFMA", out, cyc);
    run<23>("VALU + ds_read_b128 + MFMA, setprio 3", out, cyc);
    int* simd; hipMalloc(&simd, 64); hipMemset(simd, 0, 64);
    hipMalloc(&out, 256 * 1024 * 4);
    run2<0>("16 waves, roles interleaved", out, cyc, simd);
    run2<1>("16 waves, roles split by SIMD", out, cyc, simd);
    if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
    return 0;
}
