// Standalone timing/consistency harness of the bf16x6 convolution kernels (developer tool, not part
// of the library): random NHWC input and weights at the batch-256 shapes of conv2 / conv3, both
// kernel generations, HIP-event timing, bitwise comparison, timing-only knock-outs.
//   build: make -C speech-intent-recognizer_amd/csrc tools     run (GPU box): lib/bench_conv
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
#include "../csrc/bf16x6_kernels.h"
#include "../tools/legacy_kernels.h"

#define CK_(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <typename F>
static float time_us(hipStream_t st, int reps, F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 40; ++i) launch();      // long warm-up: the clocks ramp over the first milliseconds of load
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    CK_(hipGetLastError());
    return ms * 1000.0f / reps;
}

template <int CIN, int COUT, int PR, int PC, int OUT_MODE>
static void run(const char* name, int B, int H, int W) {
    const int Hp = H / 2, Wp = W / 2, reps = 60;
    const size_t nx = (size_t)B * H * W * CIN, nw = (size_t)COUT * CIN * 9;
    const size_t nout = OUT_MODE == 2 ? (size_t)B * H * W * COUT : (size_t)B * Hp * Wp * COUT;   // raw mode writes every pixel
    std::vector<float> hx(nx), hw(nw), hs(COUT), ht(COUT);
    srand(7);
    for (auto& v : hx) v = rand() / (float)RAND_MAX * 2.0f - 0.3f;
    for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    for (int c = 0; c < COUT; ++c) { hs[c] = 0.5f + rand() / (float)RAND_MAX; ht[c] = rand() / (float)RAND_MAX - 0.5f; }
    float *dx, *dw, *ds, *dt, *o1, *o2;
    unsigned short* wpb;
    CK_(hipMalloc(&dx, nx * 4)); CK_(hipMalloc(&dw, nw * 4)); CK_(hipMalloc(&ds, COUT * 4)); CK_(hipMalloc(&dt, COUT * 4));
    CK_(hipMalloc(&o1, nout * 4)); CK_(hipMalloc(&o2, nout * 4)); CK_(hipMalloc(&wpb, nw * 6));
    CK_(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK_(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK_(hipMemcpy(ds, hs.data(), COUT * 4, hipMemcpyHostToDevice)); CK_(hipMemcpy(dt, ht.data(), COUT * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK_(hipStreamCreate(&st));
    hipLaunchKernelGGL(prep_conv_w_bf16x3_kernel, dim3((CIN * 9 * COUT + 255) / 256), dim3(256), 0, st, (const float*)dw, wpb, CIN, COUT);
    const dim3 grid((W + 4 * PC - 1) / (4 * PC), (H + 8 * PR - 1) / (8 * PR), B);
    const double gf = 2.0 * B * H * W * (double)COUT * CIN * 9 * 1e-9;
    CK_(hipMemset(o1, 0, nout * 4)); CK_(hipMemset(o2, 0, nout * 4));
    float t1 = 0.0f;
    if constexpr ((PR * PC) % 4 == 0) {
        constexpr size_t lds = conv_bf16x6_lds_bytes(PR, PC);
        t1 = time_us(st, reps, [&] {
            hipLaunchKernelGGL((conv3x3_bf16x6_kernel<CIN, COUT, PR, PC, OUT_MODE, (PR * PC) / 4>), grid, dim3(256), lds, st, (const float*)dx, (const unsigned short*)wpb,
                               (const float*)ds, (const float*)dt, o1, H, W, Hp, Wp, (float2*)nullptr); });
    } else {
        // no gen-1 kernel for this tile: the consistency check is against the 4x2-patch channel-split kernel
        hipLaunchKernelGGL((conv3x3_bf16x6_ns_kernel<CIN, COUT, 4, 2, OUT_MODE, 0>), dim3((W + 7) / 8, (H + 31) / 32, B), dim3(256), conv_ns_lds_bytes(4, 2), st,
                           (const float*)dx, (const unsigned short*)wpb, (const float*)ds, (const float*)dt, o1, H, W, Hp, Wp, (float2*)nullptr);
    }
    constexpr size_t lds_ns = conv_ns_lds_bytes(PR, PC);       // the product kernel's half-major, unpadded LDS image
    float t2 = time_us(st, reps, [&] {
        hipLaunchKernelGGL((conv3x3_bf16x6_ns_kernel<CIN, COUT, PR, PC, OUT_MODE, 0>), grid, dim3(256), lds_ns, st, (const float*)dx, (const unsigned short*)wpb,
                           (const float*)ds, (const float*)dt, o2, H, W, Hp, Wp, (float2*)nullptr); });
    CK_(hipStreamSynchronize(st));
    std::vector<float> h1(nout), h2(nout);
    CK_(hipMemcpy(h1.data(), o1, nout * 4, hipMemcpyDeviceToHost)); CK_(hipMemcpy(h2.data(), o2, nout * 4, hipMemcpyDeviceToHost));
    double d = 0, mx = 0;
    for (size_t i = 0; i < nout; ++i) { d = fmax(d, fabs((double)h1[i] - h2[i])); mx = fmax(mx, fabs(h1[i])); }
    printf("%s: B=%d %dx%d %d->%d  %.2f GF  grid %dx%dx%d\n", name, B, H, W, CIN, COUT, gf, grid.x, grid.y, grid.z);
    printf("  gen1 (pixels split over waves)   %8.1f us  %7.1f TF\n", t1, gf * 1e3 / t1);
    printf("  gen2 (channels split over waves) %8.1f us  %7.1f TF   max |gen1 - gen2| = %.3e (max |out| %.2f)\n", t2, gf * 1e3 / t2, d, mx);
    float t3 = time_us(st, reps, [&] {
        hipLaunchKernelGGL((conv3x3_bf16x6_ns_kernel<CIN, COUT, PR, PC, OUT_MODE, 1>), grid, dim3(256), lds_ns, st, (const float*)dx, (const unsigned short*)wpb,
                           (const float*)ds, (const float*)dt, o2, H, W, Hp, Wp, (float2*)nullptr); });
    float t4 = time_us(st, reps, [&] {
        hipLaunchKernelGGL((conv3x3_bf16x6_ns_kernel<CIN, COUT, PR, PC, OUT_MODE, 2>), grid, dim3(256), lds_ns, st, (const float*)dx, (const unsigned short*)wpb,
                           (const float*)ds, (const float*)dt, o2, H, W, Hp, Wp, (float2*)nullptr); });
    float t5 = time_us(st, reps, [&] {
        hipLaunchKernelGGL((conv3x3_bf16x6_ns_kernel<CIN, COUT, PR, PC, OUT_MODE, 3>), grid, dim3(256), lds_ns, st, (const float*)dx, (const unsigned short*)wpb,
                           (const float*)ds, (const float*)dt, o2, H, W, Hp, Wp, (float2*)nullptr); });
    auto variant = [&](auto minb, auto pipe) {
        return time_us(st, reps, [&] {
            hipLaunchKernelGGL((conv3x3_bf16x6_ns_kernel<CIN, COUT, PR, PC, OUT_MODE, 0, decltype(minb)::value, decltype(pipe)::value>), grid, dim3(256), lds_ns, st,
                               (const float*)dx, (const unsigned short*)wpb, (const float*)ds, (const float*)dt, o2, H, W, Hp, Wp, (float2*)nullptr); });
    };
    using std::integral_constant;
    printf("  workgroups/CU x fragment pipelining (LDS %zu B):  plain 2: %.1f  3: %.1f", lds_ns, variant(integral_constant<int, 2>{}, integral_constant<int, 0>{}),
           variant(integral_constant<int, 3>{}, integral_constant<int, 0>{}));
    if (PR * PC * COUT <= 512) printf("  4: %.1f", variant(integral_constant<int, 4>{}, integral_constant<int, 0>{}));
    printf("   pipelined 2: %.1f  3: %.1f", variant(integral_constant<int, 2>{}, integral_constant<int, 1>{}), variant(integral_constant<int, 3>{}, integral_constant<int, 1>{}));
    if (PR * PC * COUT <= 512) printf("  4: %.1f", variant(integral_constant<int, 4>{}, integral_constant<int, 1>{}));
    printf(" us\n");
    printf("  gen2 knock-outs (timing only): weights once %.1f us, tile staged once %.1f us, both %.1f us\n", t3, t4, t5);
    hipFree(dx); hipFree(dw); hipFree(ds); hipFree(dt); hipFree(o1); hipFree(o2); hipFree(wpb);
}

int main() {
    run<64, 128, 2, 4, 1>("conv3", 256, 16, 50);
    run<64, 128, 2, 2, 1>("conv3, 16x8-pixel tile (4 patches per wave)", 256, 16, 50);
    run<32, 64, 4, 2, 0>("conv2", 256, 32, 100);
    run<32, 64, 2, 5, 0>("conv2, 16x20-pixel tile (5 patches per wave, 10 tiles per utterance)", 256, 32, 100);
    run<128, 64, 2, 4, 2>("conv3 data gradient", 256, 16, 50);
    run<64, 32, 4, 2, 2>("conv2 data gradient", 256, 32, 100);
    return 0;
}
