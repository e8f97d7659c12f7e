// Checks the lane -> element mapping of ds_read_b64_tr_b16 (gfx950) that gemm_tn_bf16x6_kernel / conv_wgrad rely on:
// per group of 16 consecutive lanes, lane 4q+p supplies the address of row q, columns 4p..4p+3 of a 4 x 16 block of 16-bit
// elements; lane i receives column i of the 4 rows (row q in element q).   build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4i16 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ short lds[8][64];
    for (int i = threadIdx.x; i < 8 * 64; i += 64) lds[i / 64][i % 64] = (short)((i / 64) * 256 + (i % 64));
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4i16 __attribute__((address_space(3)))*)(&lds[q][16 * g + 4 * p]));
    out[l * 4 + 0] = v.x; out[l * 4 + 1] = v.y; out[l * 4 + 2] = v.z; out[l * 4 + 3] = v.w;
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
            const int expect = e * 256 + 16 * (l >> 4) + (l & 15);
            if (h[l * 4 + e] != expect) { if (bad < 8) printf("lane %d elem %d: got %d expected %d\n", l, e, h[l * 4 + e], expect); ++bad; }
        }
    printf(bad ? "MISMATCH (%d)\n" : "tr-read mapping as documented (%d mismatches)\n", bad);
    return bad != 0;
}
