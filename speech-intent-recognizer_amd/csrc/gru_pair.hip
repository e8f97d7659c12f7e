// Launcher of the GRU backward recurrence: the matrix-core cluster kernel (gru_quad.hip) by default, else the paired-workgroup
// fp32-FMA kernels (gru_bwd_pair_kernel.h).  Own translation unit because it is
// compiled with -fno-slp-vectorize: hipcc's SLP pass packs the scalar fmaf chains into v_pk_fma_f32, whose register-pair
// constraints cost ~180 spilled VGPRs in this register-resident kernel.
#include <cstdlib>
#include "gru_bwd_pair_kernel.h"

int sir_launch_gru_bwd_pair(sir_handle* h, hipStream_t st, const float* dy, const float* gates, const float* y, const float* whh0,
                            const float* whh1, float* dgi, float* dgh, float* bsum_i, float* bsum_h, int B, int S, const void* wfrag0,
                            const void* wfrag1) {
    // wfrag0 / 1: the matrix-core kernel's resident fragments (gru_frag_prep.h; optional, used by mode 5 only)
    // SIR_BPTT (A/B and timing knock-outs, profiles/r04/ab_bptt.txt; default 5 = the matrix-core kernel, clusters of four workgroups x
    // 16 utterances, gru_bwd_quad_kernel.h): 4 = the fp32-FMA kernel in its four-k / eight-row-part layout (gru_bwd_pair_k4_kernel; 36 + k
    // its knock-outs), 0 = the two-k kernel with the padded dgh image, 1 = round 3's unpadded image, 2 = two-k + product loop on
    // v_pk_fma_f32, 3 = 1 + 2; 16 + k = knock-out k of the two-k kernel (see the kernels' KNOCK comments)
    static const int mode = getenv("SIR_BPTT") ? atoi(getenv("SIR_BPTT")) : 5;
    if (mode == 5) return sir_launch_gru_bwd_quad(h, st, dy, gates, y, whh0, whh1, dgi, dgh, bsum_i, bsum_h, B, S, wfrag0, wfrag1);
    typedef void (*kern_t)(const float*, const float*, const float*, const float*, const float*, float*, float*, float*, float*, int, int, float*,
                           unsigned int*, unsigned);
    kern_t kern = gru_bwd_pair_k4_kernel<0>;
    switch (mode) {
        case 0: kern = gru_bwd_pair_kernel<0, true, false>; break;
        case 1: kern = gru_bwd_pair_kernel<0, false, false>; break;
        case 2: kern = gru_bwd_pair_kernel<0, true, true>; break;
        case 3: kern = gru_bwd_pair_kernel<0, false, true>; break;
        case 17: kern = gru_bwd_pair_kernel<1, true, false>; break;
        case 18: kern = gru_bwd_pair_kernel<2, true, false>; break;
        case 20: kern = gru_bwd_pair_kernel<4, true, false>; break;
        case 24: kern = gru_bwd_pair_kernel<8, true, false>; break;
        case 19: kern = gru_bwd_pair_kernel<3, true, false>; break;
        case 4: kern = gru_bwd_pair_k4_kernel<0>; break;
        case 37: kern = gru_bwd_pair_k4_kernel<1>; break;
        case 38: kern = gru_bwd_pair_k4_kernel<2>; break;
        case 40: kern = gru_bwd_pair_k4_kernel<4>; break;
        default: break;
    }
    if (!h->attr_gru_bwd) {
        static_assert(GB4_LDS_BYTES <= GBP_LDS_BYTES + 1024, "LDS of the two layouts");
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GB4_LDS_BYTES > GBP_LDS_BYTES ? GB4_LDS_BYTES : GBP_LDS_BYTES)));
        h->attr_gru_bwd = true;
    }
    const size_t npairs = (B + GP_BW - 1) / GP_BW;
    unsigned epoch = 0;
    void* xbuf = nullptr;
    if (sir_xbuf_acquire(h, st, 2, npairs * 2 * 2 * 2 * GP_BW * GP_UH * 8, 0xFFFFu, &xbuf, &epoch) != SIR_OK) {
        return SIR_EHIP;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(npairs * 2), 2), dim3(GP_THREADS), GB4_LDS_BYTES > GBP_LDS_BYTES ? GB4_LDS_BYTES : GBP_LDS_BYTES, st, dy, gates, y, whh0, whh1, dgi,
                       dgh, bsum_i, bsum_h, B, S, (float*)xbuf, h->status, epoch);
    SIR_HIP_TRY(hipGetLastError());
    return SIR_OK;
}
