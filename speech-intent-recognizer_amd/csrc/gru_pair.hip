// Launcher of the paired-workgroup GRU backward recurrence (gru_bwd_pair_kernel.h).  Own translation unit because it is
// compiled with -fno-slp-vectorize: hipcc's SLP pass packs the scalar fmaf chains into v_pk_fma_f32, whose register-pair
// constraints cost ~180 spilled VGPRs in this register-resident kernel.
#include "gru_bwd_pair_kernel.h"

int sir_launch_gru_bwd_pair(sir_handle* h, hipStream_t st, const float* dy, const float* gates, const float* y, const float* whh0,
                            const float* whh1, float* dgi, float* dgh, float* bsum_i, float* bsum_h, int B, int S) {
    if (!h->attr_gru_bwd) {
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)gru_bwd_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GBP_LDS_BYTES));
        h->attr_gru_bwd = true;
    }
    const size_t npairs = (B + GP_BW - 1) / GP_BW;
    unsigned epoch = 0;
    void* xbuf = nullptr;
    if (sir_xbuf_acquire(h, st, 2, npairs * 2 * 2 * 2 * GP_BW * GP_UH * 8, 0xFFFFu, &xbuf, &epoch) != SIR_OK) {
        sir_set_error("gru_bwd_pair: exchange buffer allocation failed");
        return SIR_EHIP;
    }
    hipLaunchKernelGGL(gru_bwd_pair_kernel, dim3((unsigned)(npairs * 2), 2), dim3(GP_THREADS), GBP_LDS_BYTES, st, dy, gates, y, whh0, whh1, dgi,
                       dgh, bsum_i, bsum_h, B, S, (float*)xbuf, h->status, epoch);
    SIR_HIP_TRY(hipGetLastError());
    return SIR_OK;
}
