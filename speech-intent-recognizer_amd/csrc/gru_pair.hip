// Launcher of the paired-workgroup GRU recurrence (gru_pair_kernel.h).  Own translation unit because it
// is compiled with -fno-slp-vectorize: hipcc's SLP pass packs the scalar fmaf chains into v_pk_fma_f32,
// whose register-pair constraints cost ~180 spilled VGPRs in this register-resident kernel.
#include <stdlib.h>
#include "gru_pair_kernel.h"
#include "gru_bwd_pair_kernel.h"

// exchange-granule workspace shared by the pair and the quad kernels: the larger of the two
size_t sir_gru_pair_xbuf_bytes(int batch) {
    const size_t pair = (size_t)((batch + GP_BW - 1) / GP_BW) * 2 * 2 * 2 * GP_BW * GP_UH * 8;
    const size_t quad = sir_gru_quad_xbuf_bytes(batch);
    return pair > quad ? pair : quad;
}
size_t sir_gru_pair_flag_bytes(int batch) { return (size_t)((batch + GP_BW - 1) / GP_BW) * 2 * 2 * 4 + 256; }

int sir_launch_gru_pair(hipStream_t st, bool save, const float* gi, const float* whh0, const float* whh1, const float* bhh0,
                        const float* bhh1, float* y, int B, int S, float* gates, float* xbuf, unsigned int* flags, unsigned int* status) {
    static bool attr = false;
    if (!attr) {
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)gru_pair_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GP_LDS_BYTES));
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)gru_pair_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GP_LDS_BYTES));
        attr = true;
    }
    const size_t npairs = (B + GP_BW - 1) / GP_BW;
    SIR_HIP_TRY(hipMemsetAsync(xbuf, 0, (size_t)((B + GP_BW - 1) / GP_BW) * 2 * 2 * 2 * GP_BW * GP_UH * 8, st));    // tags are re-armed before every launch
    const dim3 grid((unsigned)(npairs * 2), 2);
    static const int nowait = getenv("SIR_GRU_DBG_NOWAIT") ? atoi(getenv("SIR_GRU_DBG_NOWAIT")) : 0;
    if (save)
        hipLaunchKernelGGL(gru_pair_kernel<true>, grid, dim3(GP_THREADS), GP_LDS_BYTES, st, gi, whh0, whh1, bhh0, bhh1, y, B, S, gates,
                           xbuf, flags, status, nowait);
    else
        hipLaunchKernelGGL(gru_pair_kernel<false>, grid, dim3(GP_THREADS), GP_LDS_BYTES, st, gi, whh0, whh1, bhh0, bhh1, y, B, S, gates,
                           xbuf, flags, status, nowait);
    SIR_HIP_TRY(hipGetLastError());
    return SIR_OK;
}

int sir_launch_gru_bwd_pair(hipStream_t st, const float* dy, const float* gates, const float* y, const float* whh0, const float* whh1,
                            float* dgi, float* dgh, float* bsum_i, float* bsum_h, int B, int S, float* xbuf, unsigned int* status) {
    static bool attr = false;
    if (!attr) {
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)gru_bwd_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GBP_LDS_BYTES));
        attr = true;
    }
    const size_t npairs = (B + GP_BW - 1) / GP_BW;
    SIR_HIP_TRY(hipMemsetAsync(xbuf, 0, npairs * 2 * 2 * 2 * GP_BW * GP_UH * 8, st));     // tags are re-armed before every launch
    hipLaunchKernelGGL(gru_bwd_pair_kernel, dim3((unsigned)(npairs * 2), 2), dim3(GP_THREADS), GBP_LDS_BYTES, st, dy, gates, y, whh0, whh1, dgi,
                       dgh, bsum_i, bsum_h, B, S, xbuf, status);
    SIR_HIP_TRY(hipGetLastError());
    return SIR_OK;
}
