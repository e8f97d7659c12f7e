// Launcher of the quad-workgroup MFMA GRU recurrence (gru_quad_kernel.h).
#include <stdlib.h>
#include "gru_quad_kernel.h"
#include "gru_bwd_quad_kernel.h"

int sir_launch_gru_quad(sir_handle* h, hipStream_t st, bool save, const float* gi, const float* whh0, const float* whh1, const float* bhh0,
                        const float* bhh1, float* y, int B, int S, float* gates, unsigned short* yplanes, const void* wfrag0,
                        const void* wfrag1) {
    if (!h->attr_gru_quad) {
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)gru_quad_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GQ_LDS_BYTES));
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)gru_quad_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GQ_LDS_BYTES));
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)gru_quad_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GQ_LDS_BYTES));
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)gru_quad_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GQ_LDS_BYTES));
        h->attr_gru_quad = true;
    }
    if (S >= 511) { sir_set_error("gru_quad: %d steps exceed the 9-bit step field of the granule tag", S); return SIR_EUNSUPPORTED; }
    if (!wfrag0 || !wfrag1) { sir_set_error("gru_quad: the prepared W_hh fragments are required (sir_prep_whh_quad / train_prep_kernel)"); return SIR_EINVAL; }
    const int clusters = ((B + GQ_NU - 1) / GQ_NU) * 2;
    unsigned epoch = 0;
    void* xbuf = nullptr;
    if (sir_xbuf_acquire(h, st, 1, (size_t)clusters * GQ_XBUF_PER_CLUSTER, 127u, &xbuf, &epoch) != SIR_OK) {
        return SIR_EHIP;
    }
    const dim3 grid(4 * (unsigned)clusters);
    // SIR_GRU_DBG: timing knock-outs and fault injection of gru_quad_kernel (see its `dbg` comment); 0 in production
    static const int dbg0 = getenv("SIR_GRU_DBG") ? atoi(getenv("SIR_GRU_DBG")) : 0;
    static const int delay = getenv("SIR_GQ_DELAY") ? atoi(getenv("SIR_GQ_DELAY")) & 31 : GQ_POLL_DELAY;      // A/B of the first poll's delay
    const int dbg = (dbg0 & ~(31 << 8)) | ((((dbg0 >> 8) & 31) ? ((dbg0 >> 8) & 31) : delay) << 8);
    // SIR_GQ_ROLES (default 1): gate arithmetic and global accesses in the coalesced thread layout -- bit 0: the gate-saving (training)
    // form, whose five 16-byte stores per lane and step were fully exposed (layer 0: 79.5 -> 71.4 us); bit 1: the inference form, where
    // the extra barrier costs more than its one to three stores (71.6 -> 74.7 us: off).  profiles/r04/ab_gq_roles.txt
    static const int roles = getenv("SIR_GQ_ROLES") ? atoi(getenv("SIR_GQ_ROLES")) : 1;
    typedef void (*kern_t)(const float*, const float*, const float*, const float*, const float*, float*, int, int, float*, unsigned long long*,
                           unsigned int*, int, unsigned, unsigned short*, const uint4*, const uint4*);
    const kern_t kern = save ? ((roles & 1) ? gru_quad_kernel<true, true> : gru_quad_kernel<true, false>)
                             : ((roles & 2) ? gru_quad_kernel<false, true> : gru_quad_kernel<false, false>);
    hipLaunchKernelGGL(kern, grid, dim3(GQ_THREADS), GQ_LDS_BYTES, st, gi, whh0, whh1, bhh0, bhh1, y, B, S, gates,
                       (unsigned long long*)xbuf, h->status, dbg, epoch, yplanes, (const uint4*)wfrag0, (const uint4*)wfrag1);
    SIR_HIP_TRY(hipGetLastError());
    return SIR_OK;
}

// inference-side preparation of one direction's resident fragments (GRU_FRAG_BYTES)
void sir_prep_whh_quad(hipStream_t st, const float* whh, void* frag) {
    hipLaunchKernelGGL(prep_whh_quad_kernel, dim3(GQ_FRAG_THREADS / 256), dim3(256), 0, st, whh, (uint4*)frag);
}

// BPTT on the matrix cores, clusters of four workgroups x 16 utterances (gru_bwd_quad_kernel.h); same contract as sir_launch_gru_bwd_pair
int sir_launch_gru_bwd_quad(sir_handle* h, hipStream_t st, const float* dy, const float* gates, const float* y, const float* whh0,
                            const float* whh1, float* dgi, float* dgh, float* bsum_i, float* bsum_h, int B, int S, const void* wfrag0,
                            const void* wfrag1) {
    if (!h->attr_gru_bwd_quad) {
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)gru_bwd_quad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BQ_LDS_BYTES));
        h->attr_gru_bwd_quad = true;
    }
    if (S >= 65535) { sir_set_error("gru_bwd_quad: %d steps exceed the 16-bit step field of the granule tag", S); return SIR_EUNSUPPORTED; }
    if (!wfrag0 || !wfrag1) { sir_set_error("gru_bwd_quad: the prepared W_hh fragments are required (train_prep_kernel)"); return SIR_EINVAL; }
    const int clusters = ((B + GQ_NU - 1) / GQ_NU) * 2;
    unsigned epoch = 0;
    void* xbuf = nullptr;
    if (sir_xbuf_acquire(h, st, 3, (size_t)clusters * BQ_XBUF_PER_CLUSTER, 0xFFFFu, &xbuf, &epoch) != SIR_OK) return SIR_EHIP;
    // SIR_BQ_DBG: timing knock-outs of gru_bwd_quad_kernel (see its `dbg` comment), 0 in production; SIR_BQ_DELAY: first poll's delay
    static const int dbg0 = getenv("SIR_BQ_DBG") ? atoi(getenv("SIR_BQ_DBG")) : 0;
    static const int delay = getenv("SIR_BQ_DELAY") ? atoi(getenv("SIR_BQ_DELAY")) & 31 : GQ_POLL_DELAY;
    const int dbg = (dbg0 & 255) | (delay << 8);
    hipLaunchKernelGGL(gru_bwd_quad_kernel, dim3(4 * (unsigned)clusters), dim3(GQ_THREADS), BQ_LDS_BYTES, st, dy, gates, y, whh0, whh1, dgi, dgh,
                       bsum_i, bsum_h, B, S, (unsigned long long*)xbuf, h->status, epoch, dbg, (const uint4*)wfrag0, (const uint4*)wfrag1);
    SIR_HIP_TRY(hipGetLastError());
    return SIR_OK;
}
