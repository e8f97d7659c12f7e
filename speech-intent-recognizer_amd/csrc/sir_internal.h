// Internal declarations shared by the HIP translation units of libsir_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <vector>
#include "sir_hip.h"

void sir_set_error(const char* fmt, ...);

#define SIR_WAVE 64               // CDNA wavefront width (hard-coded, see cdna guide section 1)
#define SIR_NFFT 1024
#define SIR_HOP 512
#define SIR_NFREQ 513
#define SIR_MAX_MELS 64

// kernels (or kernel groups) of the path, for sir_profile_*
enum SirKernelId {
    SIR_K_FEAT_FRAMES = 0, SIR_K_FEAT_NORM, SIR_K_PREP, SIR_K_CONV1, SIR_K_CONV2, SIR_K_CONV3,
    SIR_K_GEMM_IH0, SIR_K_GRU0, SIR_K_GEMM_IH1, SIR_K_GRU1, SIR_K_ATTN, SIR_K_FC,
    // training step (model_train.hip): forward groups, loss, backward groups, optimizer
    SIR_K_T_PREP, SIR_K_T_CONV1, SIR_K_T_CONV2, SIR_K_T_BN2, SIR_K_T_CONV3, SIR_K_T_BN3, SIR_K_T_GEMM_IH0, SIR_K_T_GRU0,
    SIR_K_T_DROPOUT, SIR_K_T_GEMM_IH1, SIR_K_T_GRU1, SIR_K_T_HEAD, SIR_K_CE,
    SIR_K_B_HEAD, SIR_K_B_GRU1, SIR_K_B_DW1, SIR_K_B_DX1, SIR_K_B_GRU0, SIR_K_B_DW0, SIR_K_B_DX0,
    SIR_K_B_BN3, SIR_K_B_WGRAD3, SIR_K_B_DGRAD3, SIR_K_B_BN2, SIR_K_B_WGRAD2, SIR_K_B_DGRAD2, SIR_K_B_CONV1, SIR_K_ADAM,
    SIR_K_COUNT
};

struct SirProfRec { int id; hipEvent_t e0, e1; };
static inline size_t sir_align_up_sz(size_t x, size_t a) { return (x + a - 1) / a * a; }

// polyphase resampling filter of one (orig_freq, new_freq) pair (frontend.hip), device tables
struct sir_resample_table {
    int orig_freq, new_freq;    // as passed by the caller
    int orig, nw, width, L;     // gcd-reduced rates, torchaudio's `width`, taps kept per phase
    float* taps;                // [nw][L]
    int* first;                 // [nw] index (in torchaudio's kernel row) of the first kept tap
};

struct sir_handle {
    // event profiling state (host only)
    int prof_mode;      // 0 off, 1 every kernel, 2 only prof_only
    int prof_only;
    std::vector<SirProfRec> prof_pending;
    std::vector<hipEvent_t> prof_free;
    // prepared-weight cache key (sir_model_set_weights_version)
    unsigned long long weights_version;
    // one entry per workspace that holds prepared weights (several streams may alternate workspaces)
    struct PrepEntry { const void* ws; unsigned long long version; long long key; };
    PrepEntry prep[4];
    int prep_next;
    sir_feature_config cfg;
    int device;
    // feature tables (device)
    float2* tw512;      // exp(-2*pi*i*j/512),  j = 0..511
    float2* tw1024;     // exp(-2*pi*i*k/1024), k = 0..512
    float* window;      // [1024]
    float* melw;        // compact filter weights: filter after filter, taps ascending in frequency (mel_nnz floats)
    int4* mel_desc;     // [64] per slot, filters sorted by tap count: {filter (-1 = unused), first FFT bin, taps, offset into melw}
    int mel_nnz;
    bool feat_attr_set; // the feature kernel's dynamic-LDS opt-in has been made on this handle's device
    std::vector<sir_resample_table> resample_tables;   // built on first use of a rate pair
    // device word set to 1 by a GRU recurrence kernel whose inter-workgroup exchange timed out (its results are then
    // invalid); zeroed at creation, read and cleared by sir_check_status / sir_profile_collect
    unsigned int* status;
    // Cluster kernels (the GRU recurrences) need every workgroup of a cluster resident at once; dispatch order
    // guarantees that within ONE launch, but two such launches on different streams interleave their dispatch and, once
    // their workgroups outnumber the CUs, can fill the chip with partial clusters that wait for each other for ever (seen
    // with 2 processes x 2 streams on one GPU: spin time-outs).  Launches of cluster kernels issued through this handle are
    // therefore chained: one on another stream first waits for the previous one's completion event (sir_cluster_enter).
    hipEvent_t cluster_done;
    hipStream_t cluster_stream;        // stream of the latest cluster launch (compared, never dereferenced)
    bool cluster_pending, cluster_seen, cluster_multi, cluster_always;
    int cluster_run;                   // chained mode: launches in a row that came from cluster_stream
    // hipFuncSetAttribute(MaxDynamicSharedMemorySize) latches, per handle = per device (a process-wide static would skip
    // the second device of a process that drives several)
    bool attr_gemm_v3, attr_gru_quad, attr_gru_bwd, attr_gru_bwd_quad, attr_tn, attr_wgrad;
    bool attr_wino2[16];               // conv3x3_wino2_bf16x6_kernel instantiations (model_infer.hip / model_train.hip index them)
    float* zero_page;                  // 4 KB of zeros: DMA source of the second-generation Winograd kernel's out-of-image pixels
    int num_cus;                       // persistent kernels launch one workgroup per CU
    // Exchange-granule buffers of the cluster kernels (GRU recurrences).  They are OWNED by the handle (hipMalloc), one per
    // (launch stream, kernel kind): nothing but that kernel ever writes them, so a granule found there is always one of its
    // own from an earlier launch and the launch epoch in its tag tells it apart.  (They used to be carved out of the caller's
    // workspace at a batch-dependent offset: a launch at another batch size, or any other tenant of that memory, could leave
    // arbitrary bits where the 16-bit tag of the forward kernel is polled.)  Launches on one stream are ordered, launches on
    // different streams get different buffers.  A buffer is zeroed when it is new or must cover more bytes than the
    // previous launch on it wrote -- not before every launch (4.8 us each, 4 per training step / 2 per inference batch,
    // serialised in front of a latency-bound kernel).
    struct XbufEntry { hipStream_t st; int kind; void* p; size_t cap, bytes; unsigned epoch; unsigned long long used; };
    XbufEntry xbufs[16];
    unsigned long long xbuf_clock;
    unsigned long long xbuf_evictions;   // LRU evictions so far (each costs a device-synchronising hipFree + hipMalloc on the launch path)
    // second stream of the training backward (SIR_BWD_STREAMS=1, model_train.hip): the off-chain weight-gradient launches; created on
    // first use.  ev: 0 fork behind the GRU part, 1 / 2 dz3 / dz2 ready, 3 join
    hipStream_t bwd_side;
    hipEvent_t bwd_ev[6];
};

// granule buffer + launch epoch for a cluster kernel launched on `st`; allocates / grows / zeroes the buffer when needed.
// `kind` separates kernels with different granule formats (gru_quad_kernel.h / gru_bwd_pair_kernel.h).
static inline int sir_xbuf_acquire(sir_handle* h, hipStream_t st, int kind, size_t bytes, unsigned mask, void** xbuf, unsigned* epoch) {
    sir_handle::XbufEntry* e = nullptr;
    for (auto& x : h->xbufs)
        if (x.p && x.st == st && x.kind == kind) e = &x;
    if (!e) {                                             // free slot, else the least recently used one
        e = &h->xbufs[0];
        for (auto& x : h->xbufs) {
            if (!x.p) { e = &x; break; }
            if (x.used < e->used) e = &x;
        }
        if (e->p) {                                       // hipFree waits for the device: no launch can still be using it
            ++h->xbuf_evictions;                          // (more than 16 live (stream, kind) pairs: every launch then frees + allocates)
            if (hipFree(e->p) != hipSuccess) { sir_set_error("exchange buffer: hipFree of an evicted buffer failed"); return SIR_EHIP; }
            e->p = nullptr;
        }
        e->st = st; e->kind = kind; e->cap = 0; e->bytes = 0; e->epoch = 0;
    }
    if (bytes > e->cap) {
        if (e->p && hipFree(e->p) != hipSuccess) { sir_set_error("exchange buffer: hipFree before growing failed"); return SIR_EHIP; }
        e->p = nullptr; e->cap = 0; e->bytes = 0;
        const size_t cap = sir_align_up_sz(bytes, (size_t)1 << 20);
        if (hipMalloc(&e->p, cap) != hipSuccess) { e->p = nullptr; sir_set_error("exchange buffer: hipMalloc of %zu bytes failed", cap); return SIR_EHIP; }
        e->cap = cap;
    }
    // new buffer, or more clusters than the PREVIOUS launch on it wrote: the extra granules are older than one epoch (after a
    // run of 128 smaller batches they would carry the current epoch again) -- zero.  Otherwise every granule that will be
    // polled was written by the previous launch, whose epoch differs.
    if (bytes > e->bytes && hipMemsetAsync(e->p, 0, bytes, st) != hipSuccess) { sir_set_error("exchange buffer: hipMemsetAsync failed"); return SIR_EHIP; }
    e->bytes = bytes;
    e->epoch = (e->epoch + 1) & mask;
    e->used = ++h->xbuf_clock;
    *epoch = e->epoch;
    *xbuf = e->p;
    return SIR_OK;
}

// bracket of a cluster-kernel launch (see sir_handle::cluster_done).  The completion event costs the stream a ~6 us bubble
// per launch (the next kernel waits for the marker packet), so it is only recorded while launches really come from more
// than one stream: a handle starts in single-stream mode, the first launch from a second stream drains the device once
// and switches to chained mode, 64 launches in a row from one stream switch back.
static inline int sir_cluster_enter(sir_handle* h, hipStream_t st) {
    if (!h->cluster_multi) {
        if (h->cluster_seen && h->cluster_stream != st) {
            if (hipDeviceSynchronize() != hipSuccess) return SIR_EHIP;
            h->cluster_multi = true;
            h->cluster_pending = false;
            h->cluster_run = 0;
        }
        return SIR_OK;
    }
    if (h->cluster_pending && h->cluster_stream != st)
        if (hipStreamWaitEvent(st, h->cluster_done, 0) != hipSuccess) return SIR_EHIP;
    return SIR_OK;
}
static inline int sir_cluster_leave(sir_handle* h, hipStream_t st) {
    if (h->cluster_multi) {
        if (hipEventRecord(h->cluster_done, st) != hipSuccess) return SIR_EHIP;
        h->cluster_pending = true;
        h->cluster_run = (h->cluster_stream == st) ? h->cluster_run + 1 : 0;
        if (h->cluster_run >= 64 && !h->cluster_always) { h->cluster_multi = false; h->cluster_pending = false; }
    }
    h->cluster_stream = st;
    h->cluster_seen = true;
    return SIR_OK;
}

// SIR_WINO2 (default 15): bit 0 = conv2, bit 1 = conv3, bit 2 = conv3 data gradient, bit 3 = conv2 data gradient (f16x3 only) on the producer / consumer Winograd kernel
// (conv_wino2_bf16x6_kernel.h); a cleared bit keeps the first-generation / direct kernel of that stage (A/B on one box:
// devtools/gpu_ab_wino2.sh, profiles/r03/ab_wino2.txt)
int sir_wino2_mask();
int sir_wgw_mask();      // SIR_WGW: convolution weight gradients in Winograd form: bit 0 = conv2, bit 1 = conv3 (default 3)
int sir_tn2_mask();      // SIR_TN2: GRU backward GEMMs on the producer / consumer kernel: bit 0 = dW, bit 1 = dX on 128-row tiles, bit 2 = dX on 64-row tiles, bit 3 = a dX that would take 64-row tiles runs as two K halves on 128-row tiles instead (default 15)
int sir_f16_mask();      // SIR_F16: stages on the f16x3 arithmetic (f16_split.h) instead of bf16x6: bit 0 = conv2 forward, bit 1 = conv3 forward (inference and training), bit 2 = conv3 data gradient, bit 3 = GRU backward GEMMs (dW, dX), bit 4 = conv2 data gradient (on the second-generation Winograd kernel: needs SIR_WINO2 bit 3), bit 5 = convolution weight gradients (default 63)
int sir_bwd_streams();   // SIR_BWD_STREAMS: which of the backward's weight-gradient launches run on a second, handle-owned stream (default 3; model_train.hip)

int sir_check_hip(hipError_t e, const char* what);

#define SIR_HIP_TRY(expr)                                   \
    do {                                                    \
        int _rc = sir_check_hip((expr), #expr);             \
        if (_rc != SIR_OK) return _rc;                      \
    } while (0)

// RAII: records a HIP event pair on `st` around the launches issued while it is alive
struct SirProfScope {
    sir_handle* h; int id; hipStream_t st; hipEvent_t e0, e1; bool on;
    SirProfScope(sir_handle* h_, int id_, hipStream_t st_) : h(h_), id(id_), st(st_), e0(nullptr), e1(nullptr) {
        on = h->prof_mode == 1 || (h->prof_mode == 2 && h->prof_only == id);
        if (!on) return;
        hipEvent_t ev[2];
        for (int i = 0; i < 2; ++i) {
            if (!h->prof_free.empty()) { ev[i] = h->prof_free.back(); h->prof_free.pop_back(); }
            else if (hipEventCreate(&ev[i]) != hipSuccess) { on = false; return; }
        }
        e0 = ev[0]; e1 = ev[1];
        (void)hipEventRecord(e0, st);
    }
    ~SirProfScope() {
        if (!on) return;
        (void)hipEventRecord(e1, st);
        h->prof_pending.push_back(SirProfRec{id, e0, e1});
    }
};

static inline size_t sir_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// model_train.hip
size_t sir_train_workspace_bytes_impl(int batch, int t_frames);

// features.hip
int sir_features_launch(sir_handle* h, const void* wave, int wave_dtype, int64_t wave_stride,
                        const int32_t* lengths, int batch, int max_len, float* out, int t_pad,
                        float* db_out, void* workspace, size_t workspace_bytes, const sir_augment* aug,
                        hipStream_t stream);

// GRU recurrences (gru_quad.hip: forward, clusters of four workgroups on the matrix cores; gru_pair.hip: BPTT, pairs of
// workgroups).  Both write h->status if an exchange spin times out.
int sir_launch_gru_quad(sir_handle* h, hipStream_t st, bool save, const float* gi, const float* whh0, const float* whh1, const float* bhh0,
                        const float* bhh1, float* y, int B, int S, float* gates,
                        unsigned short* yplanes = nullptr, const void* wfrag0 = nullptr, const void* wfrag1 = nullptr);
void sir_prep_whh_quad(hipStream_t st, const float* whh, void* frag);     // -> GRU_FRAG_BYTES (gru_frag_prep.h)
int sir_launch_gru_bwd_pair(sir_handle* h, hipStream_t st, const float* dy, const float* gates, const float* y, const float* whh0,
                            const float* whh1, float* dgi, float* dgh, float* bsum_i, float* bsum_h, int B, int S,
                            const void* wfrag0 = nullptr, const void* wfrag1 = nullptr);
int sir_launch_gru_bwd_quad(sir_handle* h, hipStream_t st, const float* dy, const float* gates, const float* y, const float* whh0,
                            const float* whh1, float* dgi, float* dgh, float* bsum_i, float* bsum_h, int B, int S,
                            const void* wfrag0 = nullptr, const void* wfrag1 = nullptr);
