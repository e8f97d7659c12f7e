// C ABI entry points of libsir_hip.so (see include/sir_hip.h): handle management, error state,
// argument validation.  Kernels live in features.hip / model_*.hip.
#include "sir_internal.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <vector>

static thread_local char g_err[512] = "";

void sir_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int sir_check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return SIR_OK;
    sir_set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
    return SIR_EHIP;
}

extern "C" int sir_abi_version(void) { return SIR_ABI_VERSION; }
extern "C" const char* sir_last_error(void) { return g_err; }

template <typename T>
static int upload(T** dst, const std::vector<T>& src) {
    SIR_HIP_TRY(hipMalloc((void**)dst, src.size() * sizeof(T)));
    SIR_HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return SIR_OK;
}

int sir_wino2_mask() {
    static const int m = getenv("SIR_WINO2") ? atoi(getenv("SIR_WINO2")) : 15;
    return m;
}

int sir_wgw_mask() {
    static const int m = getenv("SIR_WGW") ? atoi(getenv("SIR_WGW")) : 3;
    return m;
}

int sir_tn2_mask() {
    static const int m = getenv("SIR_TN2") ? atoi(getenv("SIR_TN2")) : 15;
    return m;
}

int sir_f16_mask() {
    static const int m = getenv("SIR_F16") ? atoi(getenv("SIR_F16")) : 63;
    return m;
}

int sir_bwd_streams() {
    // default 3: the weight-gradient launches of the backward run on a second stream owned by the handle (model_train.hip);
    // 0 = everything on the caller's stream (profiles/r04/ab_bwd_streams.txt)
    static const int m = getenv("SIR_BWD_STREAMS") ? atoi(getenv("SIR_BWD_STREAMS")) : 3;
    return m;
}

extern "C" int sir_create(const sir_feature_config* cfg, sir_handle** out) {
    if (!cfg || !out) { sir_set_error("sir_create: NULL argument"); return SIR_EINVAL; }
    if (cfg->n_fft != SIR_NFFT || cfg->hop_length != SIR_HOP) {
        sir_set_error("sir_create: only n_fft=1024 / hop=512 are built (got %d / %d)", cfg->n_fft, cfg->hop_length);
        return SIR_EUNSUPPORTED;
    }
    if (cfg->n_mels < 1 || cfg->n_mels > SIR_MAX_MELS || cfg->sample_rate <= 0) {
        sir_set_error("sir_create: n_mels=%d sample_rate=%d out of range", cfg->n_mels, cfg->sample_rate);
        return SIR_EINVAL;
    }
    sir_handle* h = new sir_handle();
    h->prof_mode = 0; h->prof_only = -1;
    h->weights_version = 0; h->prep_next = 0;
    for (auto& e : h->prep) { e.ws = nullptr; e.version = 0; e.key = -1; }
    h->tw512 = nullptr; h->tw1024 = nullptr; h->window = nullptr; h->melw = nullptr; h->mel_desc = nullptr;
    h->status = nullptr;
    h->cluster_done = nullptr; h->cluster_stream = nullptr; h->cluster_pending = false;
    h->cluster_seen = false; h->cluster_multi = false; h->cluster_run = 0;
    h->cluster_always = getenv("SIR_CLUSTER_EVENTS") && atoi(getenv("SIR_CLUSTER_EVENTS")) == 1;   // A/B switch: chained mode throughout
    if (h->cluster_always) h->cluster_multi = true;
    h->attr_gemm_v3 = h->attr_gru_quad = h->attr_gru_bwd = h->attr_gru_bwd_quad = h->attr_tn = h->attr_wgrad = false;
    for (auto& a : h->attr_wino2) a = false;
    h->zero_page = nullptr; h->num_cus = 256;
    for (auto& x : h->xbufs) { x.st = nullptr; x.p = nullptr; x.kind = 0; x.cap = 0; x.bytes = 0; x.epoch = 0; x.used = 0; }
    h->xbuf_clock = 0;
    h->xbuf_evictions = 0;
    h->bwd_side = nullptr;
    for (auto& e : h->bwd_ev) e = nullptr;
    h->cfg = *cfg;
    h->cfg.window = nullptr;
    h->cfg.mel_fb = nullptr;
    if (sir_check_hip(hipGetDevice(&h->device), "hipGetDevice") != SIR_OK) { delete h; return SIR_EHIP; }

    const double PI = 3.14159265358979323846;
    std::vector<float2> tw512(512), tw1024(SIR_NFREQ);
    for (int j = 0; j < 512; ++j) tw512[j] = make_float2((float)cos(-2.0 * PI * j / 512.0), (float)sin(-2.0 * PI * j / 512.0));
    for (int k = 0; k < SIR_NFREQ; ++k) tw1024[k] = make_float2((float)cos(-2.0 * PI * k / 1024.0), (float)sin(-2.0 * PI * k / 1024.0));
    std::vector<float> window(SIR_NFFT);
    for (int n = 0; n < SIR_NFFT; ++n)
        window[n] = cfg->window ? cfg->window[n] : (float)(0.5 - 0.5 * cos(2.0 * PI * n / SIR_NFFT));

    // dense HTK filterbank (torchaudio melscale_fbanks, norm=None), then compacted per filter
    const int nm = cfg->n_mels;
    std::vector<float> fb((size_t)SIR_NFREQ * nm);
    if (cfg->mel_fb) {
        memcpy(fb.data(), cfg->mel_fb, fb.size() * sizeof(float));
    } else {
        const double f_lo = cfg->f_min, f_hi = cfg->f_max > 0 ? cfg->f_max : cfg->sample_rate / 2;
        const double mmin = 2595.0 * log10(1.0 + f_lo / 700.0), mmax = 2595.0 * log10(1.0 + f_hi / 700.0);
        std::vector<double> fpts(nm + 2);
        for (int i = 0; i < nm + 2; ++i) fpts[i] = 700.0 * (pow(10.0, (mmin + (mmax - mmin) * i / (nm + 1)) / 2595.0) - 1.0);
        for (int k = 0; k < SIR_NFREQ; ++k) {
            const double f = (double)(cfg->sample_rate / 2) * k / (SIR_NFREQ - 1);
            for (int j = 0; j < nm; ++j) {
                const double up = (f - fpts[j]) / (fpts[j + 1] - fpts[j]), down = (fpts[j + 2] - f) / (fpts[j + 2] - fpts[j + 1]);
                const double v = up < down ? up : down;
                fb[(size_t)k * nm + j] = (float)(v > 0.0 ? v : 0.0);
            }
        }
    }
    // compact per-filter tap lists; slots = filters sorted by tap count (feat_utt_kernel: a wave handles four
    // consecutive slots, i.e. four equally long filters)
    std::vector<int> start(SIR_MAX_MELS, 0), count(SIR_MAX_MELS, 0);
    for (int j = 0; j < nm; ++j) {
        int lo = -1, hi = -1;
        for (int k = 0; k < SIR_NFREQ; ++k)
            if (fb[(size_t)k * nm + j] != 0.0f) { if (lo < 0) lo = k; hi = k; }
        if (lo >= 0) { start[j] = lo; count[j] = hi - lo + 1; }
    }
    std::vector<int> order(SIR_MAX_MELS);
    for (int j = 0; j < SIR_MAX_MELS; ++j) order[j] = j;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        const int ca = a < nm ? count[a] : -1, cb = b < nm ? count[b] : -1;      // unused slots first
        return ca < cb;
    });
    std::vector<float> melw;
    std::vector<int4> desc(SIR_MAX_MELS);
    for (int s = 0; s < SIR_MAX_MELS; ++s) {
        const int j = order[s];
        if (j >= nm) { desc[s] = make_int4(-1, 0, 0, 0); continue; }
        desc[s] = make_int4(j, start[j], count[j], (int)melw.size());
        for (int i = 0; i < count[j]; ++i) melw.push_back(fb[(size_t)(start[j] + i) * nm + j]);
    }
    if (melw.size() > 4096) {
        sir_set_error("sir_create: the mel filterbank has %zu taps; only banded (triangular) banks of <= 4096 taps are built", melw.size());
        delete h;
        return SIR_EUNSUPPORTED;
    }
    h->mel_nnz = (int)melw.size();
    melw.resize((melw.size() + 3) / 4 * 4 + 4, 0.0f);
    h->feat_attr_set = false;

    int rc = upload(&h->tw512, tw512);
    if (rc == SIR_OK) rc = upload(&h->tw1024, tw1024);
    if (rc == SIR_OK) rc = upload(&h->window, window);
    if (rc == SIR_OK) rc = upload(&h->melw, melw);
    if (rc == SIR_OK) rc = upload(&h->mel_desc, desc);
    if (rc == SIR_OK) rc = upload(&h->status, std::vector<unsigned int>(64, 0u));
    if (rc == SIR_OK) rc = upload(&h->zero_page, std::vector<float>(1024, 0.0f));
    if (rc == SIR_OK) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && cus > 0) h->num_cus = cus < 1024 ? cus : 1024;
    }
    if (rc == SIR_OK) rc = sir_check_hip(hipEventCreateWithFlags(&h->cluster_done, hipEventDisableTiming), "hipEventCreate");
    if (rc != SIR_OK) { sir_destroy(h); return rc; }
    *out = h;
    return SIR_OK;
}

extern "C" int sir_destroy(sir_handle* h) {
    if (!h) return SIR_OK;
    (void)hipFree(h->tw512); (void)hipFree(h->tw1024); (void)hipFree(h->window);
    (void)hipFree(h->melw); (void)hipFree(h->mel_desc); (void)hipFree(h->status);
    if (h->cluster_done) (void)hipEventDestroy(h->cluster_done);
    if (h->bwd_side) { (void)hipStreamSynchronize(h->bwd_side); (void)hipStreamDestroy(h->bwd_side); }
    for (auto e : h->bwd_ev) if (e) (void)hipEventDestroy(e);
    for (auto& x : h->xbufs) (void)hipFree(x.p);
    (void)hipFree(h->zero_page);
    for (auto& t : h->resample_tables) { (void)hipFree(t.taps); (void)hipFree(t.first); }
    for (auto& r : h->prof_pending) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto e : h->prof_free) (void)hipEventDestroy(e);
    delete h;
    return SIR_OK;
}

extern "C" int sir_features_fwd(sir_handle* h, const void* wave, int wave_dtype, int64_t wave_stride,
                                const int32_t* lengths, int batch, int max_len, float* out, int t_pad,
                                float* db_out, void* workspace, size_t workspace_bytes, const sir_augment* aug,
                                void* stream) {
    return sir_features_launch(h, wave, wave_dtype, wave_stride, lengths, batch, max_len, out, t_pad, db_out,
                               workspace, workspace_bytes, aug, (hipStream_t)stream);
}

extern "C" int sir_model_set_weights_version(sir_handle* h, uint64_t version) {
    if (!h) { sir_set_error("sir_model_set_weights_version: NULL handle"); return SIR_EINVAL; }
    h->weights_version = version;
    return SIR_OK;
}

// ---- cross-batch pipelining ---------------------------------------------------------------------
struct sir_pipeline {
    sir_handle* h;
    int n;
    unsigned long long seq;
    hipStream_t st[4];
    hipEvent_t ready[4], done[4];
    bool open[4], used[4];
    hipStream_t caller[4];
};

extern "C" int sir_pipeline_create(sir_handle* h, int n_slots, sir_pipeline** out) {
    if (!h || !out || n_slots < 1 || n_slots > 4) { sir_set_error("sir_pipeline_create: bad argument (1 <= n_slots <= 4)"); return SIR_EINVAL; }
    sir_pipeline* p = new sir_pipeline();
    p->h = h; p->n = n_slots; p->seq = 0;
    for (int i = 0; i < 4; ++i) { p->st[i] = nullptr; p->ready[i] = p->done[i] = nullptr; p->open[i] = p->used[i] = false; p->caller[i] = nullptr; }
    if (n_slots > 1)
        for (int i = 0; i < n_slots; ++i) {
            if (hipStreamCreateWithFlags(&p->st[i], hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&p->ready[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&p->done[i], hipEventDisableTiming) != hipSuccess) {
                sir_set_error("sir_pipeline_create: stream / event creation failed");
                sir_pipeline_destroy(p);
                return SIR_EHIP;
            }
        }
    *out = p;
    return SIR_OK;
}

extern "C" int sir_pipeline_destroy(sir_pipeline* p) {
    if (!p) return SIR_OK;
    for (int i = 0; i < 4; ++i) {
        if (p->st[i]) { (void)hipStreamSynchronize(p->st[i]); (void)hipStreamDestroy(p->st[i]); }
        if (p->ready[i]) (void)hipEventDestroy(p->ready[i]);
        if (p->done[i]) (void)hipEventDestroy(p->done[i]);
    }
    delete p;
    return SIR_OK;
}

extern "C" int sir_pipeline_begin(sir_pipeline* p, void* caller_stream, int* slot, void** slot_stream) {
    if (!p || !slot || !slot_stream) { sir_set_error("sir_pipeline_begin: NULL argument"); return SIR_EINVAL; }
    const int k = (int)(p->seq % (unsigned long long)p->n);
    if (p->open[k]) { sir_set_error("sir_pipeline_begin: slot %d is still open (sir_pipeline_end missing)", k); return SIR_EINVAL; }
    hipStream_t cs = (hipStream_t)caller_stream;
    if (p->n == 1) {
        *slot_stream = caller_stream;
    } else {
        SIR_HIP_TRY(hipEventRecord(p->ready[k], cs));            // inputs (and earlier reads of this slot's outputs) on the caller's stream
        SIR_HIP_TRY(hipStreamWaitEvent(p->st[k], p->ready[k], 0));
        *slot_stream = (void*)p->st[k];
    }
    p->open[k] = true; p->caller[k] = cs;
    *slot = k;
    ++p->seq;
    return SIR_OK;
}

extern "C" int sir_pipeline_end(sir_pipeline* p, int slot) {
    if (!p || slot < 0 || slot >= p->n || !p->open[slot]) { sir_set_error("sir_pipeline_end: slot %d is not open", slot); return SIR_EINVAL; }
    if (p->n > 1) SIR_HIP_TRY(hipEventRecord(p->done[slot], p->st[slot]));
    p->open[slot] = false; p->used[slot] = true;
    return SIR_OK;
}

extern "C" int sir_pipeline_join(sir_pipeline* p, void* caller_stream) {
    if (!p) { sir_set_error("sir_pipeline_join: NULL pipeline"); return SIR_EINVAL; }
    if (p->n > 1)
        for (int k = 0; k < p->n; ++k)
            if (p->used[k]) SIR_HIP_TRY(hipStreamWaitEvent((hipStream_t)caller_stream, p->done[k], 0));
    return SIR_OK;
}

// ---- event profiling ------------------------------------------------------------------------
static const char* const kKernelNames[SIR_K_COUNT] = {
    "feat_frames", "feat_normalise", "weight_prep", "conv1_bn_relu_pool", "conv2_mfma_bn_relu_pool",
    "conv3_mfma_bn_relu_pool", "gemm_ih_l0", "gru_recurrence_l0", "gemm_ih_l1", "gru_recurrence_l1",
    "attention_pool_fc_argmax", "unused",
    "train_weight_prep", "train_conv1_fwd", "train_conv2_fwd", "train_bn2_relu_pool", "train_conv3_fwd", "train_bn3_relu_pool",
    "train_gemm_ih_l0", "train_gru_l0", "train_dropout", "train_gemm_ih_l1", "train_gru_l1", "train_attention_fc", "ce_loss",
    "bwd_head", "bwd_gru_l1", "bwd_gru_dw_l1", "bwd_gru_dx_l1", "bwd_gru_l0", "bwd_gru_dw_l0", "bwd_gru_dx_l0",
    "bwd_bn3", "bwd_conv3_wgrad", "bwd_conv3_dgrad", "bwd_bn2", "bwd_conv2_wgrad", "bwd_conv2_dgrad", "bwd_conv1", "adam"};

extern "C" int sir_profile_kernel_count(void) { return SIR_K_COUNT; }
extern "C" const char* sir_profile_kernel_name(int id) { return (id >= 0 && id < SIR_K_COUNT) ? kKernelNames[id] : ""; }

extern "C" int sir_profile_enable(sir_handle* h, int mode, int kernel_id) {
    if (!h || mode < 0 || mode > 2) { sir_set_error("sir_profile_enable: bad argument"); return SIR_EINVAL; }
    h->prof_mode = mode;
    h->prof_only = kernel_id;
    return SIR_OK;
}

// Reads and clears the device status word behind everything queued on `stream` (host-synchronous on that stream).
static int check_status_impl(sir_handle* h, hipStream_t st, const char* who) {
    unsigned int v = 0;
    SIR_HIP_TRY(hipMemcpyAsync(&v, h->status, sizeof(v), hipMemcpyDeviceToHost, st));
    SIR_HIP_TRY(hipStreamSynchronize(st));
    if (v == 0) return SIR_OK;
    SIR_HIP_TRY(hipMemsetAsync(h->status, 0, sizeof(v), st));
    SIR_HIP_TRY(hipStreamSynchronize(st));
    if (v & 1u) {
        sir_set_error("%s: a GRU recurrence kernel timed out waiting for a peer workgroup of its cluster (status %u): the "
                      "logits / gradients produced since the last check are invalid", who, v);
        return SIR_ETIMEOUT;
    }
    if (v & 8u) {
        sir_set_error("%s: a transformed convolution weight is outside the f16x3 path's range (|U| >= 32, status %u): the outputs "
                      "computed from the clamped value are invalid", who, v);
        return SIR_EINVAL;
    }
    if (v & 4u) {
        sir_set_error("%s: sir_gather_features was given an index outside its store (status %u): those rows are zero", who, v);
        return SIR_EINVAL;
    }
    sir_set_error("%s: sir_ce_loss saw a label outside [0, num_classes) (status %u): that step's loss is NaN and its "
                  "gradients are invalid (nn.CrossEntropyLoss raises on such a target)", who, v);
    return SIR_EINVAL;
}

extern "C" int sir_check_status(sir_handle* h, void* stream) {
    if (!h) { sir_set_error("sir_check_status: NULL handle"); return SIR_EINVAL; }
    return check_status_impl(h, (hipStream_t)stream, "sir_check_status");
}

extern "C" int sir_profile_collect(sir_handle* h, double* total_ms, int64_t* launches, int n) {
    if (!h || !total_ms || !launches) { sir_set_error("sir_profile_collect: NULL argument"); return SIR_EINVAL; }
    for (int i = 0; i < n; ++i) { total_ms[i] = 0.0; launches[i] = 0; }
    for (auto& r : h->prof_pending) {
        SIR_HIP_TRY(hipEventSynchronize(r.e1));
        float ms = 0.f;
        SIR_HIP_TRY(hipEventElapsedTime(&ms, r.e0, r.e1));
        if (r.id < n) { total_ms[r.id] += ms; launches[r.id] += 1; }
        h->prof_free.push_back(r.e0);
        h->prof_free.push_back(r.e1);
    }
    h->prof_pending.clear();
    return check_status_impl(h, nullptr, "sir_profile_collect");   // the host has synchronised anyway: surface a timed-out recurrence
}
