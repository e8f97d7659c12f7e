// sir_model_infer: eval-mode CNNAudioGRU.forward + argmax (models/models.py:41-68, scripts/evaluate.py:82-83)
// as a fixed sequence of hand-written kernels on one stream.  See model_kernels.h for the kernels.
#include "bf16x6_kernels.h"
#include "f16x3_kernels.h"
#include "conv_wino_bf16x6_kernel.h"
#include "conv_wino2_bf16x6_kernel.h"
#include "gru_frag_prep.h"

namespace {

enum WsBuf {
    WS_A1 = 0,   // conv1 out  NHWC [B][32][T/2][32]
    WS_A2,       // conv2 out  NHWC [B][16][T/4][64]
    WS_X0,       // conv3 out = GRU input [B][S][1024], feature = c*8 + h
    WS_GI,       // input projections of the current GRU layer [B*S][1536]
    WS_Y0,       // GRU layer 0 output [B][S][512]
    WS_Y1,       // GRU layer 1 output [B][S][512]
    WS_CTX,      // attention-pooled context [B][512]
    WS_WP2,      // (unused, kept so that the buffer indices of sir_model_workspace_offsets stay put)
    WS_WP3,      // (unused)
    WS_BN,       // folded BN: scale[224] then shift[224] (channels of bn1|bn2|bn3)
    WS_WHT,      // W_hh fragments of the recurrence kernel, [4 (layer, direction)][GRU_FRAG_BYTES]
    WS_XS,       // f16x2 planes (f16_split.h) of the current GEMM A operand, [2][B*S][1024] fp16
    WS_WS,       // f16x2 planes of W_ih: l0 [2 directions][2][768][1024], l1 [2][2][768][512]
    WS_WCB,      // bf16x3 planes of the conv2 / conv3 weights
    WS_GXB,      // (unused: the exchange granules of the GRU clusters live in handle-owned buffers, sir_xbuf_acquire)
    WS_GFL,      // (unused)
    WS_COUNT
};

struct Dims {
    int B, T, wp1, wp2, wp3, S;
};

bool make_dims(int batch, int t_frames, Dims* d) {
    d->B = batch; d->T = t_frames;
    d->wp1 = t_frames / 2; d->wp2 = d->wp1 / 2; d->wp3 = d->wp2 / 2; d->S = d->wp3;
    return batch > 0 && d->S >= 1 && d->S <= ATT_MAX_S && batch <= 65535;
}

void ws_sizes(const Dims& d, size_t* bytes) {
    const size_t B = d.B;
    bytes[WS_A1] = B * 32 * d.wp1 * 32 * 4;
    bytes[WS_A2] = B * 16 * d.wp2 * 64 * 4;
    bytes[WS_X0] = B * d.S * 1024 * 4;
    bytes[WS_GI] = B * d.S * 1536 * 4;
    bytes[WS_Y0] = B * d.S * 512 * 4;
    bytes[WS_Y1] = B * d.S * 512 * 4;
    bytes[WS_CTX] = B * 512 * 4;
    bytes[WS_WP2] = 0;
    bytes[WS_WP3] = 0;
    bytes[WS_BN] = (size_t)2 * 224 * 4;
    bytes[WS_WHT] = 4 * GRU_FRAG_BYTES;                         // W_hh as the resident f16x2 MFMA fragments of the recurrence kernel
    bytes[WS_XS] = B * d.S * 1024 * 2 * 2;
    bytes[WS_WS] = ((size_t)2 * 2 * 768 * 1024 + (size_t)2 * 2 * 768 * 512) * 2;
    bytes[WS_WCB] = ((size_t)3 * 32 * 16 * 64 + (size_t)3 * 64 * 16 * 128 + (size_t)3 * 64 * 9 * 128) * 2;   // conv2, conv3: 16 Winograd frequencies per (cout, cin); conv3 again with 9 taps for the direct kernel (shapes the Winograd kernel does not cover)
    bytes[WS_GXB] = 0;
    bytes[WS_GFL] = 0;
}

size_t ws_layout(const Dims& d, size_t* off) {
    size_t bytes[WS_COUNT], pos = 0;
    ws_sizes(d, bytes);
    for (int i = 0; i < WS_COUNT; ++i) {
        off[i] = pos;
        pos += sir_align_up(bytes[i], 256);
    }
    return pos;
}

}  // namespace

extern "C" size_t sir_model_workspace_bytes(const sir_handle* h, int batch, int t_frames, int train) {
    (void)h;
    if (train) return sir_train_workspace_bytes_impl(batch, t_frames);
    Dims d;
    if (!make_dims(batch, t_frames, &d)) return 0;
    size_t off[WS_COUNT];
    return ws_layout(d, off);
}

extern "C" int sir_model_workspace_offsets(const sir_handle* h, int batch, int t_frames, int train, size_t* offsets,
                                           int n) {
    (void)h; (void)train;
    Dims d;
    if (!make_dims(batch, t_frames, &d) || !offsets) { sir_set_error("sir_model_workspace_offsets: bad shape"); return SIR_EINVAL; }
    size_t off[WS_COUNT];
    ws_layout(d, off);
    for (int i = 0; i < n && i < WS_COUNT; ++i) offsets[i] = off[i];
    return WS_COUNT;
}

#define SIR_KCHECK() SIR_HIP_TRY(hipGetLastError())

extern "C" int sir_model_infer(sir_handle* h, const sir_model_weights* w, const float* feats, int batch, int t_frames,
                               float* logits, int64_t* argmax, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!h || !w || !feats || !logits || !workspace) { sir_set_error("sir_model_infer: NULL argument"); return SIR_EINVAL; }
    Dims d;
    if (!make_dims(batch, t_frames, &d)) {
        sir_set_error("sir_model_infer: unsupported shape batch=%d t_frames=%d (need t_frames >= 8)", batch, t_frames);
        return SIR_EINVAL;
    }
    if (h->cfg.n_mels != 64) { sir_set_error("sir_model_infer: the model is wired for 64 mels (models.py:23)"); return SIR_EUNSUPPORTED; }
    if (w->num_classes < 1 || w->num_classes > 64) { sir_set_error("sir_model_infer: num_classes=%d", w->num_classes); return SIR_EINVAL; }
    size_t off[WS_COUNT];
    const size_t need = ws_layout(d, off);
    if (workspace_bytes < need) { sir_set_error("sir_model_infer: workspace %zu < %zu", workspace_bytes, need); return SIR_ENOMEM; }
    if (((uintptr_t)workspace & 255) != 0) { sir_set_error("sir_model_infer: workspace must be 256-byte aligned"); return SIR_EINVAL; }
    hipStream_t st = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    float* a1 = (float*)(ws + off[WS_A1]);
    float* a2 = (float*)(ws + off[WS_A2]);
    float* x0 = (float*)(ws + off[WS_X0]);
    float* gi = (float*)(ws + off[WS_GI]);
    float* y0 = (float*)(ws + off[WS_Y0]);
    float* y1 = (float*)(ws + off[WS_Y1]);
    float* ctx = (float*)(ws + off[WS_CTX]);
    float* bns = (float*)(ws + off[WS_BN]);
    float* bnt = bns + 224;
    float* wht = (float*)(ws + off[WS_WHT]);
    unsigned short* xs = (unsigned short*)(ws + off[WS_XS]);
    unsigned short* wsl0 = (unsigned short*)(ws + off[WS_WS]);
    unsigned short* wsl1 = wsl0 + (size_t)2 * 2 * 768 * 1024;
    unsigned short* wcb2 = (unsigned short*)(ws + off[WS_WCB]);
    unsigned short* wcb3 = wcb2 + (size_t)3 * 32 * 16 * 64;       // Winograd form
    unsigned short* wcb3d = wcb3 + (size_t)3 * 64 * 16 * 128;     // direct form (fallback)
    const int B = d.B, S = d.S;

    // conv2 / conv3 as Winograd F(2x2, 3x3) -- the 2x2 output tile is the pooling window -- on the producer / consumer kernel
    // (conv_wino2_bf16x6_kernel.h); shapes it does not cover (batch x map beyond 32-bit offsets) keep the first-generation kernels
    Wino2Geo geo2, geo3;
    const bool w2ok = wino2_geo(B, 32, d.wp1, 64, &geo2) && wino2_geo(B, 16, d.wp2, 128, &geo3);
    const bool w2c2 = w2ok && (sir_wino2_mask() & 1), w2c3 = w2ok && (sir_wino2_mask() & 2);
    const bool f16c2 = w2c2 && (sir_f16_mask() & 1), f16c3 = w2c3 && (sir_f16_mask() & 2);

    // ---- weight preparation -------------------------------------------------------------
    // skipped when the caller vouches (sir_model_set_weights_version) that the weights are the ones prepared
    // into this very workspace by the previous call
    const long long prep_key = ((long long)B << 32) | (unsigned)d.T;
    sir_handle::PrepEntry* pe = nullptr;
    for (auto& e : h->prep)
        if (e.ws == workspace) pe = &e;
    const bool reuse_prep = pe && h->weights_version != 0 && pe->version == h->weights_version && pe->key == prep_key;
    if (!reuse_prep) {
        SirProfScope prof(h, SIR_K_PREP, st);
        const int bn_c[3] = {32, 64, 128}, bn_o[3] = {0, 32, 96};
        for (int i = 0; i < 3; ++i)
            hipLaunchKernelGGL(prep_bn_kernel, dim3(1), dim3(128), 0, st, w->bn_w[i], w->bn_b[i], w->bn_mean[i], w->bn_var[i],
                               bns + bn_o[i], bnt + bn_o[i], bn_c[i]);
        for (int i = 0; i < 4; ++i) sir_prep_whh_quad(st, w->gru_w_hh[i], (unsigned char*)wht + (size_t)i * GRU_FRAG_BYTES);
        // (the weight planes are prepared in the arithmetic of the kernel that will read them: f16x3 for the second-generation
        // Winograd kernel's forward stages, bf16x3 for the first-generation / direct fallbacks)
        if (f16c2) hipLaunchKernelGGL(prep_conv_w_wino_f16x3_kernel, dim3((32 * 16 * 64 + 255) / 256), dim3(256), 0, st, w->conv_w[1], wcb2, 32, 64, h->status);
        else hipLaunchKernelGGL(prep_conv_w_wino_bf16x3_kernel, dim3((32 * 16 * 64 + 255) / 256), dim3(256), 0, st, w->conv_w[1], wcb2, 32, 64);
        if (f16c3) hipLaunchKernelGGL(prep_conv_w_wino_f16x3_kernel, dim3((64 * 16 * 128 + 255) / 256), dim3(256), 0, st, w->conv_w[2], wcb3, 64, 128, h->status);
        else hipLaunchKernelGGL(prep_conv_w_wino_bf16x3_kernel, dim3((64 * 16 * 128 + 255) / 256), dim3(256), 0, st, w->conv_w[2], wcb3, 64, 128);
        hipLaunchKernelGGL(prep_conv_w_bf16x3_kernel, dim3((64 * 9 * 128 + 255) / 256), dim3(256), 0, st, w->conv_w[2], wcb3d, 64, 128);
        for (int dir = 0; dir < 2; ++dir) {
            hipLaunchKernelGGL(split2h_kernel, dim3(384), dim3(256), 0, st, w->gru_w_ih[dir], 1024, wsl0 + (size_t)dir * 2 * 768 * 1024, (size_t)768, 1024);
            hipLaunchKernelGGL(split2h_kernel, dim3(192), dim3(256), 0, st, w->gru_w_ih[2 + dir], 512, wsl1 + (size_t)dir * 2 * 768 * 512, (size_t)768, 512);
        }
        if (!pe) { pe = &h->prep[h->prep_next]; h->prep_next = (h->prep_next + 1) % 4; }
        pe->ws = workspace; pe->version = h->weights_version; pe->key = prep_key;
    }
    SIR_KCHECK();

    // ---- CNN stack: conv + folded BN + ReLU + 2x2 max-pool per launch ------------------------------
    {
        SirProfScope prof(h, SIR_K_CONV1, st);
        hipLaunchKernelGGL(conv1_mfma_bn_relu_pool_kernel, dim3((d.wp1 + C1_PCOLS - 1) / C1_PCOLS, 1, B), dim3(256), 0, st, feats,
                           w->conv_w[0], bns, bnt, a1, 64, d.T, 32, d.wp1);
    }
    {
        SirProfScope prof(h, SIR_K_CONV2, st);
        if (f16c2)
            SIR_HIP_TRY((launch_conv_wino2<32, 64, 0, 0, 3, true>(st, &h->attr_wino2[5], a1, (const unsigned short*)wcb2, bns + 32, bnt + 32, a2, B, 32, d.wp1,
                                                                (float2*)nullptr, h->zero_page, h->num_cus)));
        else if (w2c2)
            SIR_HIP_TRY((launch_conv_wino2<32, 64, 0>(st, &h->attr_wino2[0], a1, (const unsigned short*)wcb2, bns + 32, bnt + 32, a2, B, 32, d.wp1,
                                                    (float2*)nullptr, h->zero_page, h->num_cus)));
        else
            hipLaunchKernelGGL((conv3x3_wino_bf16x6_kernel<32, 64, 0>), dim3(((d.wp1 + 1) / 2 + 1) / 2, 1, B), dim3(256), WINO_LDS_BYTES, st,
                               a1, (const unsigned short*)wcb2, bns + 32, bnt + 32, a2, 32, d.wp1, 16, d.wp2, (float2*)nullptr);
    }
    {
        // conv3 stores straight into the GRU input layout [B][S][c*8+h] (models.py:55-57) and writes the f16x2 planes of
        // the first input projection's A operand beside it
        SirProfScope prof(h, SIR_K_CONV3, st);
        if (f16c3)
            SIR_HIP_TRY((launch_conv_wino2<64, 128, 1, 0, 3, true>(st, &h->attr_wino2[6], a2, (const unsigned short*)wcb3, bns + 96, bnt + 96, x0, B, 16, d.wp2,
                                                                 (float2*)xs, h->zero_page, h->num_cus)));
        else if (w2c3)
            SIR_HIP_TRY((launch_conv_wino2<64, 128, 1>(st, &h->attr_wino2[1], a2, (const unsigned short*)wcb3, bns + 96, bnt + 96, x0, B, 16, d.wp2,
                                                     (float2*)xs, h->zero_page, h->num_cus)));
        else
            hipLaunchKernelGGL((conv3x3_bf16x6_ns_kernel<64, 128, 2, 2, 1, 0, 2, 1, 1>), dim3((d.wp2 + 7) / 8, 1, B), dim3(256), conv_ns_lds_bytes(2, 2, 2), st,
                               a2, (const unsigned short*)wcb3d, bns + 96, bnt + 96, x0, 16, d.wp2, 8, d.wp3, (float2*)xs);
    }
    SIR_KCHECK();

    // ---- 2-layer bidirectional GRU ----------------------------------------------------------
    const int M = B * S;
    {
        SirProfScope prof(h, SIR_K_GEMM_IH0, st);
        SIR_HIP_TRY(launch_gemm_nt_f16x3(h, st, (const unsigned short*)xs, (const unsigned short*)wsl0,
                                         (const unsigned short*)(wsl0 + (size_t)2 * 768 * 1024), w->gru_b_ih[0], w->gru_b_ih[1], gi, 1536, M, 768, 1024));
    }
    {
        SirProfScope prof(h, SIR_K_GRU0, st);
        if (sir_cluster_enter(h, st) != SIR_OK) return SIR_EHIP;
        // layer 0 also writes the f16x2 planes of ITS output: the A operand of the layer-1 projection
        const int rc = sir_launch_gru_quad(h, st, false, gi, w->gru_w_hh[0], w->gru_w_hh[1], w->gru_b_hh[0], w->gru_b_hh[1], y0, B, S, nullptr,
                                           xs, wht, (unsigned char*)wht + GRU_FRAG_BYTES);
        if (rc != SIR_OK) return rc;
        if (sir_cluster_leave(h, st) != SIR_OK) return SIR_EHIP;
    }
    {
        SirProfScope prof(h, SIR_K_GEMM_IH1, st);
        SIR_HIP_TRY(launch_gemm_nt_f16x3(h, st, (const unsigned short*)xs, (const unsigned short*)wsl1,
                                         (const unsigned short*)(wsl1 + (size_t)2 * 768 * 512), w->gru_b_ih[2], w->gru_b_ih[3], gi, 1536, M, 768, 512));
    }
    {
        SirProfScope prof(h, SIR_K_GRU1, st);
        if (sir_cluster_enter(h, st) != SIR_OK) return SIR_EHIP;
        const int rc = sir_launch_gru_quad(h, st, false, gi, w->gru_w_hh[2], w->gru_w_hh[3], w->gru_b_hh[2], w->gru_b_hh[3], y1, B, S, nullptr,
                                           nullptr, (unsigned char*)wht + 2 * GRU_FRAG_BYTES, (unsigned char*)wht + 3 * GRU_FRAG_BYTES);
        if (rc != SIR_OK) return rc;
        if (sir_cluster_leave(h, st) != SIR_OK) return SIR_EHIP;
    }
    SIR_KCHECK();

    // ---- attention pooling + classifier head ------------------------------------------------
    {
        SirProfScope prof(h, SIR_K_ATTN, st);
        hipLaunchKernelGGL(attention_pool_kernel, dim3(B), dim3(256), 0, st, y1, w->attn_w, w->attn_b, ctx, S, w->fc_w, w->fc_b,
                           w->num_classes, logits, (long long*)argmax);
    }
    SIR_KCHECK();
    return SIR_OK;
}
