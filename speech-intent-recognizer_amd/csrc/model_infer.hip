// sir_model_infer: eval-mode CNNAudioGRU.forward + argmax (models/models.py:41-68, scripts/evaluate.py:82-83)
// as a fixed sequence of hand-written kernels on one stream.  See model_kernels.h for the kernels.
#include <stdlib.h>
#include "bf16x6_kernels.h"

namespace {

enum WsBuf {
    WS_A1 = 0,   // conv1 out  NHWC [B][32][T/2][32]
    WS_A2,       // conv2 out  NHWC [B][16][T/4][64]
    WS_X0,       // conv3 out = GRU input [B][S][1024], feature = c*8 + h
    WS_GI,       // input projections of the current GRU layer [B*S][1536]
    WS_Y0,       // GRU layer 0 output [B][S][512]
    WS_Y1,       // GRU layer 1 output [B][S][512]
    WS_CTX,      // attention-pooled context [B][512]
    WS_WP2,      // prepared conv2 weights [36][64][8]
    WS_WP3,      // prepared conv3 weights [72][128][8]
    WS_BN,       // folded BN: scale[224] then shift[224] (channels of bn1|bn2|bn3)
    WS_WHT,      // transposed W_hh, [4][64][768][4]
    WS_XS,       // bf16x3 planes of the current GEMM A operand, [3][B*S][1024] bf16
    WS_WS,       // bf16x3 planes of W_ih: l0 [2][3][768][1024], l1 [2][3][768][512]
    WS_WCB,      // bf16x3 planes of the conv2 / conv3 weights
    WS_GXB,      // paired GRU: h exchange buffers [pairs*2][2][2][4][128] f32
    WS_GFL,      // paired GRU: flags [pairs*2][2] u32 + status word
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
    bytes[WS_WP2] = (size_t)36 * 64 * 8 * 4;
    bytes[WS_WP3] = (size_t)72 * 128 * 8 * 4;
    bytes[WS_BN] = (size_t)2 * 224 * 4;
    bytes[WS_WHT] = (size_t)4 * 768 * 256 * 6;                  // streaming GRU: W_hh regrouped (fp32); quad GRU: resident bf16x3 fragments
    bytes[WS_XS] = B * d.S * 1024 * 3 * 2;
    bytes[WS_WS] = ((size_t)2 * 3 * 768 * 1024 + (size_t)2 * 3 * 768 * 512) * 2;
    bytes[WS_WCB] = ((size_t)3 * 32 * 9 * 64 + (size_t)3 * 64 * 9 * 128) * 2;
    bytes[WS_GXB] = sir_gru_pair_xbuf_bytes(d.B);
    bytes[WS_GFL] = sir_gru_pair_flag_bytes(d.B);
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

int sir_gru_variant() {      // 0 = streaming fp32 kernel, 1 = paired fp32 kernel, 2 = quad MFMA kernel
    static const int v = getenv("SIR_GRU_VARIANT") ? atoi(getenv("SIR_GRU_VARIANT")) : 2;
    return v;
}

int sir_conv1_mfma() {
    static const int v = getenv("SIR_CONV1_MFMA") ? atoi(getenv("SIR_CONV1_MFMA")) : 1;
    return v;
}

int sir_conv_ns() {
    static const int v = getenv("SIR_CONV_NS") ? atoi(getenv("SIR_CONV_NS")) : 1;
    return v;
}

int sir_gemm_bf16x6_gen() {
    static const int gen = getenv("SIR_GEMM_BF16X6_GEN") ? atoi(getenv("SIR_GEMM_BF16X6_GEN")) : 2;
    return gen;
}

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
    float* wp2 = (float*)(ws + off[WS_WP2]);
    float* wp3 = (float*)(ws + off[WS_WP3]);
    float* bns = (float*)(ws + off[WS_BN]);
    float* bnt = bns + 224;
    float* wht = (float*)(ws + off[WS_WHT]);
    unsigned short* xs = (unsigned short*)(ws + off[WS_XS]);
    unsigned short* wsl0 = (unsigned short*)(ws + off[WS_WS]);
    unsigned short* wsl1 = wsl0 + (size_t)2 * 3 * 768 * 1024;
    unsigned short* wcb2 = (unsigned short*)(ws + off[WS_WCB]);
    unsigned short* wcb3 = wcb2 + (size_t)3 * 32 * 9 * 64;
    float* gxb = (float*)(ws + off[WS_GXB]);
    unsigned int* gfl = (unsigned int*)(ws + off[WS_GFL]);
    const int B = d.B, S = d.S;

    // GEMM variant: 2 = bf16x6 split on the bf16 matrix cores (default), 1/0 = fp32 MFMA (hoisted / plain)
    static const int gemm_variant = getenv("SIR_GEMM_VARIANT") ? atoi(getenv("SIR_GEMM_VARIANT")) : 2;
    static const int conv_bf16 = getenv("SIR_CONV_BF16X6") ? atoi(getenv("SIR_CONV_BF16X6")) : 1;
    // GRU variant: 1 = paired workgroups with W_hh resident on chip (default), 0 = single workgroup streaming W_hh
    const int gru_variant = sir_gru_variant();
    static const int occ = getenv("SIR_CONV_OCC") ? atoi(getenv("SIR_CONV_OCC")) : 1;   // A/B: 3-workgroup-per-CU conv configurations
    // producers write the bf16x3 planes of the next GEMM's A operand themselves (conv3 -> projection 0, GRU layer 0 ->
    // projection 1) instead of a separate split pass over the fp32 activations
    static const int fuse_env = getenv("SIR_FUSE_SPLIT") ? atoi(getenv("SIR_FUSE_SPLIT")) : 1;
    const bool fuse_x0 = fuse_env && conv_bf16 && occ && sir_conv_ns() && gemm_variant == 2;
    const bool fuse_y0 = fuse_env && gru_variant == 2 && gemm_variant == 2;
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
    if (!conv_bf16) {
        hipLaunchKernelGGL(prep_conv_w_kernel, dim3((32 * 9 * 64 + 255) / 256), dim3(256), 0, st, w->conv_w[1], wp2, 32, 64);
        hipLaunchKernelGGL(prep_conv_w_kernel, dim3((64 * 9 * 128 + 255) / 256), dim3(256), 0, st, w->conv_w[2], wp3, 64, 128);
    }
    const int bn_c[3] = {32, 64, 128}, bn_o[3] = {0, 32, 96};
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL(prep_bn_kernel, dim3(1), dim3(128), 0, st, w->bn_w[i], w->bn_b[i], w->bn_mean[i], w->bn_var[i],
                           bns + bn_o[i], bnt + bn_o[i], bn_c[i]);
    if (gru_variant == 0)
        for (int i = 0; i < 4; ++i)
            hipLaunchKernelGGL(prep_whh_kernel, dim3(768), dim3(256), 0, st, w->gru_w_hh[i], wht + (size_t)i * 768 * 256);
    if (gru_variant == 2)
        for (int i = 0; i < 4; ++i) sir_prep_whh_quad(st, w->gru_w_hh[i], (unsigned char*)wht + (size_t)i * 768 * 256 * 6);
    if (conv_bf16) {
        hipLaunchKernelGGL(prep_conv_w_bf16x3_kernel, dim3((32 * 9 * 64 + 255) / 256), dim3(256), 0, st, w->conv_w[1], wcb2, 32, 64);
        hipLaunchKernelGGL(prep_conv_w_bf16x3_kernel, dim3((64 * 9 * 128 + 255) / 256), dim3(256), 0, st, w->conv_w[2], wcb3, 64, 128);
    }
    if (gemm_variant == 2) {
        for (int dir = 0; dir < 2; ++dir) {
            hipLaunchKernelGGL(split3_kernel, dim3(384), dim3(256), 0, st, w->gru_w_ih[dir], 1024, wsl0 + (size_t)dir * 3 * 768 * 1024, (size_t)768, 1024);
            hipLaunchKernelGGL(split3_kernel, dim3(192), dim3(256), 0, st, w->gru_w_ih[2 + dir], 512, wsl1 + (size_t)dir * 3 * 768 * 512, (size_t)768, 512);
        }
    }
    if (!pe) { pe = &h->prep[h->prep_next]; h->prep_next = (h->prep_next + 1) % 4; }
    pe->ws = workspace; pe->version = h->weights_version; pe->key = prep_key;
    }
    SIR_KCHECK();

    // ---- CNN stack ------------------------------------------------------------------------
    { SirProfScope prof(h, SIR_K_CONV1, st);
    hipLaunchKernelGGL(sir_conv1_mfma() ? conv1_mfma_bn_relu_pool_kernel : conv1_bn_relu_pool_kernel,
                       dim3((d.wp1 + C1_PCOLS - 1) / C1_PCOLS, sir_conv1_mfma() ? 1 : (32 + C1_PROWS - 1) / C1_PROWS, B),
                       dim3(256), 0, st, feats, w->conv_w[0], bns, bnt, a1, 64, d.T, 32, d.wp1); }
    static const int conv2_variant = getenv("SIR_CONV2_VARIANT") ? atoi(getenv("SIR_CONV2_VARIANT")) : 0;   // A/B switch
    {
        SirProfScope prof(h, SIR_K_CONV2, st);
        if (conv_bf16) {
            constexpr size_t lds = conv_bf16x6_lds_bytes(4, 2);
            if (sir_conv_ns())
if (occ)
            hipLaunchKernelGGL((conv3x3_bf16x6_ns_kernel<32, 64, 4, 2, 0, 0, 3>), dim3((d.wp1 + 7) / 8, 1, B), dim3(256), conv_ns_lds_bytes(4, 2), st, a1,
                               (const unsigned short*)wcb2, bns + 32, bnt + 32, a2, 32, d.wp1, 16, d.wp2, (float2*)nullptr);
            else
            hipLaunchKernelGGL((conv3x3_bf16x6_ns_kernel<32, 64, 4, 2, 0>), dim3((d.wp1 + 7) / 8, 1, B), dim3(256), conv_ns_lds_bytes(4, 2), st, a1,
                               (const unsigned short*)wcb2, bns + 32, bnt + 32, a2, 32, d.wp1, 16, d.wp2, (float2*)nullptr);
            else
            hipLaunchKernelGGL((conv3x3_bf16x6_kernel<32, 64, 4, 2, 0, 2>), dim3((d.wp1 + 7) / 8, 1, B), dim3(256), lds, st, a1,
                               (const unsigned short*)wcb2, bns + 32, bnt + 32, a2, 32, d.wp1, 16, d.wp2, (float2*)nullptr);
        } else if (conv2_variant == 0) {
            constexpr size_t lds = (size_t)(8 * 4 + 2) * (4 * 2 + 2) * 36 * 4;
            hipLaunchKernelGGL((conv3x3_mfma_kernel<32, 64, 4, 2, 0, 2>), dim3((d.wp1 + 7) / 8, 1, B), dim3(256), lds, st, a1, wp2,
                               bns + 32, bnt + 32, a2, 32, d.wp1, 16, d.wp2, (float2*)nullptr);
        } else if (conv2_variant == 1) {
            constexpr size_t lds = (size_t)(8 * 4 + 2) * (4 * 2 + 2) * 20 * 4;
            hipLaunchKernelGGL((conv3x3_mfma_kernel<32, 64, 4, 2, 0, 2, 16>), dim3((d.wp1 + 7) / 8, 1, B), dim3(256), lds, st, a1, wp2,
                               bns + 32, bnt + 32, a2, 32, d.wp1, 16, d.wp2, (float2*)nullptr);
        } else {
            constexpr size_t lds = (size_t)(8 * 2 + 2) * (4 * 4 + 2) * 36 * 4;
            hipLaunchKernelGGL((conv3x3_mfma_kernel<32, 64, 2, 4, 0, 2>), dim3((d.wp1 + 15) / 16, 2, B), dim3(256), lds, st, a1, wp2,
                               bns + 32, bnt + 32, a2, 32, d.wp1, 16, d.wp2, (float2*)nullptr);
        }
    }
    {
        SirProfScope prof(h, SIR_K_CONV3, st);
        static const int conv3_variant = getenv("SIR_CONV3_VARIANT") ? atoi(getenv("SIR_CONV3_VARIANT")) : 0;
        if (conv_bf16 && conv3_variant == 0) {
        constexpr size_t lds = conv_bf16x6_lds_bytes(2, 4);
        if (sir_conv_ns())
if (occ)
        hipLaunchKernelGGL((conv3x3_bf16x6_ns_kernel<64, 128, 2, 2, 1, 0, 3>), dim3((d.wp2 + 7) / 8, 1, B), dim3(256), conv_ns_lds_bytes(2, 2), st, a2,
                           (const unsigned short*)wcb3, bns + 96, bnt + 96, x0, 16, d.wp2, 8, d.wp3, fuse_x0 ? (float2*)xs : (float2*)nullptr);
        else
        hipLaunchKernelGGL((conv3x3_bf16x6_ns_kernel<64, 128, 2, 4, 1>), dim3((d.wp2 + 15) / 16, 1, B), dim3(256), conv_ns_lds_bytes(2, 4), st, a2,
                           (const unsigned short*)wcb3, bns + 96, bnt + 96, x0, 16, d.wp2, 8, d.wp3, (float2*)nullptr);
        else
        hipLaunchKernelGGL((conv3x3_bf16x6_kernel<64, 128, 2, 4, 1, 2>), dim3((d.wp2 + 15) / 16, 1, B), dim3(256), lds, st, a2,
                           (const unsigned short*)wcb3, bns + 96, bnt + 96, x0, 16, d.wp2, 8, d.wp3, (float2*)nullptr);
        } else if (conv_bf16) {
        constexpr size_t lds = conv_bf16x6_lds_bytes(2, 2);
        hipLaunchKernelGGL((conv3x3_bf16x6_kernel<64, 128, 2, 2, 1, 1>), dim3((d.wp2 + 7) / 8, 1, B), dim3(256), lds, st, a2,
                           (const unsigned short*)wcb3, bns + 96, bnt + 96, x0, 16, d.wp2, 8, d.wp3, (float2*)nullptr);
        } else if (conv3_variant == 0) {
        constexpr size_t lds = (size_t)(8 * 2 + 2) * (4 * 2 + 2) * 36 * 4;
        hipLaunchKernelGGL((conv3x3_mfma_kernel<64, 128, 2, 2, 1, 1>), dim3((d.wp2 + 7) / 8, 1, B), dim3(256), lds, st, a2, wp3,
                           bns + 96, bnt + 96, x0, 16, d.wp2, 8, d.wp3, (float2*)nullptr);
        } else {
        constexpr size_t lds = (size_t)(8 * 2 + 2) * (4 * 2 + 2) * 20 * 4;
        hipLaunchKernelGGL((conv3x3_mfma_kernel<64, 128, 2, 2, 1, 1, 16>), dim3((d.wp2 + 7) / 8, 1, B), dim3(256), lds, st, a2, wp3,
                           bns + 96, bnt + 96, x0, 16, d.wp2, 8, d.wp3, (float2*)nullptr);
        }
    }
    SIR_KCHECK();

    // ---- 2-layer bidirectional GRU ----------------------------------------------------------
    const int M = B * S;
    const dim3 ggrid(768 / GB_N, (M + GB_M - 1) / GB_M, 2);
    const dim3 rgrid((B + GRU_BW - 1) / GRU_BW, 2);
    static bool gru_attr = false;
    if (!gru_attr) {
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)gru_recurrence_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GRU_LDS_BYTES));
        gru_attr = true;
    }
    { SirProfScope prof(h, SIR_K_GEMM_IH0, st);
    if (gemm_variant == 2) {
        if (!fuse_x0) hipLaunchKernelGGL(split3_kernel, dim3(2048), dim3(256), 0, st, (const float*)x0, 1024, xs, (size_t)M, 1024);
        SIR_HIP_TRY(launch_gemm_nt_bf16x6(st, sir_gemm_bf16x6_gen(), (const unsigned short*)xs, (const unsigned short*)wsl0,
                           (const unsigned short*)(wsl0 + (size_t)3 * 768 * 1024), w->gru_b_ih[0], w->gru_b_ih[1], gi, 1536, M, 768, 1024));
    } else if (gemm_variant == 1)
    hipLaunchKernelGGL((gemm_nt_bias_kernel<32, true>), ggrid, dim3(256), 0, st, x0, 1024, w->gru_w_ih[0], w->gru_w_ih[1], 1024,
                       w->gru_b_ih[0], w->gru_b_ih[1], gi, 1536, M, 768, 1024);
    else
    hipLaunchKernelGGL((gemm_nt_bias_kernel<32, false>), ggrid, dim3(256), 0, st, x0, 1024, w->gru_w_ih[0], w->gru_w_ih[1], 1024,
                       w->gru_b_ih[0], w->gru_b_ih[1], gi, 1536, M, 768, 1024); }
    { SirProfScope prof(h, SIR_K_GRU0, st);
    if (sir_cluster_enter(h, st) != SIR_OK) return SIR_EHIP;
    if (gru_variant == 2) {
        const int rc = sir_launch_gru_quad(st, false, gi, w->gru_w_hh[0], w->gru_w_hh[1], w->gru_b_hh[0], w->gru_b_hh[1], y0, B, S, nullptr, gxb, h->status,
                                           fuse_y0 ? xs : nullptr, wht, (unsigned char*)wht + (size_t)768 * 256 * 6);
        if (rc != SIR_OK) return rc;
    } else if (gru_variant == 1) {
        const int rc = sir_launch_gru_pair(st, false, gi, w->gru_w_hh[0], w->gru_w_hh[1], w->gru_b_hh[0], w->gru_b_hh[1], y0, B, S, nullptr, gxb, gfl, h->status);
        if (rc != SIR_OK) return rc;
    } else
    hipLaunchKernelGGL(gru_recurrence_kernel<false>, rgrid, dim3(GRU_THREADS), GRU_LDS_BYTES, st, gi, wht, w->gru_b_hh[0], w->gru_b_hh[1], y0, B, S,
                       (float*)nullptr);
    if (sir_cluster_leave(h, st) != SIR_OK) return SIR_EHIP; }
    { SirProfScope prof(h, SIR_K_GEMM_IH1, st);
    if (gemm_variant == 2) {
        if (!fuse_y0) hipLaunchKernelGGL(split3_kernel, dim3(2048), dim3(256), 0, st, (const float*)y0, 512, xs, (size_t)M, 512);
        SIR_HIP_TRY(launch_gemm_nt_bf16x6(st, sir_gemm_bf16x6_gen(), (const unsigned short*)xs, (const unsigned short*)wsl1,
                           (const unsigned short*)(wsl1 + (size_t)3 * 768 * 512), w->gru_b_ih[2], w->gru_b_ih[3], gi, 1536, M, 768, 512));
    } else
    hipLaunchKernelGGL((gemm_nt_bias_kernel<32, true>), ggrid, dim3(256), 0, st, y0, 512, w->gru_w_ih[2], w->gru_w_ih[3], 512,
                       w->gru_b_ih[2], w->gru_b_ih[3], gi, 1536, M, 768, 512); }
    { SirProfScope prof(h, SIR_K_GRU1, st);
    if (sir_cluster_enter(h, st) != SIR_OK) return SIR_EHIP;
    if (gru_variant == 2) {
        const int rc = sir_launch_gru_quad(st, false, gi, w->gru_w_hh[2], w->gru_w_hh[3], w->gru_b_hh[2], w->gru_b_hh[3], y1, B, S, nullptr, gxb, h->status,
                                           nullptr, (unsigned char*)wht + (size_t)2 * 768 * 256 * 6, (unsigned char*)wht + (size_t)3 * 768 * 256 * 6);
        if (rc != SIR_OK) return rc;
    } else if (gru_variant == 1) {
        const int rc = sir_launch_gru_pair(st, false, gi, w->gru_w_hh[2], w->gru_w_hh[3], w->gru_b_hh[2], w->gru_b_hh[3], y1, B, S, nullptr, gxb, gfl, h->status);
        if (rc != SIR_OK) return rc;
    } else
    hipLaunchKernelGGL(gru_recurrence_kernel<false>, rgrid, dim3(GRU_THREADS), GRU_LDS_BYTES, st, gi, wht + (size_t)2 * 768 * 256, w->gru_b_hh[2],
                       w->gru_b_hh[3], y1, B, S, (float*)nullptr);
    if (sir_cluster_leave(h, st) != SIR_OK) return SIR_EHIP; }
    SIR_KCHECK();

    // ---- attention pooling + classifier head ------------------------------------------------
    { SirProfScope prof(h, SIR_K_ATTN, st);
    hipLaunchKernelGGL(attention_pool_kernel, dim3(B), dim3(256), 0, st, y1, w->attn_w, w->attn_b, ctx, S, w->fc_w, w->fc_b,
                       w->num_classes, logits, (long long*)argmax); }
    SIR_KCHECK();
    return SIR_OK;
}
