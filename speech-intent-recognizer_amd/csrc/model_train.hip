// Training step of CNNAudioGRU on MI355X: forward with batch-statistics BatchNorm and saved
// activations (sir_model_train_fwd), cross-entropy (sir_ce_loss), full backward
// (sir_model_train_bwd) and multi-tensor Adam (sir_adam_step).  Replaces the body of
// train_epoch (scripts/train.py:90-107: forward, criterion, loss.backward(), optimizer.step()).
#include <cstdlib>
#include "bf16x6_kernels.h"
#include "f16x3_kernels.h"
#include "train_kernels.h"
#include "conv_wino2_bf16x6_kernel.h"
#include "wgrad_bf16x6_kernel.h"
#include "gemm_tn_bf16x6_kernel.h"
#include "gemm_tn2_bf16x6_kernel.h"
#include "wgrad_wino_bf16x6_kernel.h"

namespace {

enum TrainBuf {
    TB_A1 = 0, TB_Z2, TB_A2, TB_Z3, TB_X0, TB_GI, TB_G0, TB_G1, TB_Y0, TB_Y0D, TB_Y1, TB_CTX,
    TB_BN,        // [4][224]: scale, shift, mean, invstd (bn1|bn2|bn3 channel ranges 0,32,96)
    TB_BNB,       // [2][224]: mean dy, mean dy*xhat (backward)
    TB_STATS,     // float2 partials for BN forward/backward reductions
    TB_WP2, TB_WP3, TB_WHT, TB_WR4, TB_WP2T, TB_WP3T,
    TB_DY1, TB_DY0, TB_DGI, TB_DGH, TB_DX0, TB_DZ3, TB_DA2, TB_DZ2, TB_DA1,
    TB_SMALL,     // daw_part [B][512], dab_part [B], conv1 wgrad partials
    TB_SLAB,      // split-K / wgrad partial slabs
    TB_XS,        // f16x2 planes (f16_split.h) of the forward GEMM A operand [2][B*S][1024]
    TB_WS,        // f16x2 planes of W_ih (l0 [2 directions][2][768][1024], l1 [2][2][768][512])
    TB_WCB,       // bf16x3 conv weights: conv2, conv3 forward, then conv2, conv3 data-gradient forms
    TB_GXB,       // (unused: exchange granules live in handle-owned buffers, sir_xbuf_acquire)
    TB_GFL,       // paired GRU status word
    TB_C1M,       // conv1 input moments: 54 doubles (conv1_moments_kernel), forward -> backward
    TB_DGI1, TB_DGH1,   // gate gradients of GRU layer 1 (TB_DGI / TB_DGH hold layer 0's): layer 1's weight-gradient GEMM may run after layer 0's BPTT
    TB_SLAB2,           // split-K slabs of the GRU weight-gradient GEMMs when they run on the side stream beside the BPTT of the layer below (SIR_BWD_STREAMS=2)
    TB_COUNT
};

// K splits of the four-job bf16x6 weight-gradient launch of one GRU layer (gemm_tn_bf16x6_kernel) and its slab floats
static inline void tn_x6_plan(int tokens, int in_sz, int* tiles, int* kchunk, int* nsplit, size_t* slab_floats) {
    const int bm = (sir_tn2_mask() & 1) ? TN2_BM : TN_BM_DW;
    const int t = 2 * ((768 / bm) * ((in_sz + TN_BN - 1) / TN_BN) + (768 / bm) * 1);
    int ks = 256 / t;
    ks = ks < 1 ? 1 : (ks > 16 ? 16 : ks);
    const int kc = (((tokens + ks - 1) / ks) + TN_BK - 1) / TN_BK * TN_BK;
    const int ns = (tokens + kc - 1) / kc;
    *tiles = t; *kchunk = kc; *nsplit = ns;
    *slab_floats = (size_t)ns * 2 * 768 * ((size_t)in_sz + 256);
}

// the layer-input gradient dX = dG [W; W_reverse] of a layer whose 128-row tiles would leave CUs idle (layer 1: 6400 x 512 = 100
// tiles) runs as TWO K halves on 128-row tiles (wave tile 64 x 64) plus an ordered add, instead of 64-row tiles (wave tile 64 x 32)
static inline bool dx_splitk(int tokens, int in_sz) {
    const int nt = ((tokens + TN2_BM - 1) / TN2_BM) * ((in_sz + TN_BN - 1) / TN_BN);
    return (sir_tn2_mask() & 2) && (sir_tn2_mask() & 8) && nt < 160 && 2 * nt >= 96;
}

struct TDims {
    int B, T, wp1, wp2, wp3, S;
    int c1gx, c1gy;          // conv1 grids (ceil over un-pooled odd columns)
    int c2gx, c3gx, c3fx;      // c3fx: conv3 FORWARD grid (16x8-pixel tiles); c3gx: conv3 data-gradient grid (16x16)
    int c2wx;                  // conv2 FORWARD grid: Winograd blocks of two tile columns (4 pixels)
    int wg2_blocks, wg3_blocks, wg2_rb, wg3_rb;
    int ksplits, kchunk;
};

bool make_tdims(int batch, int t, TDims* d) {
    d->B = batch; d->T = t;
    d->wp1 = t / 2; d->wp2 = d->wp1 / 2; d->wp3 = d->wp2 / 2; d->S = d->wp3;
    if (!(batch > 0 && d->S >= 1 && d->S <= ATT_MAX_S && batch <= 65535)) return false;
    d->c1gx = ((t + 1) / 2 + C1_PCOLS - 1) / C1_PCOLS;
    d->c1gy = (32 + C1_PROWS - 1) / C1_PROWS;
    d->c2gx = (d->wp1 + 7) / 8;
    d->c2wx = ((d->wp1 + 1) / 2 + 1) / 2;
    d->c3gx = (d->wp2 + 15) / 16;
    d->c3fx = (d->wp2 + 7) / 8;
    d->wg2_rb = 32; d->wg3_rb = 16;                    // rows per workgroup of the weight-gradient kernels: one image each (256 workgroups at batch 256)
    d->wg2_blocks = batch * (32 / d->wg2_rb);
    d->wg3_blocks = batch * (16 / d->wg3_rb);
    const int K = batch * d->S;
    d->ksplits = K >= 2048 ? 8 : (K >= 256 ? 2 : 1);
    d->kchunk = ((K + d->ksplits - 1) / d->ksplits + 31) / 32 * 32;
    return true;
}

void tws_sizes(const TDims& d, size_t* n) {           // element counts (floats)
    const size_t B = d.B, S = d.S;
    n[TB_A1] = B * 32 * d.wp1 * 32;
    n[TB_Z2] = B * 32 * d.wp1 * 64;
    n[TB_A2] = B * 16 * d.wp2 * 64;
    n[TB_Z3] = B * 16 * d.wp2 * 128;
    n[TB_X0] = B * S * 1024;
    n[TB_GI] = B * S * 1536;
    n[TB_G0] = B * S * 2048;
    n[TB_G1] = B * S * 2048;
    n[TB_Y0] = B * S * 512;
    n[TB_Y0D] = B * S * 512;
    n[TB_Y1] = B * S * 512;
    n[TB_CTX] = B * 512;
    n[TB_BN] = 4 * 224;
    n[TB_BNB] = 2 * 224;
    size_t st = (size_t)d.c1gx * d.c1gy * B * 32;                       // conv1 partials (float2)
    // per-(task, tile column) statistics of the producer / consumer Winograd kernel (or per-workgroup ones of the fallback kernels)
    size_t s2 = (size_t)d.c2wx * B * 64, s3 = (size_t)d.c3fx * B * 128;
    if ((size_t)4 * 1024 * 64 > s2) s2 = (size_t)4 * 1024 * 64;         // (4 blocks per workgroup, at most 1024 workgroups = CUs)
    if ((size_t)4 * 1024 * 128 > s3) s3 = (size_t)4 * 1024 * 128;
    if (s2 > st) st = s2;
    if (s3 > st) st = s3;
    const size_t bw = (size_t)(B * 16 * d.wp1 / 64 + 64) * 128;          // bn backward partials, generous
    if (bw > st) st = bw;
    n[TB_STATS] = 2 * st;
    n[TB_WP2] = n[TB_WP3] = n[TB_WP2T] = n[TB_WP3T] = 64;   // (slots of removed kernel generations; indices kept)
    n[TB_WHT] = n[TB_WR4] = 4 * GRU_FRAG_BYTES / sizeof(float);      // W_hh of 2 layers x 2 directions as resident fragments: forward (WHT), backward (WR4)
    n[TB_DY1] = B * S * 512;
    n[TB_DY0] = B * S * 512;
    n[TB_DGI] = B * S * 1536;
    n[TB_DGH] = B * S * 1536;
    n[TB_DX0] = B * S * 1024;
    n[TB_DZ3] = n[TB_Z3];
    n[TB_DA2] = n[TB_A2];
    n[TB_DZ2] = n[TB_Z2];
    n[TB_DA1] = n[TB_A1];
    n[TB_SMALL] = B * 512 + B + 64 + (size_t)d.c1gx * d.c1gy * B * 352;      // conv1 backward partials: 32 x 11 per block
    size_t slab = (size_t)d.wg3_blocks * 9 * 128 * 64;
    const size_t s_w2 = (size_t)d.wg2_blocks * 9 * 64 * 32, s_g = (size_t)d.ksplits * 768 * 1024;
    if (s_w2 > slab) slab = s_w2;
    if (s_g > slab) slab = s_g;
    for (int in_sz : {1024, 512}) {                       // slabs of the four-job bf16x6 weight-gradient launch (size independent of the batch)
        int t_, kc_, ns_;
        size_t need;
        tn_x6_plan(d.B * d.S, in_sz, &t_, &kc_, &ns_, &need);
        if (need > slab) slab = need;
        if (dx_splitk((int)(d.B * d.S), in_sz) && (size_t)2 * d.B * d.S * in_sz > slab) slab = (size_t)2 * d.B * d.S * in_sz;
    }
    if ((size_t)64 * 16 * 128 * 64 > slab) slab = (size_t)64 * 16 * 128 * 64;      // Winograd weight-gradient slabs: 64 strips of conv3, 128 of conv2
    n[TB_SLAB] = slab + (size_t)WGR_PARTS * 16 * 128 * 64;      // + the partial sums of the two-pass wgrad reduce
    n[TB_XS] = (B * S * 1024 * 2 + 1) / 2;                       // ushort count / 2 (sizes are in floats)
    n[TB_WS] = ((size_t)2 * 2 * 768 * 1024 + (size_t)2 * 2 * 768 * 512 + 1) / 2;
    // conv2 / conv3 forward and both data gradients in Winograd form (16 frequencies; the conv2 data gradient's slot also holds its 9-tap form when the direct kernel runs it), conv3 forward again with 9 taps for the direct fallback
    n[TB_WCB] = ((size_t)(3 * 32 * 16 * 64 + 3 * 32 * 16 * 64) + (size_t)3 * 64 * 16 * 128 + (size_t)3 * 128 * 16 * 64 + (size_t)3 * 64 * 9 * 128 + 1) / 2;
    n[TB_GXB] = 64;
    n[TB_GFL] = 64;
    n[TB_C1M] = 2 * C1_NMOM;
    n[TB_DGI1] = B * S * 1536;
    n[TB_DGH1] = B * S * 1536;
    n[TB_SLAB2] = 64;
    for (int in_sz : {1024, 512}) {
        int t_, kc_, ns_;
        size_t need;
        tn_x6_plan(d.B * d.S, in_sz, &t_, &kc_, &ns_, &need);
        if (need > n[TB_SLAB2]) n[TB_SLAB2] = need;
    }
}

size_t tws_layout(const TDims& d, size_t* off) {
    size_t n[TB_COUNT], pos = 0;
    tws_sizes(d, n);
    for (int i = 0; i < TB_COUNT; ++i) {
        off[i] = pos;
        pos += sir_align_up(n[i] * sizeof(float), 256);
    }
    return pos;
}


#define KCHECK() SIR_HIP_TRY(hipGetLastError())

struct TPtrs {
    float *a1, *z2, *a2, *z3, *x0, *gi, *g0, *g1, *y0, *y0d, *y1, *ctx, *bn, *bnb, *wp2, *wp3, *wht, *wr4, *wp2t, *wp3t;
    float *dgi1, *dgh1, *slab2;
    float *dy1, *dy0, *dgi, *dgh, *dx0, *dz3, *da2, *dz2, *da1, *small, *slab;
    float2* stats;
    unsigned short *xs, *wsl0, *wsl1, *wcb2, *wcb3, *wcb2t, *wcb3t, *wcb3d;
    unsigned int* gfl;
    double* c1m;
};

TPtrs carve(void* ws, const size_t* off) {
    char* b = (char*)ws;
    TPtrs p;
    p.a1 = (float*)(b + off[TB_A1]); p.z2 = (float*)(b + off[TB_Z2]); p.a2 = (float*)(b + off[TB_A2]);
    p.z3 = (float*)(b + off[TB_Z3]); p.x0 = (float*)(b + off[TB_X0]); p.gi = (float*)(b + off[TB_GI]);
    p.g0 = (float*)(b + off[TB_G0]); p.g1 = (float*)(b + off[TB_G1]); p.y0 = (float*)(b + off[TB_Y0]);
    p.y0d = (float*)(b + off[TB_Y0D]); p.y1 = (float*)(b + off[TB_Y1]); p.ctx = (float*)(b + off[TB_CTX]);
    p.bn = (float*)(b + off[TB_BN]); p.bnb = (float*)(b + off[TB_BNB]); p.stats = (float2*)(b + off[TB_STATS]);
    p.wp2 = (float*)(b + off[TB_WP2]); p.wp3 = (float*)(b + off[TB_WP3]); p.wht = (float*)(b + off[TB_WHT]);
    p.wr4 = (float*)(b + off[TB_WR4]); p.wp2t = (float*)(b + off[TB_WP2T]); p.wp3t = (float*)(b + off[TB_WP3T]);
    p.dy1 = (float*)(b + off[TB_DY1]); p.dy0 = (float*)(b + off[TB_DY0]); p.dgi = (float*)(b + off[TB_DGI]);
    p.dgh = (float*)(b + off[TB_DGH]); p.dx0 = (float*)(b + off[TB_DX0]); p.dz3 = (float*)(b + off[TB_DZ3]);
    p.da2 = (float*)(b + off[TB_DA2]); p.dz2 = (float*)(b + off[TB_DZ2]); p.da1 = (float*)(b + off[TB_DA1]);
    p.small = (float*)(b + off[TB_SMALL]); p.slab = (float*)(b + off[TB_SLAB]);
    p.xs = (unsigned short*)(b + off[TB_XS]);
    p.wsl0 = (unsigned short*)(b + off[TB_WS]); p.wsl1 = p.wsl0 + (size_t)2 * 2 * 768 * 1024;
    p.wcb2 = (unsigned short*)(b + off[TB_WCB]); p.wcb3 = p.wcb2 + (size_t)3 * 32 * 16 * 64;
    p.wcb2t = p.wcb3 + (size_t)3 * 64 * 16 * 128; p.wcb3t = p.wcb2t + (size_t)3 * 32 * 16 * 64;
    p.wcb3d = p.wcb3t + (size_t)3 * 128 * 16 * 64;           // conv3 forward with 9 taps: only for shapes the Winograd kernel does not cover
    p.gfl = (unsigned int*)(b + off[TB_GFL]);
    p.c1m = (double*)(b + off[TB_C1M]);
    p.dgi1 = (float*)(b + off[TB_DGI1]); p.dgh1 = (float*)(b + off[TB_DGH1]);
    p.slab2 = (float*)(b + off[TB_SLAB2]);
    return p;
}

int check_common(const char* who, sir_handle* h, const sir_model_weights* w, int batch, int t, void* ws, size_t bytes,
                 TDims* d, size_t* off) {
    if (!h || !w || !ws) { sir_set_error("%s: NULL argument", who); return SIR_EINVAL; }
    if (!make_tdims(batch, t, d)) { sir_set_error("%s: unsupported shape batch=%d t_frames=%d", who, batch, t); return SIR_EINVAL; }
    if (h->cfg.n_mels != 64) { sir_set_error("%s: the model is wired for 64 mels", who); return SIR_EUNSUPPORTED; }
    if (w->num_classes < 1 || w->num_classes > 64) { sir_set_error("%s: num_classes=%d", who, w->num_classes); return SIR_EINVAL; }
    const size_t need = tws_layout(*d, off);
    if (bytes < need) { sir_set_error("%s: workspace %zu < %zu", who, bytes, need); return SIR_ENOMEM; }
    if (((uintptr_t)ws & 255) != 0) { sir_set_error("%s: workspace must be 256-byte aligned", who); return SIR_EINVAL; }
    return SIR_OK;
}

// Loss scale of the backward (a power of two, exact in fp32 both ways): head_bwd_kernel multiplies d(loss)/d(GRU output) by it and
// every kernel that writes a PARAMETER gradient behind it multiplies by its inverse, so that the intermediate gradients -- 1e-5 to
// 1e-7 at batch 256 unscaled -- sit around 2^-4 .. 2^4: inside fp16's normal range for the f16x3 contractions of the backward
// (f16_split.h), with 2^10 of head room on either side.  2^8 x batch (rounded up to a power of two) makes the scaled d(logits)
// (softmax - onehot) x 2^8 whatever the batch.  Results are bit-identical to the unscaled backward wherever the arithmetic is fp32
// or bf16x6 (scaling by 2^k commutes with every rounding there).
inline float sir_bwd_loss_scale(int batch) {
    int k = 8;
    while ((1 << (k - 8)) < batch && k < 24) ++k;
    return (float)(1u << k);
}

inline int grid_for(size_t n, int per_block = 256, int cap = 8192) {
    size_t g = (n + per_block - 1) / per_block;
    return (int)(g > (size_t)cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

size_t sir_train_workspace_bytes_impl(int batch, int t_frames) {
    TDims d;
    if (!make_tdims(batch, t_frames, &d)) return 0;
    size_t off[TB_COUNT];
    return tws_layout(d, off);
}

extern "C" int sir_model_train_workspace_offsets(const sir_handle* h, int batch, int t_frames, size_t* offsets, int n) {
    (void)h;
    TDims d;
    if (!make_tdims(batch, t_frames, &d) || !offsets) { sir_set_error("sir_model_train_workspace_offsets: bad shape"); return SIR_EINVAL; }
    size_t off[TB_COUNT];
    tws_layout(d, off);
    for (int i = 0; i < n && i < TB_COUNT; ++i) offsets[i] = off[i];
    return TB_COUNT;
}

extern "C" int sir_model_train_fwd(sir_handle* h, const sir_model_weights* w, float* const bn_running_mean[3],
                                   float* const bn_running_var[3], const float* feats, int batch, int t_frames,
                                   float bn_momentum, float dropout_p, uint64_t dropout_seed, float* logits,
                                   void* workspace, size_t workspace_bytes, void* stream_) {
    TDims d;
    size_t off[TB_COUNT];
    int rc = check_common("sir_model_train_fwd", h, w, batch, t_frames, workspace, workspace_bytes, &d, off);
    if (rc != SIR_OK) return rc;
    if (!feats || !logits || !bn_running_mean || !bn_running_var) { sir_set_error("sir_model_train_fwd: NULL argument"); return SIR_EINVAL; }
    if (dropout_p < 0.0f || dropout_p >= 1.0f) { sir_set_error("sir_model_train_fwd: dropout_p=%f", dropout_p); return SIR_EINVAL; }
    hipStream_t st = (hipStream_t)stream_;
    TPtrs p = carve(workspace, off);
    const int B = d.B, S = d.S, T = d.T;
    float *scale = p.bn, *shift = p.bn + 224, *smean = p.bn + 448, *sinv = p.bn + 672;

    // conv2 / conv3 forward and the conv3 data gradient run on the producer / consumer Winograd kernel (conv_wino2_bf16x6_kernel.h);
    // shapes it does not cover keep the first-generation / direct kernels
    Wino2Geo geo2, geo3;
    const bool w2ok_all = wino2_geo(B, 32, d.wp1, 64, &geo2) && wino2_geo(B, 16, d.wp2, 128, &geo3);
    const bool w2c2 = w2ok_all && (sir_wino2_mask() & 1), w2c3 = w2ok_all && (sir_wino2_mask() & 2);
    const bool f16c2 = w2c2 && (sir_f16_mask() & 1), f16c3 = w2c3 && (sir_f16_mask() & 2);     // forward stages on the f16x3 arithmetic
    {   // all weight re-layouts of this step, the backward's included (the weights do not change before it runs)
        SirProfScope prof(h, SIR_K_T_PREP, st);
        PrepJobs pj{};
        int nj = 0, blocks = 0;
        auto add = [&](int kind, const float* src, void* dst, int a, int b, int nblk) {
            pj.kind[nj] = kind; pj.src[nj] = src; pj.dst[nj] = dst; pj.a[nj] = a; pj.b[nj] = b; pj.block0[nj] = blocks;
            blocks += nblk; ++nj;
        };
        pj.status = h->status;
        add(f16c2 ? 6 : 4, w->conv_w[1], p.wcb2, 32, 64, (32 * 16 * 64 + 255) / 256);       // conv2 forward: Winograd frequencies
        add(f16c3 ? 6 : 4, w->conv_w[2], p.wcb3, 64, 128, (64 * 16 * 128 + 255) / 256);     // conv3 forward: Winograd frequencies
        if (!w2c3) add(1, w->conv_w[2], p.wcb3d, 64, 128, (64 * 9 * 128 + 255) / 256);
        {   // conv2 data gradient (64 -> 32): the second-generation Winograd kernel on f16x3 (its transform feeds only 32 outputs -- on
            // bf16x6 that lost to the direct kernel, with half the matrix products it wins: profiles/r04/bench_conv_f16x3.txt), else direct
            Wino2Geo geo2b;
            const bool f16d2 = wino2_geo(B, 32, d.wp1, 64, &geo2b) && (sir_wino2_mask() & 8) && (sir_f16_mask() & 16);
            if (f16d2) add(7, w->conv_w[1], p.wcb2t, 32, 64, (64 * 16 * 32 + 255) / 256);
            else add(2, w->conv_w[1], p.wcb2t, 32, 64, (32 * 9 * 64 + 255) / 256);
        }
        {   // conv3 data gradient: Winograd frequencies of the flipped taps (f16x3 planes when that stage runs on them)
            Wino2Geo geo3b;
            const bool f16d3 = wino2_geo(B, 16, d.wp2, 128, &geo3b) && (sir_wino2_mask() & 4) && (sir_f16_mask() & 4);
            add(f16d3 ? 7 : 5, w->conv_w[2], p.wcb3t, 64, 128, (128 * 16 * 64 + 255) / 256);
        }
        for (int dir = 0; dir < 2; ++dir) {
            add(0, w->gru_w_ih[dir], p.wsl0 + (size_t)dir * 2 * 768 * 1024, 1024, 768, 384);
            add(0, w->gru_w_ih[2 + dir], p.wsl1 + (size_t)dir * 2 * 768 * 512, 512, 768, 192);
        }
        for (int i = 0; i < 4; ++i) {                        // W_hh (layer i / 2, direction i % 2) as the recurrences' resident fragments
            add(8, w->gru_w_hh[i], (char*)p.wht + (size_t)i * GRU_FRAG_BYTES, 0, 0, GQ_FRAG_THREADS / 256);
            add(9, w->gru_w_hh[i], (char*)p.wr4 + (size_t)i * GRU_FRAG_BYTES, 0, 0, BQ_FRAG_THREADS / 256);
        }
        pj.block0[nj] = blocks;
        pj.njobs = nj;
        static_assert(PREP_MAX_JOBS >= 18, "job table");
        hipLaunchKernelGGL(train_prep_kernel, dim3(blocks), dim3(256), 0, st, pj);
    }
    KCHECK();

    // conv1 block: statistics pass (recompute), finalize, then the fused conv+BN+ReLU+pool pass
    {
        SirProfScope prof(h, SIR_K_T_CONV1, st);
        // conv1's BatchNorm statistics come from 54 moments of the INPUT (z_c = sum_t w_c[t] x_t: sums and sums of squares
        // of z are bilinear in the taps), so conv1 itself runs once, fused with BN + ReLU + pool
        const int tiles = d.c1gx * d.c1gy;
        int per_img = (2048 + B - 1) / B;             // workgroups per image: >= 2048 in all when the batch allows it
        per_img = per_img < 1 ? 1 : (per_img > tiles ? tiles : per_img);
        hipLaunchKernelGGL(conv1_moments_kernel, dim3(per_img, B), dim3(256), 0, st, feats, (float*)p.stats, 64, T, d.c1gx, d.c1gy);
        hipLaunchKernelGGL(conv1_moments_reduce_kernel, dim3(C1_NMOM), dim3(256), 0, st, (const float*)p.stats, per_img * B, p.c1m);
        hipLaunchKernelGGL(conv1_bn_from_moments_kernel, dim3(1), dim3(64), 0, st, (const double*)p.c1m, w->conv_w[0],
                           (double)B * 64 * T, w->bn_w[0], w->bn_b[0], bn_running_mean[0], bn_running_var[0], bn_momentum,
                           scale, shift, smean, sinv);
        hipLaunchKernelGGL(conv1_mfma_bn_relu_pool_kernel, dim3((d.wp1 + C1_PCOLS - 1) / C1_PCOLS, 1, B), dim3(256), 0, st,
                           feats, w->conv_w[0], scale, shift, p.a1, 64, T, 32, d.wp1);
    }
    // conv2 block: raw conv + partial statistics on MFMA, finalize, BN+ReLU+pool
    {
        { SirProfScope prof(h, SIR_K_T_CONV2, st);
        if (f16c2)
            SIR_HIP_TRY((launch_conv_wino2<32, 64, 2, 0, 3, true>(st, &h->attr_wino2[7], (const float*)p.a1, (const unsigned short*)p.wcb2, (const float*)nullptr,
                                                                (const float*)nullptr, p.z2, B, 32, d.wp1, p.stats, h->zero_page, h->num_cus)));
        else if (w2c2)
            SIR_HIP_TRY((launch_conv_wino2<32, 64, 2>(st, &h->attr_wino2[2], (const float*)p.a1, (const unsigned short*)p.wcb2, (const float*)nullptr,
                                                    (const float*)nullptr, p.z2, B, 32, d.wp1, p.stats, h->zero_page, h->num_cus)));
        else
            hipLaunchKernelGGL((conv3x3_wino_bf16x6_kernel<32, 64, 2>), dim3(d.c2wx, 1, B), dim3(256), WINO_LDS_BYTES, st, (const float*)p.a1,
                               (const unsigned short*)p.wcb2, (const float*)nullptr, (const float*)nullptr, p.z2, 32, d.wp1, 16, d.wp2, p.stats);
        }
        SirProfScope prof(h, SIR_K_T_BN2, st);
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(64), dim3(256), 0, st, (const float2*)p.stats, w2c2 ? (int)wino2_stat_blocks(B, 32, d.wp1, h->num_cus) : d.c2wx * B, 64,
                           (double)B * 32 * d.wp1, w->bn_w[1], w->bn_b[1], bn_running_mean[1], bn_running_var[1], bn_momentum,
                           scale + 32, shift + 32, smean + 32, sinv + 32);
        hipLaunchKernelGGL(bn_relu_pool_kernel<false>, dim3(grid_for((size_t)B * 16 * d.wp2 * 16)), dim3(256), 0, st, p.z2,
                           scale + 32, shift + 32, p.a2, B, 32, d.wp1, 64, 16, d.wp2);
    }
    {
        { SirProfScope prof(h, SIR_K_T_CONV3, st);
        if (f16c3)
            SIR_HIP_TRY((launch_conv_wino2<64, 128, 2, 0, 3, true>(st, &h->attr_wino2[8], (const float*)p.a2, (const unsigned short*)p.wcb3, (const float*)nullptr,
                                                                 (const float*)nullptr, p.z3, B, 16, d.wp2, p.stats, h->zero_page, h->num_cus)));
        else if (w2c3)
            SIR_HIP_TRY((launch_conv_wino2<64, 128, 2>(st, &h->attr_wino2[3], (const float*)p.a2, (const unsigned short*)p.wcb3, (const float*)nullptr,
                                                     (const float*)nullptr, p.z3, B, 16, d.wp2, p.stats, h->zero_page, h->num_cus)));
        else
            hipLaunchKernelGGL((conv3x3_bf16x6_ns_kernel<64, 128, 2, 2, 2, 0, 2, 1, 1>), dim3(d.c3fx, 1, B), dim3(256), conv_ns_lds_bytes(2, 2, 2), st, (const float*)p.a2,
                               (const unsigned short*)p.wcb3d, (const float*)nullptr, (const float*)nullptr, p.z3, 16, d.wp2, 8, d.wp3, p.stats);
        }
        SirProfScope prof(h, SIR_K_T_BN3, st);
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(128), dim3(256), 0, st, (const float2*)p.stats, w2c3 ? (int)wino2_stat_blocks(B, 16, d.wp2, h->num_cus) : d.c3fx * B, 128,
                           (double)B * 16 * d.wp2, w->bn_w[2], w->bn_b[2], bn_running_mean[2], bn_running_var[2], bn_momentum,
                           scale + 96, shift + 96, smean + 96, sinv + 96);
        hipLaunchKernelGGL(bn_relu_pool_kernel<true>, dim3(grid_for((size_t)B * 8 * d.wp3 * 32)), dim3(256), 0, st, p.z3,
                           scale + 96, shift + 96, p.x0, B, 16, d.wp2, 128, 8, d.wp3);
    }
    KCHECK();

    const int M = B * S;
    { SirProfScope prof(h, SIR_K_T_GEMM_IH0, st);
    hipLaunchKernelGGL(split2h_kernel, dim3(2048), dim3(256), 0, st, (const float*)p.x0, 1024, p.xs, (size_t)M, 1024);
    SIR_HIP_TRY(launch_gemm_nt_f16x3(h, st, (const unsigned short*)p.xs, (const unsigned short*)p.wsl0,
                       (const unsigned short*)(p.wsl0 + (size_t)2 * 768 * 1024), w->gru_b_ih[0], w->gru_b_ih[1], p.gi, 1536, M, 768, 1024)); }
    { SirProfScope prof(h, SIR_K_T_GRU0, st);
    if (sir_cluster_enter(h, st) != SIR_OK) return SIR_EHIP;
    rc = sir_launch_gru_quad(h, st, true, p.gi, w->gru_w_hh[0], w->gru_w_hh[1], w->gru_b_hh[0], w->gru_b_hh[1], p.y0, B, S, p.g0, nullptr,
                             (const char*)p.wht, (const char*)p.wht + GRU_FRAG_BYTES);
    if (sir_cluster_leave(h, st) != SIR_OK) return SIR_EHIP; }
    if (rc != SIR_OK) return rc;
    const float* y0in = p.y0;
    { SirProfScope prof(h, SIR_K_T_GEMM_IH1, st);
    if (dropout_p > 0.0f) {                                   // dropout + the f16x2 planes of its output in one pass
        hipLaunchKernelGGL(dropout_split2h_kernel, dim3(2048), dim3(256), 0, st, (const float*)p.y0, p.y0d, p.xs, (size_t)M * 512,
                           dropout_p, (unsigned long long)dropout_seed);
        y0in = p.y0d;
    } else {
        hipLaunchKernelGGL(split2h_kernel, dim3(2048), dim3(256), 0, st, y0in, 512, p.xs, (size_t)M, 512);
    }
    SIR_HIP_TRY(launch_gemm_nt_f16x3(h, st, (const unsigned short*)p.xs, (const unsigned short*)p.wsl1,
                       (const unsigned short*)(p.wsl1 + (size_t)2 * 768 * 512), w->gru_b_ih[2], w->gru_b_ih[3], p.gi, 1536, M, 768, 512)); }
    { SirProfScope prof(h, SIR_K_T_GRU1, st);
    if (sir_cluster_enter(h, st) != SIR_OK) return SIR_EHIP;
    rc = sir_launch_gru_quad(h, st, true, p.gi, w->gru_w_hh[2], w->gru_w_hh[3], w->gru_b_hh[2], w->gru_b_hh[3], p.y1, B, S, p.g1, nullptr,
                             (const char*)p.wht + 2 * GRU_FRAG_BYTES, (const char*)p.wht + 3 * GRU_FRAG_BYTES);
    if (sir_cluster_leave(h, st) != SIR_OK) return SIR_EHIP; }
    if (rc != SIR_OK) return rc;
    SirProfScope prof_head(h, SIR_K_T_HEAD, st);
    hipLaunchKernelGGL(attention_pool_kernel, dim3(B), dim3(256), 0, st, p.y1, w->attn_w, w->attn_b, p.ctx, S, w->fc_w,
                       w->fc_b, w->num_classes, logits, (long long*)nullptr);
    KCHECK();
    return SIR_OK;
}

extern "C" int sir_ce_loss(sir_handle* h, const float* logits, const int64_t* labels, int batch, int num_classes,
                           float* loss, float* dlogits, float grad_scale, void* stream_) {
    if (!h || !logits || !labels || !loss) { sir_set_error("sir_ce_loss: NULL argument"); return SIR_EINVAL; }
    if (batch < 1 || num_classes < 1 || num_classes > 64) { sir_set_error("sir_ce_loss: bad shape batch=%d num_classes=%d (1..64)", batch, num_classes); return SIR_EINVAL; }
    SirProfScope prof(h, SIR_K_CE, (hipStream_t)stream_);
    if (num_classes <= 32)
        hipLaunchKernelGGL(ce_loss_kernel<32>, dim3(1), dim3(256), 0, (hipStream_t)stream_, logits, (const long long*)labels, batch,
                           num_classes, loss, dlogits, grad_scale, h->status);
    else
        hipLaunchKernelGGL(ce_loss_kernel<64>, dim3(1), dim3(256), 0, (hipStream_t)stream_, logits, (const long long*)labels, batch,
                           num_classes, loss, dlogits, grad_scale, h->status);
    KCHECK();
    return SIR_OK;
}

extern "C" int sir_model_train_bwd(sir_handle* h, const sir_model_weights* w, const float* feats, const float* dlogits,
                                   int batch, int t_frames, float dropout_p, uint64_t dropout_seed,
                                   const sir_model_grads* g, void* workspace, size_t workspace_bytes, void* stream_) {
    return sir_model_train_bwd_part(h, w, feats, dlogits, batch, t_frames, dropout_p, dropout_seed, g, workspace, workspace_bytes,
                                    SIR_BWD_ALL, stream_);
}

extern "C" int sir_model_train_bwd_part(sir_handle* h, const sir_model_weights* w, const float* feats, const float* dlogits,
                                        int batch, int t_frames, float dropout_p, uint64_t dropout_seed,
                                        const sir_model_grads* g, void* workspace, size_t workspace_bytes, int part,
                                        void* stream_) {
    if (part != SIR_BWD_ALL && part != SIR_BWD_HEAD_GRU && part != SIR_BWD_CNN) {
        sir_set_error("sir_model_train_bwd_part: unknown part %d", part);
        return SIR_EINVAL;
    }
    TDims d;
    size_t off[TB_COUNT];
    int rc = check_common("sir_model_train_bwd", h, w, batch, t_frames, workspace, workspace_bytes, &d, off);
    if (rc != SIR_OK) return rc;
    if (!feats || !dlogits || !g) { sir_set_error("sir_model_train_bwd: NULL argument"); return SIR_EINVAL; }
    hipStream_t st = (hipStream_t)stream_;
    TPtrs p = carve(workspace, off);
    const int B = d.B, S = d.S, T = d.T, C = w->num_classes, M = B * S;
    float *scale = p.bn, *shift = p.bn + 224, *smean = p.bn + 448, *sinv = p.bn + 672;
    float *mdy = p.bnb, *mdyx = p.bnb + 224;
    float* daw_part = p.small;
    float* dab_part = p.small + (size_t)B * 512;
    float* c1part = p.small + (size_t)B * 512 + B + 64;
    float* bsum_i = p.slab;                              // [B][1536] x2, consumed before the slabs are used
    float* bsum_h = p.slab + (size_t)B * 1536;
    const float* y0in = dropout_p > 0.0f ? p.y0d : p.y0;
    const float gscale = sir_bwd_loss_scale(B), unscale = 1.0f / gscale;

    // ---- two-stream form (SIR_BWD_STREAMS, default 3; A/B in profiles/r04/ab_bwd_streams.txt) -------------------------------
    // The launches that nothing downstream waits for -- the GRU weight gradients of both layers and the two convolution weight
    // gradients, with their slab reduces -- go to a stream owned by the handle.  Each GRU weight-gradient GEMM forks right behind
    // ITS layer's BPTT (bit 1 of the mode): layer 1's then runs beside layer 0's BPTT, which occupies half of the CUs and leaves the
    // rest idle.  Each convolution weight gradient forks behind the BatchNorm backward that produces its dz (bit 0).  One join
    // before the call returns.  The chain dX -> BN3 -> dgrad3 -> BN2 -> dgrad2 -> conv1 stays on the caller's stream.  In the split
    // form (SIR_BWD_HEAD_GRU / SIR_BWD_CNN, data parallel) the first half joins before it returns -- its gradients are reduced
    // next.  Mode 1 (round 4's first experiment): the GRU GEMMs fork once, behind the last dX.
    if (sir_bwd_streams() && !h->bwd_side) {                 // (first use: the only allocating step, as for the exchange buffers)
        SIR_HIP_TRY(hipStreamCreateWithFlags(&h->bwd_side, hipStreamNonBlocking));
        for (auto& e : h->bwd_ev) SIR_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (!h->attr_tn) {
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn_bf16x6_kernel<true, TN_BM_DW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tn_lds_bytes(true, TN_BM_DW)));
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn_bf16x6_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TN_LDS_BYTES));
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn_bf16x6_kernel<false, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TN_LDS_BYTES_64));
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn2_bf16x6_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tn2_lds_bytes(true)));
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn2_bf16x6_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tn2_lds_bytes(false)));
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn2_bf16x6_kernel<false, 0, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tn2_lds_bytes(false, 64)));
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn2_bf16x6_kernel<true, 0, TN2_BM, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tn2_lds_bytes(true)));
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn2_bf16x6_kernel<false, 0, TN2_BM, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tn2_lds_bytes(false)));
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn2_bf16x6_kernel<false, 0, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tn2_lds_bytes(false, 64)));
        h->attr_tn = true;
    }
    // (while every kernel is being timed -- sir_profile_enable mode 1 -- the backward stays on one stream: per-kernel times of overlapped
    // launches would say nothing about the kernels)
    const bool two = sir_bwd_streams() && h->bwd_side != nullptr && h->prof_mode != 1;
    hipStream_t side = two ? h->bwd_side : st;
    // SIR_BWD_STREAMS bits: 1 = the convolution weight gradients fork to the side stream (behind the BatchNorm backward that produces
    // their dz), and in mode 1 the GRU weight-gradient GEMMs follow behind the last dX; 2 = the GRU weight-gradient GEMMs leave the
    // caller's stream right behind THEIR layer's BPTT -- layer 1's then runs beside layer 0's BPTT, which keeps one workgroup on
    // half of the CUs (gru_bwd_quad_kernel.h) and leaves the rest idle; 3 = both
    const bool dw_beside = two && (sir_bwd_streams() & 2);
    const bool fork_conv = two && (sir_bwd_streams() & 1);
    const bool defer_dw = two && !dw_beside && part == SIR_BWD_ALL;

    // all four weight-gradient GEMMs of a GRU layer (2 directions x {W_ih, W_hh}) in one bf16x6 launch + the slab reduce
    auto launch_dw = [&](int layer, hipStream_t s_) -> int {
        const float* dgi_l = layer ? p.dgi1 : p.dgi;
        const float* dgh_l = layer ? p.dgh1 : p.dgh;
        const float* yout = layer ? p.y1 : p.y0;
        const float* xin = layer ? y0in : p.x0;
        const int in_sz = layer ? 512 : 1024;
        SirProfScope prof(h, layer ? SIR_K_B_DW1 : SIR_K_B_DW0, s_);
        const bool tn2_dw = sir_tn2_mask() & 1;
        TnJobs jb{};
        float* outs[4];
        size_t sizes[4];
        jb.njobs = 4;
        jb.zeros = h->zero_page;
        int tiles = 0;
        for (int dir = 0; dir < 2; ++dir) {
            const int gi_idx = 2 * layer + dir;
            const int ja = 2 * dir, jh = 2 * dir + 1;
            jb.A[ja] = dgi_l + dir * 768; jb.lda[ja] = 1536; jb.B[ja] = xin; jb.ldb[ja] = in_sz; jb.N[ja] = in_sz; jb.shift[ja] = 0;
            outs[ja] = g->gru_w_ih[gi_idx];
            jb.A[jh] = dgh_l + dir * 768; jb.lda[jh] = 1536; jb.B[jh] = yout + dir * 256; jb.ldb[jh] = 512; jb.N[jh] = 256;
            jb.shift[jh] = dir ? 1 : -1;
            outs[jh] = g->gru_w_hh[gi_idx];
        }
        for (int j = 0; j < 4; ++j) {
            jb.tile0[j] = tiles;
            tiles += (768 / (tn2_dw ? TN2_BM : TN_BM_DW)) * ((jb.N[j] + TN_BN - 1) / TN_BN);
            sizes[j] = (size_t)768 * jb.N[j];
        }
        jb.tile0[4] = tiles;
        int tiles_chk, kchunk, nsplit;
        size_t need;
        tn_x6_plan(M, in_sz, &tiles_chk, &kchunk, &nsplit, &need);
        size_t pos = 0;
        for (int j = 0; j < 4; ++j) {
            jb.slab[j] = (dw_beside ? p.slab2 : p.slab) + pos;
            jb.slab_stride[j] = sizes[j];
            pos += sizes[j] * nsplit;
        }
        if (tn2_dw && (sir_f16_mask() & 8))                     // f16x3: the gate gradients carry the loss scale
            hipLaunchKernelGGL((gemm_tn2_bf16x6_kernel<true, 0, TN2_BM, true>), dim3(tiles, nsplit), dim3(TN2_THREADS), tn2_lds_bytes(true), s_, jb, 768, M, kchunk, S);
        else if (tn2_dw)
            hipLaunchKernelGGL(gemm_tn2_bf16x6_kernel<true>, dim3(tiles, nsplit), dim3(TN2_THREADS), tn2_lds_bytes(true), s_, jb, 768, M, kchunk, S);
        else
            hipLaunchKernelGGL((gemm_tn_bf16x6_kernel<true, TN_BM_DW>), dim3(tiles, nsplit), dim3(512), tn_lds_bytes(true, TN_BM_DW), s_, jb, 768, M, kchunk, S);
        SlabJobs sj{};
        for (int j = 0; j < 4; ++j) { sj.src[j] = jb.slab[j]; sj.out[j] = outs[j]; sj.n[j] = sizes[j]; }
        hipLaunchKernelGGL(slab_reduce_jobs_kernel, dim3(grid_for(sizes[0]), 4), dim3(256), 0, s_, sj, nsplit, unscale);
        return SIR_OK;
    };

    if (part != SIR_BWD_CNN) {
    // ---- head: fc + attention pooling ----------------------------------------------------
    { SirProfScope prof(h, SIR_K_B_HEAD, st);
    hipLaunchKernelGGL(head_bwd_kernel, dim3(B + 2 * C), dim3(256), 0, st, dlogits, w->fc_w, (const float*)p.y1, w->attn_w, w->attn_b,
                       (const float*)p.ctx, p.dy1, daw_part, dab_part, g->fc_w, g->fc_b, B, S, C, gscale);
    hipLaunchKernelGGL(head_colsum_kernel, dim3(9), dim3(256), 0, st, (const float*)daw_part, (const float*)dab_part, B, g->attn_w, g->attn_b); }
    KCHECK();

    // ---- GRU layers, top down ----------------------------------------------------------------
    for (int layer = 1; layer >= 0; --layer) {
        const float* dy = layer ? p.dy1 : p.dy0;
        const float* gates = layer ? p.g1 : p.g0;
        const float* yout = layer ? p.y1 : p.y0;
        float* dgi_l = layer ? p.dgi1 : p.dgi;
        float* dgh_l = layer ? p.dgh1 : p.dgh;
        const int in_sz = layer ? 512 : 1024;
        { SirProfScope prof(h, layer ? SIR_K_B_GRU1 : SIR_K_B_GRU0, st);
        if (sir_cluster_enter(h, st) != SIR_OK) return SIR_EHIP;
        rc = sir_launch_gru_bwd_pair(h, st, dy, gates, yout, w->gru_w_hh[2 * layer], w->gru_w_hh[2 * layer + 1], dgi_l, dgh_l, bsum_i, bsum_h,
                                     B, S, (const char*)p.wr4 + (size_t)(2 * layer) * GRU_FRAG_BYTES,
                                     (const char*)p.wr4 + (size_t)(2 * layer + 1) * GRU_FRAG_BYTES);
        if (rc != SIR_OK) return rc;
        if (sir_cluster_leave(h, st) != SIR_OK) return SIR_EHIP;
        // bias gradients first: bsum_* alias the slab area used below
        hipLaunchKernelGGL(gru_bias_colsum_kernel, dim3(24, 2), dim3(256), 0, st, (const float*)bsum_i, (const float*)bsum_h, B,
                           g->gru_b_ih[2 * layer], g->gru_b_ih[2 * layer + 1], g->gru_b_hh[2 * layer], g->gru_b_hh[2 * layer + 1], unscale); }
        if (dw_beside) {                                     // (layer 0's GEMM queues behind layer 1's on the side stream: they share the slabs)
            SIR_HIP_TRY(hipEventRecord(h->bwd_ev[4 + layer], st));
            SIR_HIP_TRY(hipStreamWaitEvent(side, h->bwd_ev[4 + layer], 0));
            // Layer 0's saved gates and outputs (65 MB) were written early in the forward and have left the 256 MB last-level cache by now;
            // layer 1's are still there, and layer 0's BPTT -- a latency chain whose polls share the L2 channels with its input misses --
            // pays 16-26 us for the difference (profiles/r04/ab_bptt.txt).  A read-and-drop pass on the side stream, beside layer 1's dX on
            // the caller's, brings them back: step -28 .. -40 us (SIR_BPTT_TOUCH=0 turns it off).  The same for the raw conv outputs ahead
            // of the BatchNorm backward was measured and LOSES (those kernels are bandwidth-bound: the reads are only moved earlier).
            static const bool touch = !getenv("SIR_BPTT_TOUCH") || atoi(getenv("SIR_BPTT_TOUCH")) != 0;
            if (touch && layer == 1)
                hipLaunchKernelGGL(cache_touch_kernel, dim3(256), dim3(256), 0, side, (const float4*)p.g0, (size_t)M * 2048 / 4, (const float4*)p.y0,
                                   (size_t)M * 512 / 4, p.small);
            rc = launch_dw(layer, side);
            if (rc != SIR_OK) return rc;
        } else if (!defer_dw) { rc = launch_dw(layer, st); if (rc != SIR_OK) return rc; }
        // gradient wrt the layer input: dgi [M][1536] x [W_ih; W_ih_reverse] [1536][in]
        SirProfScope prof(h, layer ? SIR_K_B_DX1 : SIR_K_B_DX0, st);
        float* dxin = layer ? p.dy0 : p.dx0;
        {
            TnJobs jn{};
            jn.njobs = 1;
            jn.zeros = h->zero_page;
            if (layer == 1 && dropout_p > 0.0f) { jn.drop_p = dropout_p; jn.drop_seed = dropout_seed; }   // dy0 = mask * d(y0d)
            jn.A[0] = dgi_l; jn.lda[0] = 1536;
            jn.B[0] = w->gru_w_ih[2 * layer]; jn.B2[0] = w->gru_w_ih[2 * layer + 1]; jn.brows[0] = 768; jn.ldb[0] = in_sz;
            jn.N[0] = in_sz; jn.shift[0] = 0;
            jn.slab[0] = dxin; jn.slab_stride[0] = 0;
            jn.tile0[0] = 0;
            const int ntn = (in_sz + TN_BN - 1) / TN_BN;
            int ntiles = ((M + TN_BM - 1) / TN_BM) * ntn;
            if (dx_splitk(M, in_sz)) {
                jn.drop_p = 0.0f;                                // (the dropout mask is applied by the add)
                jn.slab[0] = p.slab; jn.slab_stride[0] = (size_t)M * in_sz;
                jn.tile0[1] = ntiles;
                if (sir_f16_mask() & 8)
                    hipLaunchKernelGGL((gemm_tn2_bf16x6_kernel<false, 0, TN2_BM, true>), dim3(ntiles, 2), dim3(TN2_THREADS), tn2_lds_bytes(false), st, jn, M, 1536, 768, 1);
                else
                    hipLaunchKernelGGL(gemm_tn2_bf16x6_kernel<false>, dim3(ntiles, 2), dim3(TN2_THREADS), tn2_lds_bytes(false), st, jn, M, 1536, 768, 1);
                const bool drop = layer == 1 && dropout_p > 0.0f;
                hipLaunchKernelGGL(dx_halves_add_kernel, dim3(grid_for((size_t)M * in_sz / 4)), dim3(256), 0, st, (const float*)p.slab, (size_t)M * in_sz / 4,
                                   dxin, drop ? dropout_p : 0.0f, (unsigned long long)dropout_seed);
            } else if (ntiles < 160) {                       // too few 128-row tiles to fill the CUs: 64-row tiles
                ntiles = ((M + 63) / 64) * ntn;
                jn.tile0[1] = ntiles;
                if ((sir_tn2_mask() & 4) && (sir_f16_mask() & 8))
                    hipLaunchKernelGGL((gemm_tn2_bf16x6_kernel<false, 0, 64, true>), dim3(ntiles, 1), dim3(TN2_THREADS), tn2_lds_bytes(false, 64), st, jn, M, 1536, 1536, 1);
                else if (sir_tn2_mask() & 4)
                    hipLaunchKernelGGL((gemm_tn2_bf16x6_kernel<false, 0, 64>), dim3(ntiles, 1), dim3(TN2_THREADS), tn2_lds_bytes(false, 64), st, jn, M, 1536, 1536, 1);
                else
                    hipLaunchKernelGGL((gemm_tn_bf16x6_kernel<false, 64>), dim3(ntiles, 1), dim3(512), TN_LDS_BYTES_64, st, jn, M, 1536, 1536, 1);
            } else {
                jn.tile0[1] = ntiles;
                if ((sir_tn2_mask() & 2) && (sir_f16_mask() & 8))
                    hipLaunchKernelGGL((gemm_tn2_bf16x6_kernel<false, 0, TN2_BM, true>), dim3(ntiles, 1), dim3(TN2_THREADS), tn2_lds_bytes(false), st, jn, M, 1536, 1536, 1);
                else if (sir_tn2_mask() & 2)
                    hipLaunchKernelGGL(gemm_tn2_bf16x6_kernel<false>, dim3(ntiles, 1), dim3(TN2_THREADS), tn2_lds_bytes(false), st, jn, M, 1536, 1536, 1);
                else
                    hipLaunchKernelGGL(gemm_tn_bf16x6_kernel<false>, dim3(ntiles, 1), dim3(512), TN_LDS_BYTES, st, jn, M, 1536, 1536, 1);
            }
        }
        KCHECK();
    }
    if (defer_dw) {
        // fork: the weight-gradient GEMMs of both layers behind the last dX (their slabs reuse the area the dX halves and the bias
        // partial sums used on the caller's stream)
        SIR_HIP_TRY(hipEventRecord(h->bwd_ev[0], st));
        SIR_HIP_TRY(hipStreamWaitEvent(side, h->bwd_ev[0], 0));
        for (int layer = 1; layer >= 0; --layer) { rc = launch_dw(layer, side); if (rc != SIR_OK) return rc; }
        KCHECK();
    }
    }
    if (dw_beside) {                                         // join: the GRU gradients are final on the caller's stream (the conv chain
        SIR_HIP_TRY(hipEventRecord(h->bwd_ev[3], side));     // below does not depend on them, but the data-parallel caller reduces them next)
        if (part == SIR_BWD_HEAD_GRU) SIR_HIP_TRY(hipStreamWaitEvent(st, h->bwd_ev[3], 0));
    }
    if (part == SIR_BWD_HEAD_GRU) return SIR_OK;
    if (fork_conv && !defer_dw) {                                  // SIR_BWD_CNN of the split form: the side stream starts behind the first half
        SIR_HIP_TRY(hipEventRecord(h->bwd_ev[0], st));
        SIR_HIP_TRY(hipStreamWaitEvent(side, h->bwd_ev[0], 0));
    }

    hipStream_t cside = fork_conv ? side : st;               // stream of the convolution weight gradients
    // ---- conv3 block -------------------------------------------------------------------------
    if (!h->attr_wgrad) {
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)conv_wgrad_bf16x6_kernel<64, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)conv_wgrad_bf16x6_kernel<32, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)conv_wgrad_wino_bf16x6_kernel<64, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WgwCfg<64, 128>::lds_bytes));
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)conv_wgrad_wino_bf16x6_kernel<32, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WgwCfg<32, 64>::lds_bytes));
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)conv_wgrad_wino_bf16x6_kernel<64, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WgwCfg<64, 128>::lds_bytes));
        SIR_HIP_TRY(hipFuncSetAttribute((const void*)conv_wgrad_wino_bf16x6_kernel<32, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WgwCfg<32, 64>::lds_bytes));
        h->attr_wgrad = true;
    }
    {
        // BatchNorm backward sums from the POOLED activations x0 (GRU layout) and their gradient -- dy = da wherever a > 0 and
        // xhat at the routed maximum is (a - beta) / gamma -- instead of the four times larger raw conv output z3
        const int rows = B * d.wp3, rpb = 16, nfin = (rows + rpb - 1) / rpb;
        {
            SirProfScope prof(h, SIR_K_B_BN3, st);
            hipLaunchKernelGGL(bn_bwd_reduce_pooled_gru_kernel, dim3(nfin), dim3(256), 0, st, (const float*)p.x0, (const float*)p.dx0,
                               (const float*)p.z3, w->bn_w[2], w->bn_b[2], scale + 96, shift + 96, smean + 96, sinv + 96, p.stats, rows,
                               16, d.wp2, d.wp3, rpb);
            hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(128), dim3(256), 0, st, (const float2*)p.stats, nfin, 128,
                               (double)B * 16 * d.wp2, g->bn_w[2], g->bn_b[2], mdy + 96, mdyx + 96, unscale);
            hipLaunchKernelGGL(bn_bwd_dz_kernel<true>, dim3(grid_for((size_t)B * 8 * ((d.wp2 + 1) / 2) * 32)), dim3(256), 0, st, (const float*)p.z3,
                               (const float*)p.dx0, scale + 96, shift + 96, smean + 96, sinv + 96, mdy + 96, mdyx + 96, p.dz3, B, 16,
                               d.wp2, 128, 8, d.wp3);
        }
        if (fork_conv) {
            SIR_HIP_TRY(hipEventRecord(h->bwd_ev[1], st));
            SIR_HIP_TRY(hipStreamWaitEvent(side, h->bwd_ev[1], 0));
        }
        {
            SirProfScope prof(h, SIR_K_B_WGRAD3, cside);
            if ((sir_wgw_mask() & 2) && (size_t)B * 16 * d.wp2 * 128 * 4 < ((size_t)1 << 31)) {      // (32-bit buffer offsets)
                // Winograd form: 16 products per tile and channel pair instead of 36 (wgrad_wino_bf16x6_kernel.h)
                using Cfg3 = WgwCfg<64, 128>;
                const int strips = wgrad_wino_strips(B, 16, d.wp2, Cfg3::TPS, Cfg3::groups, h->num_cus);
                if (sir_f16_mask() & 32)
                    hipLaunchKernelGGL((conv_wgrad_wino_bf16x6_kernel<64, 128, true>), dim3(Cfg3::groups * strips), dim3(WGW_THREADS), Cfg3::lds_bytes, cside,
                                       (const float*)p.dz3, (const float*)p.a2, p.slab, B, 16, d.wp2);
                else
                hipLaunchKernelGGL((conv_wgrad_wino_bf16x6_kernel<64, 128>), dim3(Cfg3::groups * strips), dim3(WGW_THREADS), Cfg3::lds_bytes, cside,
                                   (const float*)p.dz3, (const float*)p.a2, p.slab, B, 16, d.wp2);
                float* part = p.slab + (size_t)strips * 16 * 128 * 64;
                hipLaunchKernelGGL(wgrad_wino_sum_kernel, dim3((16 * 128 * 64 / 4 + 255) / 256), dim3(256), 0, cside, (const float*)p.slab, strips,
                                   16 * 128 * 64 / 4, part);
                hipLaunchKernelGGL(wgrad_wino_finish_kernel, dim3((128 * 64 + 255) / 256), dim3(256), 0, cside, (const float*)part, 64, 128, g->conv_w[2], unscale);
            } else {
            const size_t ldsx = wgrad_x6_lds_bytes(64, 128, d.wp2);
            if (ldsx > 160 * 1024 || d.wp2 > wgrad_x6_max_w(128)) { sir_set_error("sir_model_train_bwd: t_frames too large for the weight-gradient tile"); return SIR_EUNSUPPORTED; }
            const int nslab3 = d.wg3_blocks;                  // one slab per workgroup
            hipLaunchKernelGGL((conv_wgrad_bf16x6_kernel<64, 128>), dim3(d.wg3_blocks), dim3(512), ldsx, cside, (const float*)p.dz3,
                               (const float*)p.a2, p.slab, 16, d.wp2, d.wg3_rb);
            float* part = p.slab + (size_t)nslab3 * 9 * 128 * 64;
            hipLaunchKernelGGL(wgrad_reduce_partial_kernel, dim3((9 * 128 * 64 / 4 + 255) / 256, WGR_PARTS), dim3(256), 0, cside,
                               (const float*)p.slab, nslab3, 9 * 128 * 64 / 4, part);
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((9 * 128 * 64 + 255) / 256), dim3(256), 0, cside, (const float*)part, WGR_PARTS, 64, 128,
                               g->conv_w[2], unscale);
            }
        }
        {
            // data gradient = a 128 -> 64 convolution with the flipped / transposed taps: the Winograd kernel (16 of 36 products), blocks
            // of 8 x 4 tiles for the 16-row map, raw output (train_prep_kernel of the forward built p.wcb3t)
            SirProfScope prof(h, SIR_K_B_DGRAD3, st);
            Wino2Geo geo3b;
            if (wino2_geo(B, 16, d.wp2, 128, &geo3b) && (sir_wino2_mask() & 4) && (sir_f16_mask() & 4))     // dz3 carries the loss scale: inside fp16's range
                SIR_HIP_TRY((launch_conv_wino2<128, 64, 3, 0, 3, true>(st, &h->attr_wino2[9], (const float*)p.dz3, (const unsigned short*)p.wcb3t, (const float*)nullptr,
                                                                     (const float*)nullptr, p.da2, B, 16, d.wp2, (float2*)nullptr, h->zero_page, h->num_cus)));
            else if (wino2_geo(B, 16, d.wp2, 128, &geo3b) && (sir_wino2_mask() & 4))
                SIR_HIP_TRY((launch_conv_wino2<128, 64, 3>(st, &h->attr_wino2[4], (const float*)p.dz3, (const unsigned short*)p.wcb3t, (const float*)nullptr,
                                                         (const float*)nullptr, p.da2, B, 16, d.wp2, (float2*)nullptr, h->zero_page, h->num_cus)));
            else
                hipLaunchKernelGGL((conv3x3_wino_bf16x6_kernel<128, 64, 2, 3, 1, 0, 4>), dim3(((d.wp2 + 1) / 2 + 3) / 4, 1, B), dim3(256), WINO_LDS_BYTES, st,
                                   (const float*)p.dz3, (const unsigned short*)p.wcb3t, (const float*)nullptr, (const float*)nullptr, p.da2, 16, d.wp2,
                                   8, d.wp3, (float2*)nullptr);
        }
        KCHECK();
    }
    // ---- conv2 block -------------------------------------------------------------------------
    {
        const int ppb = 64;
        const size_t npix = (size_t)B * 16 * d.wp2;
        const int nblk = (int)((npix + ppb - 1) / ppb);
        {
            SirProfScope prof(h, SIR_K_B_BN2, st);
            hipLaunchKernelGGL(bn_bwd_reduce_pooled_kernel, dim3(nblk), dim3(256), 0, st, (const float*)p.a2, (const float*)p.da2,
                               (const float*)p.z2, w->bn_w[1], w->bn_b[1], scale + 32, shift + 32, smean + 32, sinv + 32, p.stats, B, 32,
                               d.wp1, 64, 16, d.wp2, ppb);
            hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(64), dim3(256), 0, st, (const float2*)p.stats, nblk, 64,
                               (double)B * 32 * d.wp1, g->bn_w[1], g->bn_b[1], mdy + 32, mdyx + 32, unscale);
            hipLaunchKernelGGL(bn_bwd_dz_kernel<false>, dim3(grid_for((size_t)B * 16 * ((d.wp1 + 1) / 2) * 16)), dim3(256), 0, st, (const float*)p.z2,
                               (const float*)p.da2, scale + 32, shift + 32, smean + 32, sinv + 32, mdy + 32, mdyx + 32, p.dz2, B, 32,
                               d.wp1, 64, 16, d.wp2);
        }
        if (fork_conv) {
            SIR_HIP_TRY(hipEventRecord(h->bwd_ev[2], st));
            SIR_HIP_TRY(hipStreamWaitEvent(side, h->bwd_ev[2], 0));
        }
        {
            SirProfScope prof(h, SIR_K_B_WGRAD2, cside);
            if ((sir_wgw_mask() & 1) && (size_t)B * 32 * d.wp1 * 64 * 4 < ((size_t)1 << 31)) {
                using Cfg2 = WgwCfg<32, 64>;
                const int strips = wgrad_wino_strips(B, 32, d.wp1, Cfg2::TPS, Cfg2::groups, h->num_cus);
                if (sir_f16_mask() & 32)
                    hipLaunchKernelGGL((conv_wgrad_wino_bf16x6_kernel<32, 64, true>), dim3(Cfg2::groups * strips), dim3(WGW_THREADS), Cfg2::lds_bytes, cside,
                                       (const float*)p.dz2, (const float*)p.a1, p.slab, B, 32, d.wp1);
                else
                hipLaunchKernelGGL((conv_wgrad_wino_bf16x6_kernel<32, 64>), dim3(Cfg2::groups * strips), dim3(WGW_THREADS), Cfg2::lds_bytes, cside,
                                   (const float*)p.dz2, (const float*)p.a1, p.slab, B, 32, d.wp1);
                float* part = p.slab + (size_t)strips * 16 * 64 * 32;
                hipLaunchKernelGGL(wgrad_wino_sum_kernel, dim3((16 * 64 * 32 / 4 + 255) / 256), dim3(256), 0, cside, (const float*)p.slab, strips,
                                   16 * 64 * 32 / 4, part);
                hipLaunchKernelGGL(wgrad_wino_finish_kernel, dim3((64 * 32 + 255) / 256), dim3(256), 0, cside, (const float*)part, 32, 64, g->conv_w[1], unscale);
            } else {
            const size_t ldsx = wgrad_x6_lds_bytes(32, 64, d.wp1);
            if (ldsx > 160 * 1024 || d.wp1 > wgrad_x6_max_w(64)) { sir_set_error("sir_model_train_bwd: t_frames too large for the weight-gradient tile"); return SIR_EUNSUPPORTED; }
            const int nslab2 = d.wg2_blocks;                  // one slab per workgroup (its four k-split waves add up in LDS)
            hipLaunchKernelGGL((conv_wgrad_bf16x6_kernel<32, 64>), dim3(d.wg2_blocks), dim3(512), ldsx, cside, (const float*)p.dz2,
                               (const float*)p.a1, p.slab, 32, d.wp1, d.wg2_rb);
            float* part = p.slab + (size_t)nslab2 * 9 * 64 * 32;
            hipLaunchKernelGGL(wgrad_reduce_partial_kernel, dim3((9 * 64 * 32 / 4 + 255) / 256, WGR_PARTS), dim3(256), 0, cside,
                               (const float*)p.slab, nslab2, 9 * 64 * 32 / 4, part);
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((9 * 64 * 32 + 255) / 256), dim3(256), 0, cside, (const float*)part, WGR_PARTS, 32, 64,
                               g->conv_w[1], unscale);
            }
        }
        if (fork_conv) SIR_HIP_TRY(hipEventRecord(h->bwd_ev[3], side));     // (the side stream's last launch)
        {
            SirProfScope prof(h, SIR_K_B_DGRAD2, st);
            Wino2Geo geo2b;
            if (wino2_geo(B, 32, d.wp1, 64, &geo2b) && (sir_wino2_mask() & 8) && (sir_f16_mask() & 16))
                SIR_HIP_TRY((launch_conv_wino2<64, 32, 3, 0, 3, true>(st, &h->attr_wino2[10], (const float*)p.dz2, (const unsigned short*)p.wcb2t, (const float*)nullptr,
                                                                    (const float*)nullptr, p.da1, B, 32, d.wp1, (float2*)nullptr, h->zero_page, h->num_cus)));
            else
            hipLaunchKernelGGL((conv3x3_bf16x6_ns_kernel<64, 32, 4, 2, 2, 0, 4>), dim3(d.c2gx, 1, B), dim3(256), conv_ns_lds_bytes(4, 2), st,
                               (const float*)p.dz2, (const unsigned short*)p.wcb2t, (const float*)nullptr, (const float*)nullptr, p.da1, 32, d.wp1,
                               16, d.wp2, (float2*)nullptr);
        }
        KCHECK();
    }
    // ---- conv1 block ------------------------------------------------------------------------
    {
        // ONE recompute pass: (sum dy, sum dy*xhat, sum dy*x_tap) per channel; the mean terms of dz = s (dy - m1 - xhat m2) and
        // with them the rest of dW1 are closed forms in the input moments of the forward (conv1_bwd_finalize_kernel, in double)
        SirProfScope prof(h, SIR_K_B_CONV1, st);
        const dim3 g1(d.c1gx, d.c1gy, B);
        const int nblk = d.c1gx * d.c1gy * B;
        hipLaunchKernelGGL(conv1_bwd_kernel<2>, g1, dim3(256), 0, st, feats, w->conv_w[0], (const float*)p.da1, scale, shift,
                           smean, sinv, (const float*)nullptr, (const float*)nullptr, c1part, 64, T, 32, d.wp1);
        float* c1tmp = (float*)p.stats;           // [128][352] partial column sums, then [352] totals behind them
        float* c1tot = c1tmp + 128 * 352;
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((352 + 63) / 64, 128), dim3(256), 0, st, (const float*)c1part, nblk, 352,
                           352, c1tmp);
        hipLaunchKernelGGL(colsum_kernel, dim3((352 + 63) / 64), dim3(256), 0, st, (const float*)c1tmp, 128, 352, 352, c1tot);
        hipLaunchKernelGGL(conv1_bwd_finalize_kernel, dim3(1), dim3(320), 0, st, (const float*)c1tot, (const double*)p.c1m,
                           w->conv_w[0], scale, smean, sinv, (double)B * 64 * T, g->bn_w[0], g->bn_b[0], g->conv_w[0], unscale);
        KCHECK();
    }
    if (two) SIR_HIP_TRY(hipStreamWaitEvent(st, h->bwd_ev[3], 0));   // join: every gradient is final on the caller's stream
    return SIR_OK;
}

extern "C" int sir_adam_step(sir_handle* h, int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                             float* const* exp_avg_sq, const int64_t* sizes, int step, float lr, float beta1, float beta2,
                             float eps, float weight_decay, void* stream_) {
    if (!h || !params || !grads || !exp_avg || !exp_avg_sq || !sizes) { sir_set_error("sir_adam_step: NULL argument"); return SIR_EINVAL; }
    if (n_tensors < 1 || n_tensors > SIR_ADAM_MAX_TENSORS || step < 1) { sir_set_error("sir_adam_step: n_tensors=%d step=%d", n_tensors, step); return SIR_EINVAL; }
    AdamTensors ts;
    int blocks = 0;
    for (int i = 0; i < n_tensors; ++i) {
        ts.p[i] = params[i]; ts.g[i] = grads[i]; ts.m[i] = exp_avg[i]; ts.v[i] = exp_avg_sq[i]; ts.n[i] = sizes[i];
        ts.first_block[i] = blocks;
        blocks += (int)((sizes[i] + SIR_ADAM_CHUNK - 1) / SIR_ADAM_CHUNK);
    }
    ts.first_block[n_tensors] = blocks;
    ts.count = n_tensors;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    SirProfScope prof(h, SIR_K_ADAM, (hipStream_t)stream_);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, ts, lr, beta1, beta2, eps, weight_decay,
                       (float)bc1, (float)sqrt(bc2));
    KCHECK();
    return SIR_OK;
}
