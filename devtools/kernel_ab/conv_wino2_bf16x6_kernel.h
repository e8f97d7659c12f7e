// conv 3x3 (pad 1) as Winograd F(2x2, 3x3) on the bf16 matrix cores with bf16x6 products (fp32 accuracy), second
// generation: ONE persistent 512-thread workgroup per CU whose two wave groups alternate between the input transform
// (VALU + LDS) and the matrix phase (MFMA), so that on every SIMD one wave feeds the matrix pipe while its partner
// transforms -- the first generation (conv_wino_bf16x6_kernel.h: 4 waves, loads -> transform -> barrier -> MFMA -> barrier
// per 16-channel chunk, 3 workgroups per CU) ran those phases back to back and its time was their SUM (timing knock-outs
// on MI355X at batch 256: patch loads 34 us, transform 29, weights 16, MFMA 18, stores 7, skeleton 43 of 138).
//
//   Y = A^T [ sum_cin (G g G^T) . (B^T d B) ] A     d: 4x4 input patch, g: 3x3 taps, Y: 2x2 outputs (= one pooling window)
//   per frequency f = 4 i + j of the transform one GEMM  M_f[tile][cout] = sum_cin V_f[tile][cin] U_f[cin][cout].
//
// Task = 32 tiles (8 tile rows x 4 tile columns; columns are numbered across the whole batch, g = image * TW + tx, so a
// task may straddle two images and no column is wasted) x 64 output channels (layers with 128: two tasks per block).
//   matrix role   wave (i = wv >> 1, n = wv & 1): transform row i (4 frequencies), channel slice 32 n..: 4 accumulators M[j]
//                 (64 registers) that live across all chunks, no per-chunk folding; U fragments (3 x 16 B per lane) are
//                 streamed from L2 one frequency ahead, V fragments come from LDS (ds_read_b128, 1 KB runs).
//                 (A 64-tile task -- two accumulator sets per wave, every U fragment used twice -- was built first: 128 live
//                 accumulator registers beside the transform's working set do not fit the 256 of a two-wave SIMD, hipcc
//                 spilled three accumulator tiles into the matrix loop.)
//   groups        A = waves 0-3 (rows i = 0, 1), B = waves 4-7 (rows 2, 3).  A group transforms the 8 frequencies it will
//                 consume itself into ITS half of the V area, then multiplies; the groups run one phase apart:
//                     phase 2c     A: transform chunk c      B: MFMA chunk c - 1
//                     phase 2c + 1 A: MFMA chunk c           B: transform chunk c
//                 one __syncthreads per phase; V is single-buffered (each half is written and read by the same group).
//   raw patches   the 18 x 10 pixel region of a chunk (16 channels) arrives by LDS-DMA (global_load_lds_dwordx4: no VGPRs,
//                 asynchronous) into a two-slot ring, issued by group B at the START of its matrix phase two chunks ahead --
//                 its later weight loads return in order behind them, so the pieces have landed by the end of the phase
//                 without an extra wait.  LDS image: pixel (lr, lc) at position lr * 11 + lc + ((lr >> 1) & 1), its four
//                 16-byte channel groups XOR-swizzled by ((lr >> 2) & 1) << 1: the transform's ds_read_b128 lane groups
//                 (8 tiles x 2 channel groups) hit 16 different slots of the 256-byte bank line for every patch offset
//                 (brute-forced, devtools/kernel_ab/wino2_lds_layout.py) while a DMA piece still reads whole 64-byte pixels.
//   epilogue      column inverse transform in registers, row transform across the four waves of a channel slice through
//                 LDS (each wave finishes one tile column of the task: it receives 3 x 2 KB), then BN + ReLU + 2x2 max
//                 (the 2x2 outputs of a tile ARE the pooling window) or raw outputs (+ per-task channel statistics).
// Weights: prep_conv_w_wino_bf16x3 layout wpb[plane][chunk * 16 + f][cout][16 ch] with column j = 3 negated (shared with
// the first generation); data gradients use the same kernel on the transposed / flipped taps (prep_conv_wT_wino_bf16x3).
// ---------------------------------------------------------------------------------------------------------------------------
// STATUS (round 3): EXPERIMENT, NOT LINKED INTO libsir_hip.so.  Bit-compatible with the direct kernel on every shape and
// output mode (devtools/kernel_ab/bench_conv.hip `wino2`: <= 1.4e-5 on |out| <= 8.6, ragged shapes, GRU layout + planes,
// statistics, data gradient), but SLOWER than what it was meant to replace.  Three structures were built and timed on MI355X
// at batch 256 (conv2 32 -> 64; first-generation Winograd kernel: 138 us, direct: 150 us):
//   1. two wave groups alternating transform / matrix phases, 64-tile tasks (128 accumulator registers per wave): hipcc
//      spilled three accumulator tiles into the matrix loop (456-720 B of scratch per lane); not run;
//   2. the same with 32-tile tasks (64 accumulator registers, no spills): 199 us.  In-kernel s_memtime stamps: a transform
//      phase takes 1550-1950 cycles (245 instructions of which 74 are v_mov_b64 shuffles for the packed adds, LDS reads
//      issued three at a time because both roles' state is live in every wave), a matrix phase 1800-2100 cycles for 24 MFMAs
//      (768) -- weights one frequency (192 cycles) ahead of an L2 round trip of 500-800; with the weights a whole phase
//      ahead (48 more registers) the spills came back: 285 us;
//   3. this file: producer waves (transform only) and consumer waves (matrix only, 8 accumulators), double-buffered V, one
//      barrier per chunk: 360 us -- the consumer keeps 128 accumulator + 64 pending-output + 48 weight-ring registers, hipcc
//      spills the weight ring into the matrix loop and every scratch reload queues behind the outstanding global loads
//      (vector memory returns in order).
// What it would take: hand-scheduled consumer code (weights through LDS-DMA instead of registers, the row exchange
// finished inside the step so that no output registers survive a barrier) -- i.e. an assembly-level kernel, not another
// hipcc variant.  The first-generation kernel stays the product path for conv2; conv3 and the data gradients stay direct.
// ---------------------------------------------------------------------------------------------------------------------------
#pragma once
#include "../../speech-intent-recognizer_amd/csrc/conv_wino_bf16x6_kernel.h"

constexpr int W2_RS = 11;                                   // pixel positions per raw row (10 + skew)
constexpr int W2_NPOS = 18 * W2_RS;
constexpr int W2_RAW_PIECES = (W2_NPOS * 4 + 63) / 64;      // 13 DMA pieces of 1 KB
constexpr int W2_RAW_BYTES = W2_RAW_PIECES * 1024;          // 13,312
constexpr int W2_PLB = 16 * 1024;                           // bytes per V plane: [f][1 KB]
constexpr int W2_V_BYTES = 3 * W2_PLB;                      // 49,152 per chunk; two buffers
constexpr int W2_XCH_BYTES = 2 * 16 * 1024;                 // hand-over of the partial outputs between the two consumer waves of a channel slice
constexpr int W2_LDS_BYTES = 2 * W2_V_BYTES + 2 * W2_RAW_BYTES + W2_XCH_BYTES;   // 157,696
constexpr int W2_PPW = (W2_RAW_PIECES + 3) / 4;             // DMA pieces per wave of group B

struct Wino2Geo {
    int H, W;            // input = output map (pixels)
    int TW;              // tile columns per image = ceil(W / 2)
    int NG;              // tile columns of the batch = B * TW
    int RBN;             // 8-row tile blocks per image = (H / 2) / 8
    int NS;              // spatial tasks = RBN * ceil(NG / 4)
    int Hp, Wp;          // pooled map (OUT_MODE 0 / 1)
    int B;
};
static inline bool wino2_geo(int B, int H, int W, Wino2Geo* g) {
    g->B = B; g->H = H; g->W = W; g->TW = (W + 1) / 2; g->NG = B * g->TW; g->RBN = H / 16;
    g->NS = g->RBN * ((g->NG + 3) / 4); g->Hp = H / 2; g->Wp = W / 2;
    // 32-bit element offsets and whole 8-row blocks (the model's maps are 32 and 16 rows high)
    return H % 16 == 0 && W >= 1 && B >= 1;
}

// OUT_MODE 0: pooled NHWC (BN + ReLU + max), 1: pooled in the GRU layout [b][tx][co * Hp + ty] (+ its bf16x3 planes through
// `stats`), 2: raw NHWC + per-task channel statistics (float2 {sum, sum of squares} at stats[task * COUT + co]), 3: raw NHWC
// DBG (devtools/kernel_ab/bench_conv.hip only): s_memtime stamps of the first producer and the first consumer wave of workgroup 0
__device__ long long w2_dbg_stamps[2][32];
template <int CIN, int COUT, int OUT_MODE, int DBG = 0>
__global__ __launch_bounds__(512, 2) void conv3x3_wino2_bf16x6_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ wpb, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ out, Wino2Geo geo, float2* __restrict__ stats) {
    constexpr int NCH = CIN / 16, G = NCH * 16, NCHO = COUT / 64;
    static_assert(CIN % 32 == 0 && COUT % 64 == 0, "even chunk count (ring parity) and 64-channel tasks");
    extern __shared__ __attribute__((aligned(1024))) unsigned char w2s[];
    unsigned char* const vbuf = w2s;                                    // [2][3 planes][16 f][1 KB]
    unsigned char* const rawbuf = w2s + 2 * W2_V_BYTES;                 // [2][13 KB]
    float4* const xch = reinterpret_cast<float4*>(w2s + 2 * W2_V_BYTES + 2 * W2_RAW_BYTES);   // [2 slices][16 float4][64 lanes]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool producer = wv < 4;
    const int wg = wv & 3;
    const int H = geo.H, W = geo.W, TW = geo.TW, NG = geo.NG, RBN = geo.RBN;
    const int ntask_s = (geo.NS - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // spatial tasks of this workgroup
    const int ntask = ntask_s * NCHO;
    if (ntask <= 0) return;
    const int nsteps = ntask * NCH;                                     // chunks of this workgroup; step s: producers chunk s, consumers chunk s - 1

    auto task_geo = [&](int lt, int& g0, int& ty0, int& ch, int& s) {
        s = (int)blockIdx.x + (lt / NCHO) * (int)gridDim.x;
        ch = lt % NCHO;
        const int rb = s % RBN, cb = s / RBN;
        g0 = 4 * cb; ty0 = 8 * rb;
    };
    int nst = 0;
    auto stamp = [&]() {
        if (DBG && blockIdx.x == 0 && lane == 0 && wg == 0 && nst < 32) w2_dbg_stamps[producer ? 0 : 1][nst] = __builtin_amdgcn_s_memtime();
        ++nst;
    };

    if (producer) {
        // ================= producers: raw patches (LDS-DMA) -> B^T d B -> bf16x3 -> V[step & 1] =================================
        // wave wg = (row pair tR of the transform, 8-channel half tH); lane = (tile tm, 4-channel group tP1)
        const int tR = wg >> 1, tH = wg & 1, tm = lane >> 1, tP1 = lane & 1;
        const int tty = tm & 7, ttx = tm >> 3, tpart = 2 * tH + tP1;
        // element offsets (without the chunk term) of this lane's DMA pieces for a task: piece k = wg + 4 i, slot = 64 k + lane
        auto raw_offsets = [&](int g0, int ty0, unsigned (&off)[W2_PPW]) {
#pragma unroll
            for (int ii = 0; ii < W2_PPW; ++ii) {
                const int slot = 64 * (wg + 4 * ii) + lane, pos = slot >> 2, sp = slot & 3;
                int lr = pos / W2_RS;
                int lc = pos - lr * W2_RS - ((lr >> 1) & 1);
                lr = min(lr, 17);
                lc = min(max(lc, 0), 9);
                const int part = sp ^ (((lr >> 2) & 1) << 1);
                const int gy = min(max(2 * ty0 - 1 + lr, 0), H - 1);
                const int P = min(max(2 * g0 - 1 + lc, 0), 2 * NG - 1);
                const int bb = P / (2 * TW), px = min(P - bb * 2 * TW, W - 1);
                off[ii] = (unsigned)(((bb * H + gy) * W + px) * CIN + part * 4);
            }
        };
        auto raw_issue = [&](const unsigned (&off)[W2_PPW], int c, int slot_buf) {
#pragma unroll
            for (int ii = 0; ii < W2_PPW; ++ii) {
                const int k = __builtin_amdgcn_readfirstlane(wg) + 4 * ii;
                if (k < W2_RAW_PIECES)
                    __builtin_amdgcn_global_load_lds((sir_gptr_t)(x + off[ii] + c * 16), (sir_lptr_t)(rawbuf + slot_buf * W2_RAW_BYTES + k * 1024), 16, 0, 0);
            }
        };
        int g0, ty0, ch, s_idx;
        task_geo(0, g0, ty0, ch, s_idx);
        unsigned roff[W2_PPW];
        raw_offsets(g0, ty0, roff);
        raw_issue(roff, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                // raw chunk 0 has landed
#pragma unroll 1
        for (int s = 0; s <= nsteps; ++s) {
            stamp();
            if (s < nsteps) {
                const int lt = s / NCH, c = s - lt * NCH;
                // raw chunk s + 1 into the ring slot that chunk s - 1 left (its last reads were before the barrier): one step to land
                if (s + 1 < nsteps) {
                    if (c + 1 < NCH) raw_issue(roff, c + 1, (s + 1) & 1);
                    else {
                        task_geo(lt + 1, g0, ty0, ch, s_idx);
                        raw_offsets(g0, ty0, roff);                     // roff now belongs to the NEXT task; this step's geometry is in `cur` below
                        raw_issue(roff, 0, (s + 1) & 1);
                    }
                }
                // geometry of THIS step's task (task_geo again: roff / g0 may already be the next task's)
                int cg0, cty0, cch, cs;
                task_geo(lt, cg0, cty0, cch, cs);
                const int gcol = cg0 + ttx;
                const int timg = gcol / TW, ttxg = gcol - timg * TW;
                const int tyg = cty0 + tty;
                const bool tv = gcol < NG;
                const unsigned char* rb = rawbuf + (s & 1) * W2_RAW_BYTES;
                unsigned char* const vdst = vbuf + (s & 1) * W2_V_BYTES + (8 * tR) * 1024 + tH * 512 + tm * 16 + tP1 * 8;   // + plane * PLB + (4 il + j) * 1024
                // patch rows tR .. tR + 2 (row pair 0: i = 0: d0 - d2, i = 1: d1 + d2; pair 1 (rows 1,2,3): i = 2: d2 - d1, i = 3: d1 - d3)
                float4 L[3][4];
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    const int lr = 2 * tty + tR + rr, gy = 2 * tyg - 1 + tR + rr;
                    const bool rv = tv && gy >= 0 && gy < H;
                    const unsigned char* rp = rb + (lr * W2_RS + 2 * ttx + ((lr >> 1) & 1)) * 64 + ((tpart ^ (((lr >> 2) & 1) << 1)) * 16);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int gx = 2 * ttxg - 1 + j;
                        float4 v = *reinterpret_cast<const float4*>(rp + j * 64);
                        if (!(rv && gx >= 0 && gx < W)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                        L[rr][j] = v;
                    }
                }
#pragma unroll
                for (int il = 0; il < 2; ++il) {
                    float4 R[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 a0 = L[0][j], a1 = L[1][j], a2 = L[2][j];
                        if (tR == 0) R[j] = il == 0 ? make_float4(a0.x - a2.x, a0.y - a2.y, a0.z - a2.z, a0.w - a2.w)
                                                    : make_float4(a1.x + a2.x, a1.y + a2.y, a1.z + a2.z, a1.w + a2.w);
                        else         R[j] = il == 0 ? make_float4(a1.x - a0.x, a1.y - a0.y, a1.z - a0.z, a1.w - a0.w)
                                                    : make_float4(a0.x - a2.x, a0.y - a2.y, a0.z - a2.z, a0.w - a2.w);
                    }
                    const float4 V[4] = {make_float4(R[0].x - R[2].x, R[0].y - R[2].y, R[0].z - R[2].z, R[0].w - R[2].w),
                                         make_float4(R[1].x + R[2].x, R[1].y + R[2].y, R[1].z + R[2].z, R[1].w + R[2].w),
                                         make_float4(R[2].x - R[1].x, R[2].y - R[1].y, R[2].z - R[1].z, R[2].w - R[1].w),
                                         make_float4(R[1].x - R[3].x, R[1].y - R[3].y, R[1].z - R[3].z, R[1].w - R[3].w)};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        uint2 sh, sm, sl;
                        split3_quad(V[j], sh, sm, sl);
                        unsigned char* d = vdst + (4 * il + j) * 1024;
                        *reinterpret_cast<uint2*>(d) = sh;
                        *reinterpret_cast<uint2*>(d + W2_PLB) = sm;
                        *reinterpret_cast<uint2*>(d + 2 * W2_PLB) = sl;
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the raw pieces issued at the top of this step
            }
            stamp();
            __syncthreads();
        }
        return;
    }

    // ================= consumers: M_f += V_f U_f on the matrix cores, inverse transform, epilogue ==================================
    // wave (n = wg & 1, rp = wg >> 1): channel slice 32 n.., transform rows 2 rp, 2 rp + 1 (8 frequencies, 8 accumulators)
    const int mn = wg & 1, rp = wg >> 1, m = lane & 31, h = lane >> 5;
    const uint4* const wp4 = reinterpret_cast<const uint4*>(wpb) + (size_t)(mn * 32 + m) * 2 + h;     // + (p * G + g) * COUT * 2 + ch * 128
    uint4 wq[4][3];                                                     // ring of four frequencies, refilled four frequencies ahead
    auto load_w = [&](int gidx, int chh, uint4 (&q)[3]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) q[p] = wp4[(p * G + gidx) * (COUT * 2) + chh * 128];
    };
    int g0, ty0, ch, s_idx;
    task_geo(0, g0, ty0, ch, s_idx);
#pragma unroll
    for (int k = 0; k < 4; ++k) load_w(8 * rp + k, ch, wq[k]);
    f32x16 acc[8];
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.0f;
    __syncthreads();                                                    // (pairs with the producers' "raw chunk 0 has landed")
    bool pending = false;                                               // rp == 0: a hand-over of the previous task waits in LDS
    int pg0 = 0, pty0 = 0, pch = 0, ps = 0;
    float Yp[2][2][16];

    // finish a task on the rp == 0 wave: add the partner's partial outputs, BN + ReLU + pool / raw stores (+ statistics)
    auto finish = [&]() {
        const float4* src = xch + (size_t)mn * 1024 + lane;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = src[((a * 2 + b) * 4 + q) * 64];
                    Yp[a][b][4 * q] += v.x; Yp[a][b][4 * q + 1] += v.y; Yp[a][b][4 * q + 2] += v.z; Yp[a][b][4 * q + 3] += v.w;
                }
        const int co = pch * 64 + mn * 32 + m;
        float ssum = 0.0f, ssq = 0.0f;
        float sc_ = 1.0f, sh_ = 0.0f;
        if (OUT_MODE <= 1) { sc_ = scale[co]; sh_ = shift[co]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {                                   // accumulator registers 4 q .. 4 q + 3: tile column q, tile rows 4 h + e
            const int gc = pg0 + q;
            const int img = gc / TW, tx = gc - img * TW;
            const bool tvalid = gc < NG;
            if (OUT_MODE <= 1) {
                float pooled[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = 0.0f;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) v = fmaxf(v, fmaf(Yp[a][b][4 * q + e], sc_, sh_));
                    pooled[e] = v;
                }
                if (tvalid && tx < geo.Wp) {
                    if (OUT_MODE == 0) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            out[(((size_t)img * geo.Hp + pty0 + 4 * h + e) * geo.Wp + tx) * COUT + co] = pooled[e];
                    } else {
                        const size_t oidx = ((size_t)img * geo.Wp + tx) * (COUT * geo.Hp) + (size_t)co * geo.Hp + pty0 + 4 * h;
                        const float4 v4 = make_float4(pooled[0], pooled[1], pooled[2], pooled[3]);
                        *reinterpret_cast<float4*>(out + oidx) = v4;
                        if (stats) {           // bf16x3 planes [3][B * Wp][COUT * Hp] of the following GEMM's A operand
                            unsigned short* planes = reinterpret_cast<unsigned short*>(stats);
                            const size_t plane = (size_t)geo.B * geo.Wp * (COUT * geo.Hp);
                            uint2 hh, mm, ll;
                            split3_quad(v4, hh, mm, ll);
                            *reinterpret_cast<uint2*>(planes + oidx) = hh;
                            *reinterpret_cast<uint2*>(planes + plane + oidx) = mm;
                            *reinterpret_cast<uint2*>(planes + 2 * plane + oidx) = ll;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const int gy = 2 * (pty0 + 4 * h + e) + a, gx = 2 * tx + b;
                            if (tvalid && gx < W) {
                                const float v = Yp[a][b][4 * q + e];
                                out[(((size_t)img * H + gy) * W + gx) * COUT + co] = v;
                                ssum += v;
                                ssq = fmaf(v, v, ssq);
                            }
                        }
            }
        }
        if (OUT_MODE == 2 && stats) {                                   // the wave owns all 32 tiles of its 32 channels: no cross-wave reduction
            ssum += __shfl_xor(ssum, 32);
            ssq += __shfl_xor(ssq, 32);
            if (h == 0) stats[(size_t)ps * COUT + co] = make_float2(ssum, ssq);
        }
    };

#pragma unroll 1
    for (int s = 0; s <= nsteps; ++s) {
        stamp();
        if (rp == 0 && pending) { finish(); pending = false; }
        if (s >= 1) {
            const int sc = s - 1, lt = sc / NCH, c = sc - lt * NCH;
            const bool last = c + 1 == NCH;
            int g0n = g0, ty0n = ty0, chn = ch, sn = s_idx;
            if (last && lt + 1 < ntask) task_geo(lt + 1, g0n, ty0n, chn, sn);
            const unsigned char* abase = vbuf + (sc & 1) * W2_V_BYTES + h * 512 + m * 16 + (8 * rp) * 1024;
            const int gcur = c * 16 + 8 * rp, gnxt = (last ? 0 : c + 1) * 16 + 8 * rp;
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                bf16x8 a[3], bq[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(abase + p * W2_PLB + f * 1024));
                    bq[p] = __builtin_bit_cast(bf16x8, wq[f & 3][p]);
                }
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // small terms first
#pragma unroll
                for (int t6 = 0; t6 < 6; ++t6) acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t6]], bq[PB[t6]], acc[f], 0, 0, 0);
                // four frequencies ahead (768 cycles of matrix work: about one L2 round trip under load)
                if (f < 4) load_w(gcur + f + 4, ch, wq[f & 3]);
                else load_w(gnxt + f - 4, last ? chn : ch, wq[f & 3]);
            }
            if (last) {
                // column inverse transform (U_{i3} is stored negated), then this wave's share of the row transform:
                //   rows 0, 1 (rp = 0): Y[0][b] = W0 + W1, Y[1][b] = W1;   rows 2, 3 (rp = 1): Y[0][b] = W2, Y[1][b] = -W2 - W3
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x16 w0, w1;
                    if (b == 0) { w0 = acc[0] + acc[1] + acc[2]; w1 = acc[4] + acc[5] + acc[6]; }
                    else        { w0 = acc[1] - acc[2] + acc[3]; w1 = acc[5] - acc[6] + acc[7]; }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        Yp[0][b][r] = rp == 0 ? w0[r] + w1[r] : w0[r];
                        Yp[1][b][r] = rp == 0 ? w1[r] : -w0[r] - w1[r];
                    }
                }
                if (rp == 1) {
                    float4* dst = xch + (size_t)mn * 1024 + lane;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                dst[((a * 2 + b) * 4 + q) * 64] = make_float4(Yp[a][b][4 * q], Yp[a][b][4 * q + 1], Yp[a][b][4 * q + 2], Yp[a][b][4 * q + 3]);
                } else {
                    pending = true; pg0 = g0; pty0 = ty0; pch = ch; ps = s_idx;
                }
#pragma unroll
                for (int f = 0; f < 8; ++f)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[f][r] = 0.0f;
                g0 = g0n; ty0 = ty0n; ch = chn; s_idx = sn;
            }
        }
        stamp();
        __syncthreads();
    }
    if (rp == 0 && pending) finish();
}

// `attr_done`: the caller's per-device latch of the dynamic-LDS opt-in of THIS instantiation (sir_handle::attr_wino2[...])
template <int CIN, int COUT, int OUT_MODE, int DBG = 0>
static inline hipError_t launch_conv_wino2(hipStream_t st, bool* attr_done, const float* x, const unsigned short* wpb, const float* scale,
                                           const float* shift, float* out, int B, int H, int W, float2* stats, int max_wg = 256) {
    Wino2Geo g;
    if (!wino2_geo(B, H, W, &g)) return hipErrorInvalidValue;
    if (!*attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_wino2_bf16x6_kernel<CIN, COUT, OUT_MODE, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS_BYTES);
        if (e != hipSuccess) return e;
        *attr_done = true;
    }
    const int nwg = g.NS < max_wg ? g.NS : max_wg;
    hipLaunchKernelGGL((conv3x3_wino2_bf16x6_kernel<CIN, COUT, OUT_MODE, DBG>), dim3(nwg), dim3(512), W2_LDS_BYTES, st, x, wpb, scale, shift, out, g, stats);
    return hipGetLastError();
}
