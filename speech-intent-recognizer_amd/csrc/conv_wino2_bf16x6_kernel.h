// conv 3x3 (pad 1) as Winograd F(2x2, 3x3) on the bf16 matrix cores with bf16x6 products (fp32 accuracy), second
// generation: ONE persistent 768-thread workgroup per CU with SPECIALISED waves -- 4 producers (input transform: VALU + LDS)
// and 8 consumers (matrix cores) -- so that on every SIMD one producer wave transforms the next 16-channel chunk while two
// consumer waves multiply the current one.  The first generation (conv_wino_bf16x6_kernel.h: 4 waves, loads -> transform ->
// barrier -> MFMA -> barrier per chunk, 3 workgroups per CU) ran those phases back to back and its time was their SUM (timing
// knock-outs on MI355X at batch 256, conv2: patch loads 34 us, transform 29, weights 16, MFMA 18, stores 7, skeleton 43 of 138).
//
//   Y = A^T [ sum_cin (G g G^T) . (B^T d B) ] A     d: 4x4 input patch, g: 3x3 taps, Y: 2x2 outputs (= one pooling window)
//   per frequency f = 4 i + j of the transform one GEMM  M_f[tile][cout] = sum_cin V_f[tile][cin] U_f[cin][cout].
//
// Task = 32 tiles (8 tile rows x 4 tile columns; columns are numbered across the whole batch, g = image * TW + tx, so a task
// may straddle two images and no column is wasted) x 64 output channels (layers with 128: two tasks per block); a workgroup
// walks its tasks chunk by chunk, step s: producers chunk s, consumers chunk s - 1, ONE barrier per step.
//   producers     raw 18 x 10 pixel patches of a chunk arrive by LDS-DMA (global_load_lds_dwordx4: no VGPRs, asynchronous) into
//                 a three-slot ring, issued two steps ahead; rows above / below the image, the pad column of an odd-width map
//                 and tile columns past the batch are sourced from a zero page, so the transform needs selects only in the one
//                 task in ~12 that holds an image boundary column.  LDS image: pixel (lr, lc) at position lr * 11 + lc +
//                 ((lr >> 1) & 1), its four 16-byte channel groups XOR-swizzled by ((lr >> 2) & 1) << 1: the ds_read_b128
//                 lane groups (8 tiles x 2 channel groups) hit 16 different slots of the 256-byte bank line for every patch
//                 offset (brute-forced: devtools/kernel_ab/wino2_lds_layout.py) while a DMA piece still reads whole 64-byte
//                 pixels.  B^T d B, the three-way bf16 split and 24 ds_write_b64 into V[step & 1] follow.  The producers' LDS
//                 accesses are inline asm and the barrier is a bare s_barrier behind s_waitcnt lgkmcnt(0): hipcc orders every LDS
//                 access it can see behind ALL outstanding LDS-DMA (vmcnt(0)) and __syncthreads() drains vmcnt as well.
//   consumers     wave (i, n): transform row i (4 frequencies), channel slice 32 n..: 4 accumulators that live across all chunks
//                 (no per-chunk folding); V fragments from LDS (ds_read_b128, 1 KB runs), U fragments (3 x 16 B per lane) streamed
//                 from L2 one chunk ahead, two frequencies at a time and fenced (left alone hipcc sinks all twelve loads to the
//                 end of the step and the next step starts with an L2 round trip).
//   epilogue      column inverse transform in registers, row transform across the four consumer waves of a channel slice through
//                 the V buffer the task's last chunk just left (each wave finishes one tile column: it receives 3 x 2 KB; two
//                 extra barriers per task), then BN + ReLU + 2x2 max (the 2x2 outputs of a tile ARE the pooling window) or raw
//                 outputs (+ channel statistics, accumulated in registers over the workgroup's tasks: 4 blocks per workgroup).
// Weights: prep_conv_w_wino_bf16x3 layout wpb[plane][chunk * 16 + f][cout][16 ch] with column j = 3 negated (shared with the
// first generation); data gradients use the same kernel on the transposed / flipped taps (prep_conv_wT_wino_bf16x3).
// Measured (devtools/kernel_ab/bench_conv.hip `wino2`, batch 256, four rotating inputs, one box): conv2 143 us against 184 direct
// (first generation ~0.90 of direct), conv3 131 / 160, conv3 data gradient 124 / 183; outputs within 1.4e-5 of the direct kernel
// on |out| <= 8.6 for every mode, ragged shapes included.  Timing knock-outs say what bounds it now: the 96 KB of U fragments a
// step pulls through the CU's vector-memory path (a fragment serves ONE 32-tile accumulator: 64-tile tasks need 128
// accumulator registers per consumer, more than the 168 of a three-wave SIMD) and the task epilogue (~2 us of exchange, output
// arithmetic and stores per task with the producers parked at its barriers); transform and MFMAs hide behind each other.
// Earlier structures of this kernel (two wave groups alternating roles with 64- and 32-tile tasks, 4 producers + 4 two-row
// consumers) were slower than the first generation -- spilled accumulators, weights one frequency ahead of an L2 round trip,
// LDS reads serialised behind DMA waits; their numbers are in DESIGN.md section 4.
#pragma once
#include "conv_wino_bf16x6_kernel.h"

constexpr int W2_RS = 11;                                   // pixel positions per raw row (10 + skew)
constexpr int W2_NPOS = 18 * W2_RS;
constexpr int W2_RAW_PIECES = (W2_NPOS * 4 + 63) / 64;      // 13 DMA pieces of 1 KB
constexpr int W2_RAW_BYTES = W2_RAW_PIECES * 1024;          // 13,312
constexpr int W2_PLB = 16 * 1024;                           // bytes per V plane: [f][1 KB]
constexpr int W2_V_BYTES = 3 * W2_PLB;                      // 49,152 per chunk; two buffers
constexpr int W2_NRAW = 3;                                  // raw ring slots: a chunk's DMA pieces get two steps to land
constexpr int W2_LDS_BYTES = 2 * W2_V_BYTES + W2_NRAW * W2_RAW_BYTES;   // 138,240
// F16: two V planes per buffer (32 KB) + a DEDICATED 48 KB exchange area for the epilogue's row transform: 64 + 39 + 48 KB
constexpr int W2_XCH_BYTES = 3 * W2_PLB;
constexpr int w2_v_bytes(bool f16) { return f16 ? 2 * W2_PLB : W2_V_BYTES; }
constexpr int w2_lds_bytes(bool f16) { return f16 ? 2 * 2 * W2_PLB + W2_NRAW * W2_RAW_BYTES + W2_XCH_BYTES : W2_LDS_BYTES; }   // 154,624 / 138,240
constexpr int W2_THREADS = 768;                             // 4 producer waves + 8 consumer waves
constexpr int W2_PPW = (W2_RAW_PIECES + 3) / 4;             // DMA pieces per wave of group B

// Exact unsigned division by a launch-time constant (the round-up method: q = (t + ((n - t) >> 1)) >> sh with t = mulhi(m, n); a shift for
// powers of two).  A hardware-free 32-bit division costs ~25 vector instructions on this ISA; the producers' per-task address set-up held ten
// of them -- ~1000 of the ~8000 cycles of a two-chunk task on the kernel's critical waves (fine-grained stamps in
// profiles/r04/bench_conv_wino2_f16x3.txt).
struct W2Div { unsigned m; int sh; int pow2; unsigned d; };
static inline W2Div w2_div_make(unsigned d) {
    W2Div r{0u, 0, 0, d};
    if ((d & (d - 1)) == 0) { r.pow2 = 1; while ((1u << r.sh) < d) ++r.sh; return r; }
    int l = 0;
    while ((1ull << l) < d) ++l;                             // ceil(log2 d)
    r.m = (unsigned)((((1ull << l) - d) << 32) / d + 1);
    r.sh = l - 1;
    return r;
}
__device__ __forceinline__ int w2_div(int n, const W2Div& d) {
    const unsigned un = (unsigned)n;
#ifdef SIR_W2_OLDADDR                                        // A/B build only (devtools/gpu_r4ac.sh): the ISA's division sequence
    return (int)(un / d.d);
#endif
    if (d.pow2) return (int)(un >> d.sh);
    const unsigned t = __umulhi(d.m, un);
    return (int)((t + ((un - t) >> 1)) >> d.sh);
}

struct Wino2Geo {
    int H, W;            // input = output map (pixels)
    int TW;              // tile columns per image = ceil(W / 2)
    int NG;              // tile columns of the batch = B * TW
    int RBN;             // 8-row tile blocks per image = (H / 2) / 8
    int NS;              // spatial tasks = RBN * ceil(NG / 4)
    int Hp, Wp;          // pooled map (OUT_MODE 0 / 1)
    int B;
    W2Div dTW, d2TW, dRBN;   // divisions by TW, 2 TW, RBN (all operands are non-negative)
};
// false: shape outside what the kernel covers (whole 8-tile-row blocks, 32-bit element offsets) -- the caller keeps the
// first-generation / direct kernel for it
static inline bool wino2_geo(int B, int H, int W, int cmax, Wino2Geo* g) {
    g->B = B; g->H = H; g->W = W; g->TW = (W + 1) / 2; g->NG = B * g->TW; g->RBN = H / 16;
    g->NS = g->RBN * ((g->NG + 3) / 4); g->Hp = H / 2; g->Wp = W / 2;
    g->dTW = w2_div_make((unsigned)g->TW); g->d2TW = w2_div_make(2u * (unsigned)g->TW); g->dRBN = w2_div_make((unsigned)(g->RBN > 0 ? g->RBN : 1));
    return H % 16 == 0 && W >= 1 && B >= 1 && (size_t)B * H * W * cmax < ((size_t)1 << 31) && (size_t)g->NG * 2 < ((size_t)1 << 30);
}
// statistics blocks of OUT_MODE 2: one per (workgroup, transform-row wave); `max_wg` as passed to launch_conv_wino2
static inline size_t wino2_stat_blocks(int B, int H, int W, int max_wg) {
    const size_t ns = (size_t)(H / 16) * (((size_t)B * ((W + 1) / 2) + 3) / 4);
    return (ns < (size_t)max_wg ? ns : (size_t)max_wg) * 4;
}

// OUT_MODE 0: pooled NHWC (BN + ReLU + max), 1: pooled in the GRU layout [b][tx][co * Hp + ty] (+ its f16x2 planes through
// `stats`), 2: raw NHWC + channel statistics (float2 {sum, sum of squares} at stats[(workgroup * 4 + row wave) * COUT + co]), 3: raw NHWC
// ---- LDS access behind the compiler's back (producer waves) ---------------------------------------------------------------
// hipcc orders every LDS read / write it can see behind ALL outstanding LDS-DMA of the wave (s_waitcnt vmcnt(0): it cannot prove
// that the DMA target and the access do not alias) and w2_barrier() drains vmcnt as well.  The producers keep two chunks of
// DMA pieces in flight, so their raw-patch reads and V writes are inline asm (invisible to the waitcnt pass), the waits are counted
// by hand and the barrier is a bare s_barrier behind s_waitcnt lgkmcnt(0).
typedef float w2_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned w2_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void w2_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// four 16-byte reads at base + 64 j, no wait (the caller waits once for all rows)
__device__ __forceinline__ void w2_read_row(unsigned base, w2_f32x4& r0, w2_f32x4& r1, w2_f32x4& r2, w2_f32x4& r3) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:128\n\tds_read_b128 %3, %4 offset:192"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(base) : "memory");
}
template <int OFF>
__device__ __forceinline__ void w2_write64(unsigned base, uint2 v) {
    const w2_u32x2 d = {v.x, v.y};
    asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(base), "v"(d), "n"(OFF) : "memory");
}

// DBG (devtools/kernel_ab/bench_conv.hip only): s_memtime stamps of the first producer and the first consumer wave of workgroup 0
__device__ long long w2_dbg_stamps[2][32];
__device__ long long w2_dbg_fine[8][8];              // producer wave 0 of workgroup 0, steps 8..15: loop top, DMA issued, raw patches read, V written, DMA wait over, barrier passed
// Fourth structure: 12 waves per workgroup -- 4 producers (raw patches by LDS-DMA three chunks deep, B^T d B, bf16x3, V[step & 1])
// and 8 consumers (wave (i, n): transform row i, channel slice n: 4 accumulators, weights of its 4 frequencies one chunk ahead),
// one barrier per chunk; a task's row transform goes through the V buffer its last chunk just left (two extra barriers per task).
// F16 (round 4): the contraction on the fp16 matrix cores with the two-way split (f16_split.h) instead of bf16x6 -- the producers
// write TWO planes (Vh, Vl' = residual * 2^11) instead of three (a third fewer split instructions and LDS writes), the consumers
// issue THREE products per frequency and 16-deep step instead of six (Vl' Uh, Vh Ul', Vh (Uh 2^11): one accumulator, 2^11 too large,
// scaled back in the epilogue; weights from prep_conv_w_wino_f16x3: same layout, TWO fp16 planes (Uh, Ul') -- Uh 2^11 is formed in
// registers, so the U stream that bounds the kernel is a third shorter as well).  The part runs at its power
// cap: the matrix products ARE the energy.  Needs inputs inside fp16's range (activations: yes; gradients only under the loss
// scale of the backward).  With 32 KB V buffers the LDS has room for a DEDICATED exchange area of the epilogue's row transform
// (the bf16x6 form borrows the V buffer the task's last chunk just left and needs two extra barriers, A and B, around that, with
// the producers parked at them): a consumer writes its pieces right behind its last MFMAs, the step's ONE barrier publishes them,
// and the producers are transforming the next chunk meanwhile -- the epilogue was 26 of the f16x3 kernel's 109 us (knock-outs,
// profiles/r04/bench_conv_wino2_f16x3.txt).
template <int CIN, int COUT, int OUT_MODE, int DBG = 0, int PRIO = 3, bool F16 = false>
__global__ __launch_bounds__(W2_THREADS, 3) void conv3x3_wino2_bf16x6_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ wpb, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ out, Wino2Geo geo, float2* __restrict__ stats, const float* __restrict__ zeros) {
    constexpr int NCH = CIN / 16, G = NCH * 16, NCHO = COUT >= 64 ? COUT / 64 : 1;
    static_assert(CIN % 16 == 0 && (COUT % 64 == 0 || COUT == 32), "16-channel chunks; 64-channel tasks (32: the n = 1 consumer waves only keep the barriers)");
    static_assert(!F16 || NCH >= 2, "F16: a task's exchange pieces are read behind its closing barrier; the next task's are written >= one barrier later");
    extern __shared__ __attribute__((aligned(1024))) unsigned char w2s[];
    constexpr int VB = w2_v_bytes(F16);                                 // bytes per V buffer
    unsigned char* const vbuf = w2s;                                    // [2][3 (F16: 2) planes][16 f][1 KB]
    unsigned char* const rawbuf = w2s + 2 * VB;                         // [3][13 KB]
    unsigned char* const xchbuf = rawbuf + W2_NRAW * W2_RAW_BYTES;      // F16 only: [8 waves][3 pieces][2 KB]

    // the wave index as a SCALAR (readfirstlane): everything derived from it -- roles, rows, channel slices, weight addresses -- then
    // lives in SGPRs; derived from threadIdx alone hipcc keeps it all in vector registers (+40 VGPRs in the consumer loop: spills)
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wv < 4;
    const int H = geo.H, W = geo.W, TW = geo.TW, NG = geo.NG, RBN = geo.RBN;
    const int ntask_s = (geo.NS - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // spatial tasks of this workgroup
    const int ntask = ntask_s * NCHO;
    if (ntask <= 0) return;
    const int nsteps = ntask * NCH;                                     // chunks of this workgroup; step s: producers chunk s, consumers chunk s - 1

    auto task_geo = [&](int lt, int& g0, int& ty0, int& ch, int& s) {
        s = (int)blockIdx.x + (lt / NCHO) * (int)gridDim.x;
        ch = lt % NCHO;
        const int cb = w2_div(s, geo.dRBN), rb = s - cb * RBN;
        g0 = 4 * cb; ty0 = 8 * rb;
    };
    // ---- raw-patch DMA: WHO issues it ------------------------------------------------------------------------------------------
    // The four producer waves (piece k = wave + 4 i, counted vmcnt waits in their loop) -- except in the F16 form of a 32-channel layer (the
    // conv2 data gradient), where the four consumer waves of the missing second channel slice have nothing to do but keep the barriers: they take
    // the DMA over, on the producers' own schedule (chunk s + 2 at the top of step s, landed by the end of step s + 1).  Stamps inside a producer's
    // step put the issue of its 3-4 LDS-DMA pieces at ~620 of ~3500 cycles (155 per piece) on the kernel's critical waves.  (Handing the pieces to
    // the WORKING consumers of the 64-channel forms was measured and lost 10 %: behind their MFMAs a chunk has one step instead of two to land,
    // and the in-order return puts it in front of their weight fragments -- profiles/r04/bench_conv_wino2_f16x3.txt.)
#ifdef SIR_W2_PRODDMA                                        // A/B build only (devtools/gpu_r4af.sh)
    constexpr bool IDLE_DMA = false;
#else
    constexpr bool IDLE_DMA = F16 && COUT < 64;
#endif
    constexpr int D_STRIDE = 4, D_PPW = (W2_RAW_PIECES + D_STRIDE - 1) / D_STRIDE;
    const int d_stride = D_STRIDE;
    const int d_rank = producer ? wv : ((wv - 4) >> 1);               // (the idle waves are 5, 7, 9, 11)
    // (the task-independent part of a piece's address -- raw row, column, channel group, "hole" positions of the padded image -- is computed
    // once; per task a piece then costs a clamp, one magic-number division and the bounds test instead of ~60 instructions)
    int pre_lr[D_PPW], pre_lc[D_PPW], pre_part[D_PPW];
#pragma unroll
    for (int ii = 0; ii < D_PPW; ++ii) {
        const int slot = 64 * (d_rank + d_stride * ii) + lane, pos = slot >> 2, sp = slot & 3;
        const int lr = pos / W2_RS;
        const int lc = pos - lr * W2_RS - ((lr >> 1) & 1);
        const bool hole = lr > 17 || lc < 0 || lc > 9;
        pre_lr[ii] = hole ? (1 << 20) : lr;                           // a hole fails the row test of every task
        pre_lc[ii] = lc;
        pre_part[ii] = (sp ^ (((lr >> 2) & 1) << 1)) * 4;
    }
    auto raw_offsets = [&](int g0, int ty0, unsigned (&off)[D_PPW]) {
#ifdef SIR_W2_OLDADDR                                        // A/B build only (devtools/gpu_r4ac.sh): everything recomputed per task, the ISA's division
#pragma unroll
        for (int ii = 0; ii < D_PPW; ++ii) {
            const int slot = 64 * (d_rank + d_stride * ii) + lane, pos = slot >> 2, sp = slot & 3;
            int lr = pos / W2_RS;
            int lc = pos - lr * W2_RS - ((lr >> 1) & 1);
            const bool hole = lr > 17 || lc < 0 || lc > 9;
            const int part = sp ^ (((lr >> 2) & 1) << 1);
            const int gy = 2 * ty0 - 1 + lr;
            const int P = 2 * g0 - 1 + lc;
            const int Pc = min(max(P, 0), 2 * NG - 1);
            const int bb = Pc / (2 * TW), px = Pc - bb * 2 * TW;
            const bool ok = !hole && gy >= 0 && gy < H && P >= 0 && P < 2 * NG && px < W;
            off[ii] = ok ? (unsigned)(((bb * H + gy) * W + px) * CIN + part * 4) : ~0u;
        }
        return;
#endif
#pragma unroll
        for (int ii = 0; ii < D_PPW; ++ii) {
            const int gy = 2 * ty0 - 1 + pre_lr[ii];
            const int P = 2 * g0 - 1 + pre_lc[ii];
            const int Pc = min(max(P, 0), 2 * NG - 1);
            const int bb = w2_div(Pc, geo.d2TW), px = Pc - bb * 2 * TW;
            const bool ok = gy >= 0 && gy < H && P >= 0 && P < 2 * NG && px < W;
            off[ii] = ok ? (unsigned)(((bb * H + gy) * W + px) * CIN + pre_part[ii]) : ~0u;
        }
    };
        auto raw_issue = [&](const unsigned (&off)[D_PPW], int c, int slot_buf) {
#pragma unroll
        for (int ii = 0; ii < D_PPW; ++ii) {
            const int k = d_rank + d_stride * ii;
            if (k < W2_RAW_PIECES) {
                const float* src = off[ii] == ~0u ? zeros : x + off[ii];
                __builtin_amdgcn_global_load_lds((sir_gptr_t)(src + c * 16), (sir_lptr_t)(rawbuf + slot_buf * W2_RAW_BYTES + k * 1024), 16, 0, 0);
            }
        }
    };
    // chunk q (global index over this workgroup's tasks) -> issue its pieces into ring slot q % 3
    int ig0, ity0, ich, is_;
    unsigned roff[D_PPW];
    int roff_task = -1;
    auto issue_chunk = [&](int q) {
        if (q >= nsteps) return;
        const int lt = q / NCH, c = q - lt * NCH;
        if (lt != roff_task) { task_geo(lt, ig0, ity0, ich, is_); raw_offsets(ig0, ity0, roff); roff_task = lt; }
        raw_issue(roff, c, q % W2_NRAW);
    };
    int nst = 0;
    auto stamp = [&]() {
        if (DBG && blockIdx.x == 0 && lane == 0 && (wv == 0 || wv == 4) && nst < 32) w2_dbg_stamps[producer ? 0 : 1][nst] = __builtin_amdgcn_s_memtime();
        ++nst;
    };

    if (producer) {
        // ================= producers ==============================================================================================
        // (their VALU stream competes with two MFMA-issuing consumer waves for the SIMD's issue port: priority to the producer)
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
#ifndef SIR_W2_CLAMP
        // MODE.FP16_OVFL = 1 for the producer waves: a conversion to fp16 that overflows gives +-65504 instead of infinity, which is what
        // the split's two v_med3 clamps per pair were for (32 of the ~170 vector instructions per row pair and step)
        if (F16) __builtin_amdgcn_s_setreg((1 /* HW_REG_MODE */) | (23 << 6) | (0 << 11), 1);
#endif
        const int wg = wv;
        // wave wg = (row pair tR of the transform, 8-channel half tH); lane = (tile tm, 4-channel group tP1)
        const int tR = wg >> 1, tH = wg & 1, tm = lane >> 1, tP1 = lane & 1;
        const int tty = tm & 7, ttx = tm >> 3, tpart = 2 * tH + tP1;
        // DMA source of this lane's pieces for a task: element offset into x, or ~0u = the zero page (rows above / below the image,
        // the pad column of an odd-width map, tile columns past the batch: the transform then needs no selects for them)
        // loop-invariant LDS offsets of this thread: patch pixel (row rr, j = 0) inside a raw slot, its V destination inside a V buffer
        unsigned ra_rel[3];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const int lr = 2 * tty + tR + rr;
            ra_rel[rr] = (unsigned)((lr * W2_RS + 2 * ttx + ((lr >> 1) & 1)) * 64 + ((tpart ^ (((lr >> 2) & 1) << 1)) * 16));
        }
        const unsigned raw_a = (unsigned)(uintptr_t)rawbuf, v_a = (unsigned)(uintptr_t)vbuf + (8 * tR) * 1024 + tH * 512 + tm * 16 + tP1 * 8;
        if (!IDLE_DMA) {
            issue_chunk(0);
            issue_chunk(1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        w2_barrier();                                                   // raw chunks 0 and 1 have landed
        bool edge = false, z0 = false, z3 = false;                      // the task holds an image boundary column; this thread's tile sits at one
        auto fine = [&](int s_, int k_) {
            if (DBG && blockIdx.x == 0 && lane == 0 && wv == 0 && s_ >= 8 && s_ < 16) w2_dbg_fine[s_ - 8][k_] = __builtin_amdgcn_s_memtime();
        };
#pragma unroll 1
        for (int s = 0; s <= nsteps; ++s) {
            stamp();
            fine(s, 0);
            if (s < nsteps) {
                if (!IDLE_DMA && !(DBG & 2)) issue_chunk(s + 2);        // into the slot chunk s - 1 left (its reads ended before the last barrier)
                fine(s, 1);
                if (s % NCH == 0) {                                     // first chunk of a task: where are its image boundaries
                    int cg0, cty0, cch, cs;
                    task_geo(s / NCH, cg0, cty0, cch, cs);
                    const int t0 = cg0 - w2_div(cg0, geo.dTW) * TW;     // tile column of the task's first column inside its image
                    edge = t0 == 0 || t0 + 3 >= TW - 1;
                    const int txx = (t0 + ttx) - w2_div(t0 + ttx, geo.dTW) * TW;
                    z0 = txx == 0; z3 = txx == TW - 1;
                }
                const unsigned rb = raw_a + (s % W2_NRAW) * W2_RAW_BYTES, vd = v_a + (s & 1) * VB;
                if (!(DBG & 4)) {
                w2_f32x4 q[3][4];
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) w2_read_row(rb + ra_rel[rr], q[rr][0], q[rr][1], q[rr][2], q[rr][3]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                fine(s, 2);
                if (edge) {                                             // (one task in ~12: the halo column belongs to the neighbouring image)
                    // a real branch (the asm keeps hipcc from turning it into 24 selects on the common path).  SELECT zero, do not
                    // multiply by it: the halo column was loaded from the neighbouring clip, and 0 * Inf / NaN of a corrupt
                    // neighbour would leak into this clip's boundary columns (eval-mode samples are independent in the reference)
                    asm volatile("" ::: "memory");
                    const w2_f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr) { q[rr][0] = z0 ? zero4 : q[rr][0]; q[rr][3] = z3 ? zero4 : q[rr][3]; }
                }
                // patch rows tR .. tR + 2 (row pair 0: i = 0: d0 - d2, i = 1: d1 + d2; pair 1 (rows 1,2,3): i = 2: d2 - d1, i = 3: d1 - d3).
                // tR is wave-uniform: two straight-line copies under a branch (as one body hipcc computed both and selected: 32 v_cndmask)
                auto rows = [&](auto trc) {
                    constexpr int TR = decltype(trc)::value;
#pragma unroll
                    for (int il = 0; il < 2; ++il) {
                        w2_f32x4 R[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (TR == 0) R[j] = il == 0 ? q[0][j] - q[2][j] : q[1][j] + q[2][j];
                            else         R[j] = il == 0 ? q[1][j] - q[0][j] : q[0][j] - q[2][j];
                        }
                        const w2_f32x4 V[4] = {R[0] - R[2], R[1] + R[2], R[2] - R[1], R[1] - R[3]};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const unsigned d = vd + (4 * il + j) * 1024;
                            if constexpr (F16) {
                                uint2 sh, sl;
#ifdef SIR_W2_CLAMP                                          // A/B build only: the split's own clamps (v_med3 per value)
                                split2h_pair(V[j].x, V[j].y, sh.x, sl.x);
                                split2h_pair(V[j].z, V[j].w, sh.y, sl.y);
#else
                                split2h_pair_ovfl(V[j].x, V[j].y, sh.x, sl.x);
                                split2h_pair_ovfl(V[j].z, V[j].w, sh.y, sl.y);
#endif
                                w2_write64<0>(d, sh);
                                w2_write64<W2_PLB>(d, sl);
                            } else {
                                uint2 sh, sm, sl;
                                split3_pair(V[j].x, V[j].y, sh.x, sm.x, sl.x);
                                split3_pair(V[j].z, V[j].w, sh.y, sm.y, sl.y);
                                w2_write64<0>(d, sh);
                                w2_write64<W2_PLB>(d, sm);
                                w2_write64<2 * W2_PLB>(d, sl);
                            }
                        }
                    }
                };
                if (tR == 0) rows(std::integral_constant<int, 0>{});
                else rows(std::integral_constant<int, 1>{});
                if (DBG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                fine(s, 3);
                }
                // chunk s + 1 (issued a step ago) must have landed before the barrier; the pieces of chunk s + 2, just issued, may still fly
                // (a wave issues 3 or 4 pieces per chunk: allowing its 3 newest operations to be outstanding is safe for both)
                if (!IDLE_DMA) {
                    if (s + 2 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W2_RAW_PIECES / 4) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            stamp();
            fine(s, 4);
            if (!F16 && s >= 1 && (s - 1) % NCH == NCH - 1) { w2_barrier(); w2_barrier(); }     // bf16x6: the consumers' row-transform exchange (barriers A, B)
            w2_barrier();
            fine(s, 5);
        }
        return;
    }

    // ================= consumers ==================================================================================================
    // wave (n = cw & 1, i = cw >> 1): channel slice 32 n.., transform row i (frequencies 4 i .. 4 i + 3, 4 accumulators)
    const int cw = wv - 4, mn = cw & 1, mi = cw >> 1, m = lane & 31, h = lane >> 5;
    if (COUT < 64 && mn == 1) {                                         // a 32-channel layer has no second slice: join the barriers --
        if (IDLE_DMA) {                                                 // -- and (F16) feed the raw-patch ring in the producers' stead
            if (!(DBG & 2)) { issue_chunk(0); issue_chunk(1); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            w2_barrier();                                               // raw chunks 0 and 1 have landed
#pragma unroll 1
            for (int s = 0; s <= nsteps; ++s) {
                if (s < nsteps) {
                    if (!(DBG & 2)) issue_chunk(s + 2);                 // into the slot chunk s - 1 left (its reads ended before the last barrier)
                    // chunk s + 1 (issued a step ago) must have landed before the barrier; the 3 or 4 pieces just issued may still fly
                    if (s + 2 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W2_RAW_PIECES / 4) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                w2_barrier();
            }
            return;
        }
        w2_barrier();
        w2_barrier();
        for (int lt = 0; lt < ntask; ++lt)
            for (int c = 0; c < NCH + (F16 ? 0 : 2); ++c) w2_barrier();
        return;
    }
    // U fragment address = uniform part (plane, frequency, channel block, slice: scalar registers) + this lane's 32-bit byte offset
    const unsigned wlane = (unsigned)((m * 2 + h) * 16);
    const unsigned char* const wbase = reinterpret_cast<const unsigned char*>(wpb) + (size_t)mn * 1024;
    constexpr int NPW = F16 ? 2 : 3;                                    // weight planes
    uint4 wq[4][NPW];                                                   // U fragments of the wave's four frequencies, one chunk ahead
    auto load_w = [&](int gidx, int chh, uint4 (&q)[NPW]) {
#pragma unroll
        for (int p = 0; p < NPW; ++p)
            q[p] = *reinterpret_cast<const uint4*>(wbase + ((size_t)(p * G + gidx) * (COUT * 2) + (size_t)chh * 128) * 16 + wlane);
    };
    int g0, ty0, ch, s_idx;
    task_geo(0, g0, ty0, ch, s_idx);
#pragma unroll
    for (int j = 0; j < 4; ++j) load_w(4 * mi + j, ch, wq[j]);
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    w2_barrier();                                                    // (pairs with the producers' "raw chunks 0 and 1 have landed")
    float run_s[NCHO], run_q[NCHO];                                    // OUT_MODE 2: channel statistics of this wave's outputs, per channel block, over ALL its tasks
#pragma unroll
    for (int k = 0; k < NCHO; ++k) { run_s[k] = 0.0f; run_q[k] = 0.0f; }

    stamp();
    stamp();
    w2_barrier();                                                       // step 0: the producers transform the first chunk
    // step s = lt * NCH + c + 1 multiplies chunk sc = s - 1.  Nested loops (tasks, chunks) rather than one step loop with a runtime
    // "task ends here" flag: with the epilogue inside the step loop hipcc kept its working set alive through the matrix steps and
    // spilled the weight ring into them (376-480 B of scratch per lane for 64 and 128 input channels)
#pragma unroll 1
    for (int lt = 0; lt < ntask; ++lt) {
        int g0n = g0, ty0n = ty0, chn = ch, sn = s_idx;
        if (lt + 1 < ntask) task_geo(lt + 1, g0n, ty0n, chn, sn);
#pragma unroll 1
        for (int c = 0; c < NCH; ++c) {
            stamp();
            const int sc = lt * NCH + c;
            const bool task_end = c == NCH - 1;
            const unsigned char* abase = vbuf + (sc & 1) * VB + h * 512 + m * 16 + (4 * mi) * 1024;
            const int gnxt = (task_end ? 0 : c + 1) * 16 + 4 * mi;
            // two frequencies at a time (two independent accumulator chains in flight), each pair followed AT ONCE by the loads of ITS
            // weights for the next chunk, fenced: left to itself hipcc sinks all twelve weight loads to the end of the step and the next
            // step starts with an L2 round trip
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                constexpr int NPA = F16 ? 2 : 3;
                bf16x8 a[2][NPA], bq[2][3];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
                    for (int p = 0; p < NPA; ++p)
                        a[jj][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(abase + p * W2_PLB + (2 * jp + jj) * 1024));
                    if constexpr (F16) {                               // planes (Uh, Ul') -> fragments (Uh, Uh 2^11, Ul')
                        const f16x8 uh = __builtin_bit_cast(f16x8, wq[2 * jp + jj][0]);
                        bq[jj][0] = __builtin_bit_cast(bf16x8, uh);
                        bq[jj][1] = __builtin_bit_cast(bf16x8, uh * (_Float16)2048.0f);
                        bq[jj][2] = __builtin_bit_cast(bf16x8, wq[2 * jp + jj][1]);
                    } else {
#pragma unroll
                        for (int p = 0; p < 3; ++p) bq[jj][p] = __builtin_bit_cast(bf16x8, wq[2 * jp + jj][p < NPW ? p : 0]);
                    }
                }
                if (DBG & 8) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        acc[2 * jp + jj][0] += __builtin_bit_cast(float4, a[jj][0]).x + __builtin_bit_cast(float4, a[jj][1]).y + __builtin_bit_cast(float4, a[jj][NPA - 1]).z +
                                               __builtin_bit_cast(float4, bq[jj][0]).x + __builtin_bit_cast(float4, bq[jj][1]).y + __builtin_bit_cast(float4, bq[jj][2]).z;
                } else if constexpr (F16) {
                    // a: 0 = Vh, 1 = Vl'; bq: 0 = Uh, 1 = Uh 2^11, 2 = Ul' -- the two cross terms first, then the main one
                    constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 2, 1};
#pragma unroll
                    for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj)
                            acc[2 * jp + jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[jj][HA[t3]]),
                                                                                      __builtin_bit_cast(f16x8, bq[jj][HB[t3]]), acc[2 * jp + jj], 0, 0, 0);
                } else {
                    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // small terms first
#pragma unroll
                    for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj)
                            acc[2 * jp + jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[jj][PA[t6]], bq[jj][PB[t6]], acc[2 * jp + jj], 0, 0, 0);
                }
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) if (!(DBG & 64)) load_w(gnxt + 2 * jp + jj, task_end ? chn : ch, wq[2 * jp + jj]);
                __builtin_amdgcn_sched_barrier(0);
            }
            stamp();
            if (!task_end) w2_barrier();
        }
        const int sc = lt * NCH + NCH - 1;                              // the task's last chunk: its V buffer becomes the exchange area
        if (DBG & 16) { if (!F16) { w2_barrier(); w2_barrier(); } w2_barrier(); if (acc[0][0] == 1234.5f && acc[1][1] + acc[2][2] + acc[3][3] == 4.0f) out[0] = 1.0f; }
        else {
            {
                if (!F16) w2_barrier();                              // A (bf16x6): every consumer has read its last fragments of V[sc & 1]
                // column inverse transform (U_{i3} is stored negated); finisher k = row index of slice mn completes accumulator registers
                // 4 k .. 4 k + 3 (tile column k, tile rows 4 h + e).  Piece (source i -> finisher k): W_i[b][4 k + e] as two float4 (b)
                float4* const xch = reinterpret_cast<float4*>(F16 ? xchbuf : vbuf + (sc & 1) * VB);
                float own[2][4];
#pragma unroll
                for (int b = 0; b < 2; ++b) {                          // one b at a time: 16 live registers instead of 32 beside the accumulators
                    const f32x16 Wc = b == 0 ? acc[0] + acc[1] + acc[2] : acc[1] - acc[2] + acc[3];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (k == mi) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) own[b][e] = Wc[4 * k + e];
                        } else {
                            float4* dst = xch + (size_t)((mn * 4 + mi) * 3 + ((k - mi - 1) & 3)) * 128 + lane;
                            dst[b * 64] = make_float4(Wc[4 * k], Wc[4 * k + 1], Wc[4 * k + 2], Wc[4 * k + 3]);
                        }
                    }
                }
                w2_barrier();                                        // B: the pieces are in LDS (F16: this IS the step's closing barrier -- the
                                                                     // producers are already on the next chunk, nobody parks)
                // Y[0][b] = (W0 + W1) + W2, Y[1][b] = (W1 - W2) - W3: always in THIS order, whichever row the finishing wave holds itself --
                // the tile column a clip lands on depends on its position in the batch, and a clip's logits must not (bit for bit)
                float Y[2][2][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int kk = (mi - i - 1) & 3;                   // piece index of source row i for this finisher (3: i is this wave itself)
                    const float4* src = xch + (size_t)((mn * 4 + i) * 3 + (kk < 3 ? kk : 0)) * 128 + lane;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const float4 v = src[b * 64];
                        const float w[4] = {i == mi ? own[b][0] : v.x, i == mi ? own[b][1] : v.y, i == mi ? own[b][2] : v.z, i == mi ? own[b][3] : v.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (i == 0) Y[0][b][e] = w[e];
                            else if (i <= 2) Y[0][b][e] += w[e];
                            if (i == 1) Y[1][b][e] = w[e];
                            else if (i >= 2) Y[1][b][e] -= w[e];
                        }
                    }
                }
                if (!F16) w2_barrier();                              // C (bf16x6: this step's closing barrier, early): the pieces are in registers, the
                                                                     // producers may overwrite this V buffer -- they transform the next chunk while the outputs are finished here
                const int co = ch * 64 + mn * 32 + m;
                float ssum = 0.0f, ssq = 0.0f;
                float sc_ = 1.0f, sh_ = 0.0f;
                constexpr float DESC = F16 ? H3_LO_INV : 1.0f;         // the f16x3 accumulators are 2^11 too large (exact power of two)
                if (OUT_MODE <= 1) { sc_ = scale[co] * DESC; sh_ = shift[co]; }
                {
                    const int gc = g0 + mi;
                    const int img = w2_div(gc, geo.dTW), tx = gc - img * TW;
                    const bool tvalid = gc < NG;
                    if (OUT_MODE <= 1) {
                        float pooled[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = 0.0f;
#pragma unroll
                            for (int a = 0; a < 2; ++a)
#pragma unroll
                                for (int b = 0; b < 2; ++b) v = fmaxf(v, fmaf(Y[a][b][e], sc_, sh_));
                            pooled[e] = v;
                        }
                        if (tvalid && tx < geo.Wp) {
                            if (OUT_MODE == 0) {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    out[(((size_t)img * geo.Hp + ty0 + 4 * h + e) * geo.Wp + tx) * COUT + co] = pooled[e];
                            } else {
                                const size_t oidx = ((size_t)img * geo.Wp + tx) * (COUT * geo.Hp) + (size_t)co * geo.Hp + ty0 + 4 * h;
                                const float4 v4 = make_float4(pooled[0], pooled[1], pooled[2], pooled[3]);
                                *reinterpret_cast<float4*>(out + oidx) = v4;
                                if (stats) {           // f16x2 planes [2][B * Wp][COUT * Hp] (f16_split.h) of the following GEMM's A operand
                                    unsigned short* planes = reinterpret_cast<unsigned short*>(stats);
                                    const size_t plane = (size_t)geo.B * geo.Wp * (COUT * geo.Hp);
                                    uint2 hh, ll;
                                    split2h_quad(v4, hh, ll);
                                    *reinterpret_cast<uint2*>(planes + oidx) = hh;
                                    *reinterpret_cast<uint2*>(planes + plane + oidx) = ll;
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
                                    const int gy = 2 * (ty0 + 4 * h + e) + a, gx = 2 * tx + b;
                                    if (tvalid && gx < W) {
                                        const float v = Y[a][b][e] * DESC;
                                        out[(((size_t)img * H + gy) * W + gx) * COUT + co] = v;
                                        ssum += v;
                                        ssq = fmaf(v, v, ssq);
                                    }
                                }
                    }
                }
                if (OUT_MODE == 2) {                                     // (kept in registers across the tasks: one partial per workgroup and row wave)
#pragma unroll
                    for (int k = 0; k < NCHO; ++k)
                        if (k == ch) { run_s[k] += ssum; run_q[k] += ssq; }
                }
            }
        }
        g0 = g0n; ty0 = ty0n; ch = chn; s_idx = sn;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    }
    if (OUT_MODE == 2 && stats) {
        // statistics block (workgroup, row wave): 4 gridDim.x blocks of COUT channels in all (wino2_stat_blocks), fixed summation order
#pragma unroll
        for (int k = 0; k < NCHO; ++k) {
            float a = run_s[k], q = run_q[k];
            a += __shfl_xor(a, 32);
            q += __shfl_xor(q, 32);
            if (h == 0) stats[((size_t)blockIdx.x * 4 + mi) * COUT + k * 64 + mn * 32 + m] = make_float2(a, q);
        }
    }
}

// `attr_done`: the caller's per-device latch of the dynamic-LDS opt-in of THIS instantiation (sir_handle::attr_wino2[...])
template <int CIN, int COUT, int OUT_MODE, int DBG = 0, int PRIO = 3, bool F16 = false>
static inline hipError_t launch_conv_wino2(hipStream_t st, bool* attr_done, const float* x, const unsigned short* wpb, const float* scale,
                                           const float* shift, float* out, int B, int H, int W, float2* stats, const float* zeros, int max_wg = 256) {
    Wino2Geo g;
    if (!wino2_geo(B, H, W, CIN > COUT ? CIN : COUT, &g)) return hipErrorInvalidValue;
    if (!*attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_wino2_bf16x6_kernel<CIN, COUT, OUT_MODE, DBG, PRIO, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, w2_lds_bytes(F16));
        if (e != hipSuccess) return e;
        *attr_done = true;
    }
    const int nwg = g.NS < max_wg ? g.NS : max_wg;
    hipLaunchKernelGGL((conv3x3_wino2_bf16x6_kernel<CIN, COUT, OUT_MODE, DBG, PRIO, F16>), dim3(nwg), dim3(W2_THREADS), w2_lds_bytes(F16), st, x, wpb, scale, shift, out, g, stats, zeros);
    return hipGetLastError();
}
