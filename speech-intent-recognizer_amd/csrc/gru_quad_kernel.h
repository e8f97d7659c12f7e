// Quad-workgroup GRU recurrence on the matrix cores (fp32 accuracy).  Round 4: the products run on the fp16 cores with the
// two-way split of f16_split.h ("f16x3": W = Wh + Wl' 2^-11, h = hh + hl' 2^-11; three products per fp32 product -- Wl' hh and
// Wh hl' into one accumulator, Wh hh into another, folded as acc0 + 2^-11 acc1 -- instead of bf16x6's six): 72 instead of 144
// MFMAs per wave and step, 192 instead of 288 resident weight registers, 2 instead of 3 h planes in LDS and in the granules.
// h is in (-1, 1) and W_hh O(0.1): well inside fp16's range.  The description below is of the structure, unchanged since round 2:
//
// gru_pair_kernel computes W_hh h with fp32 FMAs and is bound by VALU issue (~64 us of FMA issue per
// layer at batch 256 plus 25 dependent exchanges).  Here the recurrent product runs on MFMA:
//   * one CLUSTER of four workgroups owns 16 utterances of one direction for all S steps;
//   * workgroup q owns hidden units [64 q, 64 q + 64), i.e. 192 gate rows of W_hh; its 4 waves each own 16
//     units x 3 gates = three 16 x 256 row tiles, kept for the whole sequence IN REGISTERS as bf16x3 planes
//     (3 gates x 8 k-steps x 3 planes x 4 VGPRs = 288 VGPRs per lane: one wave per SIMD, 512-register budget);
//   * per step a wave issues 3 tiles x 8 k-steps x 6 plane products = 144 v_mfma_f32_16x16x32_bf16 with
//     A = W tile (rows = units) and B = h^T (columns = the 16 utterances), so that a lane ends up with the
//     r, z, n pre-activations of 4 consecutive units of ONE utterance in its three accumulators: the gate
//     arithmetic is lane-local;
//   * h lives in LDS as three bf16 planes [utterance][k] (double-buffered by step parity, rows padded to
//     528 B); every workgroup needs all 256 units, so after each step the four quarters exchange their
//     64 x 16 new values through global memory: one 8-byte granule per value = {16-bit tag = launch epoch + step + 1,
//     hi, lo' fp16, 16 spare bits}, written write-through (sc1, the code of an agent-scope relaxed atomic store) two granules per
//     16-byte store, and polled with 16-byte sc1 loads until every tag matches (recipe R2 of cdna_hip_programming.md
//     G16: the data is the flag -- each granule carries its own tag, so nothing depends on a pair landing together).
//     Granule buffers alternate by step parity; a buffer is zeroed when new to the handle, not per launch.
// Results differ from the fp32-FMA kernel only by the bf16x6 product rounding (~2^-24 relative per product).
#pragma once
#include "bf16x6_kernels.h"
#include "gru_frag_prep.h"

typedef unsigned int gq_u32x4 __attribute__((ext_vector_type(4)));   // native vector: usable as an inline-asm operand
constexpr int GQ_NU = 16;                 // utterances per cluster (the MFMA N dimension)
constexpr int GQ_UQ = 64;                 // hidden units per workgroup
constexpr int GQ_THREADS = 256;
constexpr int GQ_ROWB = 256 * 2 + 32;     // bytes per utterance row of one h plane: 34 sixteen-byte slots = 2 mod 16 puts the 16 lanes of every
                                          // ds_read_b128 group ({0-3, 12-15, 20-27}, ...) on 16 different slots; with + 16 (rounds 2-3) lanes
                                          // (n, kg) = (12, 0) and (11, 1) of one group shared a slot (PMC r04: conflict ratio 0.49)
constexpr int GQ_PLANEB = GQ_NU * GQ_ROWB;
constexpr int GQ_NPL = 2;                 // h planes: hi, scaled residual
constexpr int GQ_BUFB = GQ_NPL * GQ_PLANEB;    // one parity buffer: 17,408 B
constexpr size_t GQ_LDS_PLANES = 2 * (size_t)GQ_BUFB;
constexpr int GQ_TROW = 64 + 4;                              // floats per (utterance, gate) row of the layout-crossing image
constexpr size_t GQ_LDS_BYTES = GQ_LDS_PLANES + (size_t)GQ_NU * 3 * GQ_TROW * 4;      // (+ 13 KB, used by the ROLES form)
constexpr unsigned GQ_SPIN_LIMIT = 1u << 22;
constexpr int GQ_POLL_DELAY = 8;           // default (16 before the f16x3 step got shorter: profiles/r04/ab_gq_delay.txt), x 64 cycles between the granule stores and the first poll round (see the poll loop); passed in dbg bits 8-12
constexpr size_t GQ_XBUF_PER_CLUSTER = (size_t)2 * 4 * GQ_NU * GQ_UQ * 8;   // [parity][quarter][wave][store 0 | 1][lane][2] granules (producer-thread order)

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// One v_mfma_f32_16x16x32_f16 whose A operand (a resident W_hh fragment) is taken from an ACCUMULATION register.  The recurrence kernels
// keep 192 registers of weight fragments for the whole sequence; left to the register allocator they overflow the 256 architectural
// VGPRs and it parks part of the working set in AGPRs, copying it back and forth every step (64 v_accvgpr_read + 48 v_accvgpr_write
// per step in the forward kernel, 121 + 103 in the BPTT kernel: ~0.2-0.35 us of a 2.7-4 us step).  gfx90a+ matrix instructions read
// A / B from AGPRs directly, so the fragments are pinned there by the operand constraint and nothing is copied.  hipcc cannot see an
// MFMA inside inline asm: the wait states it would insert between the matrix results and their first vector use are added by hand
// (gq_mfma_fence: 16 idle cycles cover the 4-pass result latency; gq_mfma_enter: 4 in front of the block cover vector writes of its operands).
__device__ __forceinline__ void gq_mfma_aw(f32x4_t& acc, const f16x8& w, const f16x8& h) {
#ifdef SIR_GQ_BUILTIN_MFMA                                   // A/B build only (devtools/gpu_r4q.sh): compiler-managed registers
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, h, acc, 0, 0, 0);
#else
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(h));
#endif
}
__device__ __forceinline__ void gq_mfma_wa(f32x4_t& acc, const f16x8& h, const f16x8& w) {        // the fragment as the B operand
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(h), "a"(w));
}
// wait states in front of / behind a block of inline-asm MFMAs, TIED to the accumulators by in-out operands so that neither the
// accumulators' initialisation nor their first vector use can be scheduled to the wrong side of the idle cycles
__device__ __forceinline__ void gq_mfma_enter(f32x4_t (&a)[3], f32x4_t (&b)[3]) {
    asm volatile("s_nop 3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));
}
__device__ __forceinline__ void gq_mfma_fence(f32x4_t (&a)[3], f32x4_t (&b)[3]) {
    asm volatile("s_nop 15" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));
}
__device__ __forceinline__ void gq_mfma_enter(f32x4_t (&a)[4], f32x4_t (&b)[4]) {
    asm volatile("s_nop 3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}
__device__ __forceinline__ void gq_mfma_fence(f32x4_t (&a)[4], f32x4_t (&b)[4]) {
    asm volatile("s_nop 15" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}

// gate non-linearities from the hardware exp2 / rcp (1 ulp each) instead of the libm expf / tanhf / IEEE division:
// this kernel evaluates 12 of them per lane and step on ONE wave per SIMD, where the ~110-instruction libm
// forms cost ~1 us per step.  Absolute error ~1e-7 (sigmoid) / ~2e-7 (tanh), far inside the 2e-5 logit tolerance.
__device__ __forceinline__ float gq_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float gq_tanh(float x) {
    // 1 - 2 / (1 + e^{2x}); saturates cleanly: e^{2x} -> inf gives 1, -> 0 gives -1
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x));
}

// W_hh of one direction as the kernel's resident MFMA fragments (gru_frag_prep.h: [quarter][wave][gate][k-step][plane][lane] uint4);
// inference prepares this once per weights version, training once per step inside train_prep_kernel: the kernel prologue is then
// 48 coalesced 16-byte loads per lane instead of 48 row-strided fp32 loads and the split
static __global__ __launch_bounds__(256) void prep_whh_quad_kernel(const float* __restrict__ whh, uint4* __restrict__ frag) {
    prep_whh_quad_elem(whh, frag, blockIdx.x * 256 + threadIdx.x);
}

// xbuf   [clusters][2 parity][4 quarters][16 utterances][64 units] 8-byte granules; the 16-bit tag is {7-bit launch epoch of
//        this buffer, 9-bit step + 1}: a granule of an earlier launch never matches, so the buffer is zeroed only when it is
//        new or grows (sir_xbuf_epoch), not before every launch
// status set to 1 if a spin times out (results are then invalid; cannot happen while a cluster is co-resident)
// ROLES: two thread roles as in gru_bwd_quad_kernel.h -- the MFMA result layout (lane = utterance lane & 15, units 16 wave + 4 (lane >> 4) + j)
// only for the product; the gate arithmetic and everything that touches global memory (gi loads, y / gate / plane stores) in a GATE layout
// (utterance 4 wave + lane / 16, units 4 (lane & 15) + j): 256 contiguous bytes per 16 lanes instead of 64 separate 16-byte requests per
// instruction.  The pre-activations cross over through a 13 KB LDS image and one more barrier per step.
template <bool SAVE, bool ROLES = false>
__global__ __launch_bounds__(GQ_THREADS) void gru_quad_kernel(
    const float* __restrict__ gi, const float* __restrict__ whh0, const float* __restrict__ whh1,
    const float* __restrict__ bhh0, const float* __restrict__ bhh1, float* __restrict__ y, int B, int S,
    float* __restrict__ gates, unsigned long long* xbuf, unsigned int* status, int dbg, unsigned epoch, unsigned short* __restrict__ yplanes,
    const uint4* __restrict__ wfrag0, const uint4* __restrict__ wfrag1) {
    // wfrag0/1: prep_whh_quad_elem output for direction 0 / 1 (required)
    // yplanes (optional): f16x2 planes [2][B * S][512] (f16_split.h) of y, the A operand of the next layer's input projection
    // dbg (timing experiments only, results invalid): bit 0 = do not wait for the granules, bit 1 = skip the MFMAs,
    // bit 2 = skip publish + receive; fault injection for the status-word test: bit 3 = quarter 3 never publishes (its
    // peers time out), bit 4 = spin limit 4096 instead of 2^22 (so that the injected timeout takes milliseconds);
    // bits 8-12 = delay of the first poll round in units of 64 cycles (the launcher passes GQ_POLL_DELAY or SIR_GQ_DELAY; results stay valid)
    extern __shared__ __attribute__((aligned(16))) unsigned char qlds[];
    // blockIdx.x = L -> (quarter, cluster).  Workgroup L runs on XCD L % 8 (round-robin dispatch): with the clusters in eights the four
    // quarters of a cluster are L = x, x + 8, x + 16, x + 24 of a block of 32 -- one XCD, so that the per-step exchange stays inside
    // that XCD's L2 instead of crossing the fabric (correctness does not depend on the placement: the granule stores / loads are
    // agent-scope either way)
    const int nclusters = gridDim.x >> 2;
    const bool xcd_order = (nclusters & 7) == 0 && !(dbg & 32);          // (dbg bit 5: plain order, for the A/B)
    const int q = xcd_order ? (blockIdx.x >> 3) & 3 : blockIdx.x & 3;
    const int cluster = xcd_order ? (blockIdx.x & 7) + 8 * (blockIdx.x >> 5) : blockIdx.x >> 2;
    const int dir = cluster & 1, grp = cluster >> 1;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 15, kg = lane >> 4;                 // MFMA column (utterance) / k-group and D row group
    // gate role (== the MFMA result layout unless ROLES): utterance gutt of the cluster, units gul0 .. gul0 + 3 of the quarter
    const int gutt = ROLES ? 4 * wv + (lane >> 4) : n;
    const int gul0 = ROLES ? 4 * (lane & 15) : wv * 16 + kg * 4;
    const int b = grp * GQ_NU + gutt;
    const bool bvalid = b < B;
    const int u0 = q * GQ_UQ + gul0;                         // first of this thread's 4 hidden units
    float* const tl = reinterpret_cast<float*>(qlds + GQ_LDS_PLANES);      // ROLES: [16 utterances][3 gates][GQ_TROW] pre-activations
    const float* __restrict__ bhh = dir ? bhh1 : bhh0;
    unsigned long long* xc = xbuf + (size_t)cluster * (2 * 4 * GQ_NU * GQ_UQ);

    // ---- resident weights: A fragments of the three gate tiles, f16x2 planes ----------------------------
    // (loaded straight INTO accumulation registers, where gq_mfma_aw reads them: see its comment; the launcher insists on prepared fragments)
    f16x8 wf[3][8][GQ_NPL];
    {
        const uint4* wsrc = (dir ? wfrag1 : wfrag0) + (size_t)((q * 4 + wv) * 3 * 8 * GQ_NPL) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int p = 0; p < GQ_NPL; ++p)
#ifdef SIR_GQ_BUILTIN_MFMA
                    wf[g][s][p] = __builtin_bit_cast(f16x8, wsrc[((g * 8 + s) * GQ_NPL + p) * 64]);
#else
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(wf[g][s][p]) : "v"(wsrc + ((g * 8 + s) * GQ_NPL + p) * 64));
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float4 bh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bh[g] = *reinterpret_cast<const float4*>(bhh + g * 256 + u0);
    for (int i = tid; i < (int)(GQ_LDS_BYTES / 16); i += GQ_THREADS) reinterpret_cast<uint4*>(qlds)[i] = make_uint4(0, 0, 0, 0);
    float4 hprev = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    // gate pre-activations of the input side, fetched one step ahead
    float4 gin[3];
    auto load_gi = [&](int t, float4 (&dst)[3]) {
#pragma unroll
        for (int g = 0; g < 3; ++g) dst[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bvalid) {
            const float* gp = gi + ((size_t)b * S + t) * 1536 + dir * 768 + u0;
#pragma unroll
            for (int g = 0; g < 3; ++g) dst[g] = *reinterpret_cast<const float4*>(gp + g * 256);
        }
    };
    load_gi(dir ? S - 1 : 0, gin);
    const int frag_off = n * GQ_ROWB + kg * 16;              // this lane's chunk inside a k-step of an h plane row
    // receive role: thread t fetches what thread t of each of the three other quarters stored (same utterance n, same units
    // relative to the quarter): granule order in the buffer is [wave][store 0 | store 1][lane][2], so that every store and
    // every poll instruction covers 1 KB of consecutive bytes = 16 whole cache lines
    const int xoff = wv * 256 + lane * 2;                    // granule offset of this thread inside a (parity, quarter) block
    bool timed_out = false;

    for (int step = 0; step < S; ++step) {
        const int t = dir ? (S - 1 - step) : step;
        float4 gcur[3] = {gin[0], gin[1], gin[2]};
        const unsigned char* hb = qlds + (step & 1) * GQ_BUFB;
        unsigned char* hnb = qlds + ((step + 1) & 1) * GQ_BUFB;

        // ---- W_hh h on the matrix cores ---------------------------------------------------------------
        f32x4_t acc[3], accx[3];                             // accx: the two cross terms, 2^11 too large
#pragma unroll
        for (int g = 0; g < 3; ++g) { acc[g] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; accx[g] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
        // all 16 h fragments are read before the first MFMA (pinned): one wave per SIMD has nothing else to hide
        // the LDS latency behind, and the compiler otherwise reads each fragment right before its use
        f16x8 hf[8][GQ_NPL];
        if (!(dbg & 2)) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int p = 0; p < GQ_NPL; ++p)
                hf[s][p] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(hb + p * GQ_PLANEB + frag_off + s * 64));
        __builtin_amdgcn_sched_barrier(0);
        gq_mfma_enter(acc, accx);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int g = 0; g < 3; ++g) gq_mfma_aw(accx[g], wf[g][s][1], hf[s][0]);
#pragma unroll
            for (int g = 0; g < 3; ++g) gq_mfma_aw(accx[g], wf[g][s][0], hf[s][1]);
#pragma unroll
            for (int g = 0; g < 3; ++g) gq_mfma_aw(acc[g], wf[g][s][0], hf[s][0]);
        }
        gq_mfma_fence(acc, accx);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[g][j] = fmaf(accx[g][j], H3_LO_INV, acc[g][j]);
        }
        if constexpr (ROLES) {                               // matrix layout -> gate layout (the image was last read before the step's closing barrier)
#pragma unroll
            for (int g = 0; g < 3; ++g)
                *reinterpret_cast<float4*>(tl + (n * 3 + g) * GQ_TROW + wv * 16 + kg * 4) = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
            __syncthreads();
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const float4 v = *reinterpret_cast<const float4*>(tl + (gutt * 3 + g) * GQ_TROW + gul0);
                acc[g][0] = v.x; acc[g][1] = v.y; acc[g][2] = v.z; acc[g][3] = v.w;
            }
        }

        // ---- gates for 4 units x 1 utterance ------------------------------------------------------------
        float hn4[4], r4[4], z4[4], n4[4], hh4[4];
        const float gr[4] = {gcur[0].x, gcur[0].y, gcur[0].z, gcur[0].w}, gz[4] = {gcur[1].x, gcur[1].y, gcur[1].z, gcur[1].w};
        const float gn[4] = {gcur[2].x, gcur[2].y, gcur[2].z, gcur[2].w};
        const float br[4] = {bh[0].x, bh[0].y, bh[0].z, bh[0].w}, bz[4] = {bh[1].x, bh[1].y, bh[1].z, bh[1].w};
        const float bn[4] = {bh[2].x, bh[2].y, bh[2].z, bh[2].w};
        const float hp[4] = {hprev.x, hprev.y, hprev.z, hprev.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float hr = acc[0][j] + br[j], hz = acc[1][j] + bz[j];
            hh4[j] = acc[2][j] + bn[j];
            r4[j] = gq_sigmoid(gr[j] + hr);
            z4[j] = gq_sigmoid(gz[j] + hz);
            n4[j] = gq_tanh(gn[j] + r4[j] * hh4[j]);
            hn4[j] = (1.0f - z4[j]) * n4[j] + z4[j] * hp[j];
        }
        hprev = make_float4(hn4[0], hn4[1], hn4[2], hn4[3]);

        // ---- publish: own LDS planes (next parity) + one granule per value for the three other quarters ----
        uint2 ph, pl;
        split2h_quad(hprev, ph, pl);
        {
            unsigned char* d = hnb + gutt * GQ_ROWB + u0 * 2;
            *reinterpret_cast<uint2*>(d) = ph;
            *reinterpret_cast<uint2*>(d + GQ_PLANEB) = pl;
        }
        if (!(dbg & 4) && !((dbg & 8) && q == 3)) {
            unsigned long long* gs = xc + ((size_t)(step & 1) * 4 + q) * (GQ_NU * GQ_UQ) + xoff;
            const unsigned long long tag = (unsigned long long)((epoch << 9) | (unsigned)(step + 1)) << 48;
            const unsigned hh_[4] = {ph.x & 0xFFFFu, ph.x >> 16, ph.y & 0xFFFFu, ph.y >> 16};
            const unsigned ll_[4] = {pl.x & 0xFFFFu, pl.x >> 16, pl.y & 0xFFFFu, pl.y >> 16};
            unsigned long long gr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) gr[j] = tag | ((unsigned long long)hh_[j] << 32) | ((unsigned long long)ll_[j] << 16);
            // the lane's four granules go out as two 16-byte write-through stores (sc1 = what a relaxed agent-scope
            // atomic store compiles to) instead of four 8-byte ones -- half the store instructions and memory transactions
            // in front of the poll.  Every granule carries its own tag, so nothing depends on the pair landing together.
            const gq_u32x4 g01 = {(unsigned)gr[0], (unsigned)(gr[0] >> 32), (unsigned)gr[1], (unsigned)(gr[1] >> 32)};
            const gq_u32x4 g23 = {(unsigned)gr[2], (unsigned)(gr[2] >> 32), (unsigned)gr[3], (unsigned)(gr[3] >> 32)};
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:1024 sc1"
                         :: "v"(gs), "v"(g01), "v"(g23) : "memory");
        }
        // ---- receive the other three quarters' values of this step into the next-parity planes ------------
        if (step + 1 < S && !(dbg & 4)) {
            // all 12 granule loads of a poll round are independent (issued back to back, one wait); a round is
            // repeated as a whole until every tag matches -- a per-granule retry serialises the round trips
            unsigned long long v[3][4];
            unsigned spins = 0;
            const unsigned long long want = (unsigned long long)((epoch << 9) | (unsigned)(step + 1));
            const unsigned long long* src0 = xc + (size_t)(step & 1) * 4 * (GQ_NU * GQ_UQ) + xoff;
            // The partners' granules need ~0.8 us to become visible.  A poll round issued at once reaches memory before them,
            // comes back stale after ~0.7 us and the round that succeeds starts only then; a first round that leaves ~0.45 us
            // later is the one that succeeds (measured per layer launch: 106 us with no delay, 102.4 at 8 x 64 cycles,
            // 100.4 at 16, 102 at 20, 107 at 24).  dbg bits 8-12 add to the delay (experiments).
            for (int i = 0; i < ((dbg >> 8) & 31); ++i) __builtin_amdgcn_s_sleep(1);
            for (;;) {
                {
                    // six 16-byte loads (sc1: past the non-coherent caches, like the relaxed agent-scope atomic loads they replace),
                    // issued back to back, one wait
                    const unsigned long long* p0 = src0 + (size_t)(q <= 0 ? 1 : 0) * GQ_NU * GQ_UQ;
                    const unsigned long long* p1 = src0 + (size_t)(q <= 1 ? 2 : 1) * GQ_NU * GQ_UQ;
                    const unsigned long long* p2 = src0 + (size_t)(q <= 2 ? 3 : 2) * GQ_NU * GQ_UQ;
                    gq_u32x4 a0, a1, b0, b1, c0, c1;
                    asm volatile("global_load_dwordx4 %0, %6, off sc1\n\t"
                                 "global_load_dwordx4 %1, %6, off offset:1024 sc1\n\t"
                                 "global_load_dwordx4 %2, %7, off sc1\n\t"
                                 "global_load_dwordx4 %3, %7, off offset:1024 sc1\n\t"
                                 "global_load_dwordx4 %4, %8, off sc1\n\t"
                                 "global_load_dwordx4 %5, %8, off offset:1024 sc1\n\t"
                                 "s_waitcnt vmcnt(0)"
                                 : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1), "=&v"(c0), "=&v"(c1)
                                 : "v"(p0), "v"(p1), "v"(p2) : "memory");
                    auto u64 = [](unsigned lo, unsigned hi) { return (unsigned long long)lo | ((unsigned long long)hi << 32); };
                    v[0][0] = u64(a0.x, a0.y); v[0][1] = u64(a0.z, a0.w); v[0][2] = u64(a1.x, a1.y); v[0][3] = u64(a1.z, a1.w);
                    v[1][0] = u64(b0.x, b0.y); v[1][1] = u64(b0.z, b0.w); v[1][2] = u64(b1.x, b1.y); v[1][3] = u64(b1.z, b1.w);
                    v[2][0] = u64(c0.x, c0.y); v[2][1] = u64(c0.z, c0.w); v[2][2] = u64(c1.x, c1.y); v[2][3] = u64(c1.z, c1.w);
                }
                bool ok = true;
#pragma unroll
                for (int qi = 0; qi < 3; ++qi)
#pragma unroll
                    for (int j = 0; j < 4; ++j) ok = ok && ((v[qi][j] >> 48) == want);
                if (ok || (dbg & 1) || timed_out) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > ((dbg & 16) ? 4096u : GQ_SPIN_LIMIT)) {      // give up for good: later steps do not spin again
                    __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    timed_out = true;
                    break;
                }
            }
#pragma unroll
            for (int qi = 0; qi < 3; ++qi) {
                const int qs = qi + (qi >= q ? 1 : 0);
                uint2 rh, rl;
                rh.x = (unsigned)((v[qi][0] >> 32) & 0xFFFFu) | ((unsigned)((v[qi][1] >> 32) & 0xFFFFu) << 16);
                rh.y = (unsigned)((v[qi][2] >> 32) & 0xFFFFu) | ((unsigned)((v[qi][3] >> 32) & 0xFFFFu) << 16);
                rl.x = (unsigned)((v[qi][0] >> 16) & 0xFFFFu) | ((unsigned)((v[qi][1] >> 16) & 0xFFFFu) << 16);
                rl.y = (unsigned)((v[qi][2] >> 16) & 0xFFFFu) | ((unsigned)((v[qi][3] >> 16) & 0xFFFFu) << 16);
                unsigned char* d = hnb + gutt * GQ_ROWB + (qs * GQ_UQ + gul0) * 2;
                *reinterpret_cast<uint2*>(d) = rh;
                *reinterpret_cast<uint2*>(d + GQ_PLANEB) = rl;
            }
        }
        // the stores of this step and the loads of the next one are issued only now: a hand-off costs what sits in the
        // CONSUMER CU's memory queue ahead of the poll (MI355X_MICROARCH.md, handoff-1to1: 0.8 us idle, 2.5-2.9 loaded)
        if (bvalid) {
            const size_t yidx = ((size_t)b * S + t) * 512 + dir * 256 + u0;
            *reinterpret_cast<float4*>(y + yidx) = hprev;
            if (yplanes) {
                const size_t plane = (size_t)B * S * 512;
                *reinterpret_cast<uint2*>(yplanes + yidx) = ph;          // (the planes this step published: the same split)
                *reinterpret_cast<uint2*>(yplanes + plane + yidx) = pl;
            }
            if (SAVE) {
                float* gsv = gates + (((size_t)b * S + t) * 2 + dir) * 1024 + u0;
                *reinterpret_cast<float4*>(gsv) = make_float4(r4[0], r4[1], r4[2], r4[3]);
                *reinterpret_cast<float4*>(gsv + 256) = make_float4(z4[0], z4[1], z4[2], z4[3]);
                *reinterpret_cast<float4*>(gsv + 512) = make_float4(n4[0], n4[1], n4[2], n4[3]);
                *reinterpret_cast<float4*>(gsv + 768) = make_float4(hh4[0], hh4[1], hh4[2], hh4[3]);
            }
        }

        if (step + 1 < S) load_gi(dir ? (S - 2 - step) : step + 1, gin);
        __syncthreads();                                     // next-parity planes complete; this parity's reads are done
    }
}

// Measured with timing knock-outs (dbg), current kernel (103 us per layer launch = 4.1 us per step): without the MFMAs
// 66 us (MFMA phase 1.5 us), without the exchange 65 us (hand-off 1.5 us: stores + one ~0.7 us poll round + 0.4 us of
// waiting), with neither 29 us (gates, LDS planes, barrier, output stores: 1.2 us).  The hand-off was 2.1 us while a lane
// issued four 8-byte stores and twelve 8-byte polls into a [utterance][unit] granule layout (each instruction touched
// 32 cache lines in 16-byte pieces): 126 -> 120 us with 16-byte accesses, -> 108 us with the producer-thread layout,
// -> 103 us with the delayed first poll.
// Earlier measurements (8-byte granule accesses): of the ~5 us step, MFMA phase 1.3 us, gates + stores 0.5 us, exchange ~2.1 us.  A
// poll round issued with NO granule stores ahead of it in the same wave costs 0.6 us; behind the wave's own write-through
// stores it costs 2.1 us -- but the remote stores need about that long to become visible anyway.  Tried and removed: a
// dedicated publisher wave (re-reads the workgroup's values from LDS and stores all 1024 granules) with the other three
// waves polling store-free: 173 us per layer against 125 us (the extra barrier and the single wave's 16 stores lengthen
// the critical path more than the store-free polls shorten it).
// Tried and removed: THREE utterance groups per cluster visited round robin (a group's granules then have two whole phases,
// > 4 us, to arrive; 48 workgroups at batch 256): results identical, but a phase costs 2.5 us of MFMA + gates + barrier plus
// ~0.9 us for the poll round even though its data has long arrived, i.e. 250 us per layer; with 2-3 streams the pipeline
// ran at 344-370 k utterances/s against 390 k.  A timing knock-out of the whole exchange in THIS kernel (72 us per layer)
// bounds what hiding the exchange could give: 431 k with two streams (+11 %).
// Tried and removed: two utterance groups per cluster ("ping-pong": while group g's granules travel, group g^1
// runs its MFMA/gate phase on the same resident weights; 64 workgroups instead of 128 at batch 256).  Correct,
// but a phase took 4.1 us instead of the expected ~2.2 us (205 us per layer against 126 us), and with 2-3 HIP
// streams the whole pipeline was no faster (348-378 k against 375-389 k utterances/s).
