// Quad-workgroup GRU back-propagation through time on the matrix cores (round 4): the backward mirror of gru_quad_kernel.h.
//
// gru_bwd_pair_kernel.h computes  (W_hh^T dgh)  with fp32 FMAs: 768 per thread and step on all 256 CUs, ~2.6 us of FMA issue
// per step, 112-117 us per layer launch at batch 256 (profiles/r04/ab_bptt.txt).  Here, as in the forward recurrence:
//   * one CLUSTER of four workgroups owns 16 utterances of one direction for all S steps;
//   * workgroup q owns hidden units [64 q, 64 q + 64): it does their gate-gradient arithmetic and holds THEIR 192 gate rows of
//     W_hh for the whole sequence in registers -- transposed: A = W_hh^T tile (16 k x 32 own gate rows) as f16x2 planes
//     (f16_split.h), 4 destination quarters x 6 row steps x 2 planes x 4 VGPRs = 192 VGPRs per lane, one wave per SIMD;
//   * per step, a thread in its GATE role (utterance 4 wv + (lane >> 4), units 64 q + 4 (lane & 15) + j: 16-byte accesses,
//     256 contiguous bytes per 16 lanes) turns the carried d(h) into the three gate gradients, stores the dgi / dgh rows and puts
//     the hidden-side ones into LDS as the B operand dgh^T [own gate row][utterance] (two fp16 planes, double-buffered by parity);
//   * wave wv then forms, for EACH of the four quarters d, the 16 x 16 tile  sum_{own rows} W_hh[row][64 d + 16 wv + m] dgh[row][n]
//     (4 tiles x 6 row steps x 3 plane products = 72 v_mfma_f32_16x16x32_f16): the partial of (W_hh^T dgh) over this workgroup's
//     rows; a lane (n = lane & 15, kg = lane >> 4) holds units 16 wv + 4 kg + j of utterance n of each tile (MATRIX role);
//   * the tile of the own quarter stays in registers; the other three travel as 8-byte {value, tag} granules into the inbox of
//     their quarter (two granules per 16-byte write-through store, [wave][store][lane][2] order: every store / poll instruction
//     covers 1 KB of whole cache lines), and every thread completes its four values with what thread (wv, lane) of the three
//     other quarters sent: a reduce-scatter where the forward kernel has an all-gather.  Summation order is fixed: bit-reproducible.
//     The completed values go back to the gate role through a 4 KB LDS image (one more barrier per step).
// Accuracy: the f16x3 product (three MFMAs per fp32 product, hi * hi and the two cross terms in separate f32 accumulators); the
// gate gradients carry the loss scale of the backward (model_train.hip) and sit inside fp16's range like the dW operands.
#pragma once
#include "gru_quad_kernel.h"

constexpr int BQ_ROWB = 192 * 2 + 32;              // bytes per utterance row of one dgh plane: 26 sixteen-byte slots = 10 mod 16,
                                                   // which puts the 16 lanes of every ds_read_b128 group on 16 different slots
constexpr int BQ_PLANEB = GQ_NU * BQ_ROWB;
constexpr int BQ_BUFB = 2 * BQ_PLANEB;             // one parity buffer (hi, lo' planes): 13,312 B
constexpr int BQ_TROW = 64 + 4;                     // floats per utterance row of the matrix -> gate layout transposition image
constexpr size_t BQ_LDS_BYTES = 2 * (size_t)BQ_BUFB + (size_t)GQ_NU * BQ_TROW * 4;
constexpr int BQ_BLOCK = 4 * 2 * 64 * 2;           // granules of one (parity, destination, source) block: [wave][store 0 | 1][lane][2]
constexpr size_t BQ_XBUF_PER_CLUSTER = (size_t)2 * 4 * 4 * BQ_BLOCK * 8;      // [parity][destination quarter][source quarter] blocks

// dbg (timing experiments only, results invalid): bit 0 = do not wait for the granules, bit 1 = skip the MFMAs, bit 2 = skip
// publish + receive, bit 3 = no dgi / dgh stores, bit 4 = no input loads after the first step; bits 8-12 = delay of the first poll
// round in units of 64 cycles (results stay valid)
__global__ __launch_bounds__(GQ_THREADS) void gru_bwd_quad_kernel(
    const float* __restrict__ dy, const float* __restrict__ gates, const float* __restrict__ y, const float* __restrict__ whh0,
    const float* __restrict__ whh1, float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ bsum_i,
    float* __restrict__ bsum_h, int B, int S, unsigned long long* xbuf, unsigned int* status, unsigned epoch, int dbg,
    const uint4* __restrict__ wfrag0, const uint4* __restrict__ wfrag1) {
    // wfrag0 / 1: prep_whh_bwd_quad_elem output for direction 0 / 1 (gru_frag_prep.h; required)
    extern __shared__ __attribute__((aligned(16))) unsigned char bqlds[];
    // workgroup -> (quarter, cluster) as in gru_quad_kernel: the four quarters of a cluster on ONE XCD
    const int nclusters = gridDim.x >> 2;
    const bool xcd_order = (nclusters & 7) == 0;
    const int q = xcd_order ? (blockIdx.x >> 3) & 3 : blockIdx.x & 3;
    const int cluster = xcd_order ? (blockIdx.x & 7) + 8 * (blockIdx.x >> 5) : blockIdx.x >> 2;
    const int dir = cluster & 1, grp = cluster >> 1;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 15, kg = lane >> 4;
    unsigned long long* xc = xbuf + (size_t)cluster * (2 * 4 * 4 * BQ_BLOCK);

    // ---- resident weights: A fragments (rows m = k index inside the tile, columns = own gate rows) -------------------
    // own gate row kk in [0, 192): gate kk >> 6, unit 64 q + (kk & 63) -> row (kk >> 6) * 256 + 64 q + (kk & 63) of W_hh
    f16x8 wf[4][6][2];                                       // (loaded straight into accumulation registers: gq_mfma_aw, gru_quad_kernel.h)
    {
        const uint4* wsrc = (dir ? wfrag1 : wfrag0) + (size_t)((q * 4 + wv) * 4 * 6 * 2) * 64 + lane;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int s = 0; s < 6; ++s)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#ifdef SIR_GQ_BUILTIN_MFMA
                    wf[d][s][p] = __builtin_bit_cast(f16x8, wsrc[((d * 6 + s) * 2 + p) * 64]);
#else
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(wf[d][s][p]) : "v"(wsrc + ((d * 6 + s) * 2 + p) * 64));
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // ---- two thread roles ------------------------------------------------------------------------------------------------
    // MATRIX role (the MFMA result layout): lane (n = lane & 15, kg = lane >> 4) of wave wv holds units 16 wv + 4 kg + j of utterance n.
    // GATE role: lane (uq = lane & 15, us = lane >> 4) of wave wv does units 4 uq + j of utterance 4 wv + us -- the 16 lanes of a
    // load / store pass are then 256 contiguous bytes of ONE utterance row.  Doing the gate arithmetic in the matrix layout
    // instead (first version) put neighbouring lanes 150 KB apart: 64 separate 16-byte requests per instruction, ~0.11 us per
    // instruction and step on the CU's address unit, 12 instructions per step (timing knock-outs in profiles/r04/ab_bptt.txt: loads
    // + stores 30-48 us of a 117-137 us launch; one unit x four utterances per lane with 4-byte accesses: still 25-45 us).  The
    // completed (W_hh^T dgh) goes from the matrix to the gate layout through a 4 KB LDS image and one more barrier per step.
    const int uq = lane & 15, us = lane >> 4;
    const int bg = grp * GQ_NU + 4 * wv + us;                // the gate role's utterance
    const bool bvalid = bg < B;
    const int ulg = 4 * uq, ug = q * GQ_UQ + ulg;            // first of the gate role's 4 units: inside the quarter / of the layer
    float* tl = reinterpret_cast<float*>(bqlds + 2 * BQ_BUFB);              // [16 utterances][BQ_TROW] floats
    struct In { float4 r, z, nn, hn, hp, dy; };
    const float4 f4z = make_float4(0.f, 0.f, 0.f, 0.f);
    // inputs are fetched two steps ahead, right behind a poll, when the memory queue is empty
    auto fetch = [&](int it_, In& o) {
        o.r = f4z; o.z = f4z; o.nn = f4z; o.hn = f4z; o.hp = f4z; o.dy = f4z;
        if (!bvalid || it_ >= S) return;
        const int step_ = S - 1 - it_;
        const int t_ = dir ? (S - 1 - step_) : step_;
        const int tp_ = dir ? t_ + 1 : t_ - 1;
        const size_t row_ = (size_t)bg * S + t_;
        const float* gs = gates + (row_ * 2 + dir) * 1024 + ug;
        o.r = *reinterpret_cast<const float4*>(gs);
        o.z = *reinterpret_cast<const float4*>(gs + 256);
        o.nn = *reinterpret_cast<const float4*>(gs + 512);
        o.hn = *reinterpret_cast<const float4*>(gs + 768);
        if (step_ > 0) o.hp = *reinterpret_cast<const float4*>(y + ((size_t)bg * S + tp_) * 512 + dir * 256 + ug);
        o.dy = *reinterpret_cast<const float4*>(dy + row_ * 512 + dir * 256 + ug);
    };
    In cur, nxt;
    fetch(0, cur);
    fetch(1, nxt);

    float dhs[4] = {0.f, 0.f, 0.f, 0.f};                     // (W_hh^T dgh) of the gate role's units, from the step before
    float dhz[4] = {0.f, 0.f, 0.f, 0.f};                     // d(h) carried through the z gate
    float sum_r[4] = {0.f, 0.f, 0.f, 0.f}, sum_z[4] = {0.f, 0.f, 0.f, 0.f}, sum_n[4] = {0.f, 0.f, 0.f, 0.f}, sum_nr[4] = {0.f, 0.f, 0.f, 0.f};
    const int frag_off = n * BQ_ROWB + kg * 16;              // B fragment of dgh^T: column = utterance n, 8 own rows per lane
    const int xoff = wv * 256 + lane * 2;                    // granule offset of this thread inside a block
    bool timed_out = false;

    for (int it = 0; it < S; ++it) {
        const int step = S - 1 - it;
        const int t = dir ? (S - 1 - step) : step;
        unsigned char* pb = bqlds + (it & 1) * BQ_BUFB;

        // ---- gate gradients of 4 units x 1 utterance (gate role) ---------------------------------------------------------
        const float r[4] = {cur.r.x, cur.r.y, cur.r.z, cur.r.w}, zg[4] = {cur.z.x, cur.z.y, cur.z.z, cur.z.w};
        const float nn[4] = {cur.nn.x, cur.nn.y, cur.nn.z, cur.nn.w}, hn[4] = {cur.hn.x, cur.hn.y, cur.hn.z, cur.hn.w};
        const float hp[4] = {cur.hp.x, cur.hp.y, cur.hp.z, cur.hp.w}, dyv[4] = {cur.dy.x, cur.dy.y, cur.dy.z, cur.dy.w};
        float drp[4], dzp[4], dnp[4], dnr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float dh = dyv[j] + dhz[j] + dhs[j];
            const float dn = dh * (1.0f - zg[j]);
            const float dz = dh * (hp[j] - nn[j]);
            dnp[j] = dn * (1.0f - nn[j] * nn[j]);
            drp[j] = dnp[j] * hn[j] * r[j] * (1.0f - r[j]);
            dzp[j] = dz * zg[j] * (1.0f - zg[j]);
            dnr[j] = dnp[j] * r[j];
            dhz[j] = dh * zg[j];
            sum_r[j] += drp[j]; sum_z[j] += dzp[j]; sum_n[j] += dnp[j]; sum_nr[j] += dnr[j];
        }
        const float4 drp4 = make_float4(drp[0], drp[1], drp[2], drp[3]), dzp4 = make_float4(dzp[0], dzp[1], dzp[2], dzp[3]);
        const float4 dnp4 = make_float4(dnp[0], dnp[1], dnp[2], dnp[3]), dnr4 = make_float4(dnr[0], dnr[1], dnr[2], dnr[3]);

        if (it + 1 < S) {
            // ---- dgh^T planes of this step: [utterance][own gate row 64 g + ulg + j] ---------------------------------------
            {
                uint2 h0, l0, h1, l1, h2, l2;
                split2h_quad(drp4, h0, l0);
                split2h_quad(dzp4, h1, l1);
                split2h_quad(dnr4, h2, l2);
                unsigned char* dst = pb + (4 * wv + us) * BQ_ROWB + ulg * 2;
                *reinterpret_cast<uint2*>(dst) = h0;
                *reinterpret_cast<uint2*>(dst + 128) = h1;
                *reinterpret_cast<uint2*>(dst + 256) = h2;
                *reinterpret_cast<uint2*>(dst + BQ_PLANEB) = l0;
                *reinterpret_cast<uint2*>(dst + BQ_PLANEB + 128) = l1;
                *reinterpret_cast<uint2*>(dst + BQ_PLANEB + 256) = l2;
            }
            __syncthreads();          // planes of this parity complete (the other parity was last read before the previous barrier A);
                                      // every gate-role thread has read the transposition image of the step before

            // ---- partial of W_hh^T dgh over the own rows, one 16 x 16 tile (units x utterances) per destination quarter ----
            f32x4_t acc[4], accx[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) { acc[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; accx[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
            if (!(dbg & 2)) {
                f16x8 gf[6][2];
#pragma unroll
                for (int s = 0; s < 6; ++s)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        gf[s][p] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(pb + p * BQ_PLANEB + frag_off + s * 64));
                __builtin_amdgcn_sched_barrier(0);
                gq_mfma_enter(acc, accx);
#pragma unroll
                for (int s = 0; s < 6; ++s) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) gq_mfma_aw(accx[d], wf[d][s][1], gf[s][0]);
#pragma unroll
                    for (int d = 0; d < 4; ++d) gq_mfma_aw(accx[d], wf[d][s][0], gf[s][1]);
#pragma unroll
                    for (int d = 0; d < 4; ++d) gq_mfma_aw(acc[d], wf[d][s][0], gf[s][0]);
                }
                gq_mfma_fence(acc, accx);
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[d][j] = fmaf(accx[d][j], H3_LO_INV, acc[d][j]);
            }

            // ---- publish the tiles of the three other quarters into their inboxes -------------------------------------
            const unsigned tagv = (epoch << 16) | (unsigned)(it + 1);
            if (!(dbg & 4)) {
#pragma unroll
                for (int di = 0; di < 3; ++di) {
                    const int d = di + (di >= q ? 1 : 0);
                    f32x4_t pv = acc[0];
                    pv = d == 1 ? acc[1] : pv; pv = d == 2 ? acc[2] : pv; pv = d == 3 ? acc[3] : pv;
                    unsigned long long* gs = xc + ((size_t)((it & 1) * 4 + d) * 4 + q) * BQ_BLOCK + xoff;
                    const gq_u32x4 g01 = {__float_as_uint(pv[0]), tagv, __float_as_uint(pv[1]), tagv};
                    const gq_u32x4 g23 = {__float_as_uint(pv[2]), tagv, __float_as_uint(pv[3]), tagv};
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:1024 sc1"
                                 :: "v"(gs), "v"(g01), "v"(g23) : "memory");
                }
            }
            f32x4_t own = acc[0];
            own = q == 1 ? acc[1] : own; own = q == 2 ? acc[2] : own; own = q == 3 ? acc[3] : own;

            // ---- receive the three other quarters' partials of the own units ------------------------------------------
            float rv[3][4];
#pragma unroll
            for (int si = 0; si < 3; ++si)
#pragma unroll
                for (int j = 0; j < 4; ++j) rv[si][j] = 0.0f;
            if (!(dbg & 4)) {
                const unsigned long long* src0 = xc + (size_t)((it & 1) * 4 + q) * 4 * BQ_BLOCK + xoff;
                const unsigned long long* p0 = src0 + (size_t)(q <= 0 ? 1 : 0) * BQ_BLOCK;
                const unsigned long long* p1 = src0 + (size_t)(q <= 1 ? 2 : 1) * BQ_BLOCK;
                const unsigned long long* p2 = src0 + (size_t)(q <= 2 ? 3 : 2) * BQ_BLOCK;
                unsigned spins = 0;
                for (int i = 0; i < ((dbg >> 8) & 31); ++i) __builtin_amdgcn_s_sleep(1);      // (see gru_quad_kernel's poll loop)
                for (;;) {
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
                    rv[0][0] = __uint_as_float(a0.x); rv[0][1] = __uint_as_float(a0.z); rv[0][2] = __uint_as_float(a1.x); rv[0][3] = __uint_as_float(a1.z);
                    rv[1][0] = __uint_as_float(b0.x); rv[1][1] = __uint_as_float(b0.z); rv[1][2] = __uint_as_float(b1.x); rv[1][3] = __uint_as_float(b1.z);
                    rv[2][0] = __uint_as_float(c0.x); rv[2][1] = __uint_as_float(c0.z); rv[2][2] = __uint_as_float(c1.x); rv[2][3] = __uint_as_float(c1.z);
                    const bool ok = a0.y == tagv && a0.w == tagv && a1.y == tagv && a1.w == tagv && b0.y == tagv && b0.w == tagv &&
                                    b1.y == tagv && b1.w == tagv && c0.y == tagv && c0.w == tagv && c1.y == tagv && c1.w == tagv;
                    if (ok || (dbg & 1) || timed_out) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > GQ_SPIN_LIMIT) {                  // give up for good: later steps do not spin again
                        __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        timed_out = true;
                        break;
                    }
                }
            }
            // fixed order: the own partial, then the other quarters by rising index; matrix layout -> gate layout through LDS
            *reinterpret_cast<float4*>(tl + n * BQ_TROW + 16 * wv + 4 * kg) =
                make_float4(((own[0] + rv[0][0]) + rv[1][0]) + rv[2][0], ((own[1] + rv[0][1]) + rv[1][1]) + rv[2][1],
                            ((own[2] + rv[0][2]) + rv[1][2]) + rv[2][2], ((own[3] + rv[0][3]) + rv[1][3]) + rv[2][3]);
            __syncthreads();
            const float4 dv = *reinterpret_cast<const float4*>(tl + (4 * wv + us) * BQ_TROW + ulg);
            dhs[0] = dv.x; dhs[1] = dv.y; dhs[2] = dv.z; dhs[3] = dv.w;
        }

        // ---- this step's rows of dgi / dgh and the inputs of the step after next: issued behind the poll (what sits in the CU's
        //      memory queue ahead of a poll lengthens the hand-off) ------------------------------------------------------------
        if (bvalid && !(dbg & 8)) {
            const size_t row = (size_t)bg * S + t;
            float* gi_o = dgi + row * 1536 + dir * 768 + ug;
            float* gh_o = dgh + row * 1536 + dir * 768 + ug;
            *reinterpret_cast<float4*>(gi_o) = drp4;
            *reinterpret_cast<float4*>(gi_o + 256) = dzp4;
            *reinterpret_cast<float4*>(gi_o + 512) = dnp4;
            *reinterpret_cast<float4*>(gh_o) = drp4;
            *reinterpret_cast<float4*>(gh_o + 256) = dzp4;
            *reinterpret_cast<float4*>(gh_o + 512) = dnr4;
        }
        cur = nxt;
        if (!(dbg & 16) || it == 0) fetch(it + 2, nxt);
    }
    if (bvalid) {
        float* bi = bsum_i + (size_t)bg * 1536 + dir * 768 + ug;
        float* bh = bsum_h + (size_t)bg * 1536 + dir * 768 + ug;
        const float4 sr = make_float4(sum_r[0], sum_r[1], sum_r[2], sum_r[3]), sz = make_float4(sum_z[0], sum_z[1], sum_z[2], sum_z[3]);
        *reinterpret_cast<float4*>(bi) = sr;
        *reinterpret_cast<float4*>(bi + 256) = sz;
        *reinterpret_cast<float4*>(bi + 512) = make_float4(sum_n[0], sum_n[1], sum_n[2], sum_n[3]);
        *reinterpret_cast<float4*>(bh) = sr;
        *reinterpret_cast<float4*>(bh + 256) = sz;
        *reinterpret_cast<float4*>(bh + 512) = make_float4(sum_nr[0], sum_nr[1], sum_nr[2], sum_nr[3]);
    }
}
