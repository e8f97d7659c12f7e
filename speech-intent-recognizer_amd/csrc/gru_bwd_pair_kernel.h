// Paired-workgroup GRU back-propagation through time (mirror of gru_pair_kernel for the backward pass).
//
// gru_bwd_kernel re-streams W_hh (786 KB per direction) from L2 at every step: 238 us per layer at batch 256.
// Here, as in the forward pair kernel, TWO workgroups share 4 utterances of one direction and workgroup `half`
// keeps the 384 gate rows of ITS 128 hidden units resident for all S steps (10/16 in registers, 6/16 in LDS) --
// but laid out for the transposed product: thread (k pair, row part) holds W[row][k] and W[row][k + 128] for 96 of
// the 384 own rows (two k per thread halve the LDS broadcast reads of dgh, which bound the first version).
// Per step:
//   1. thread (unit, utterance) does the gate-gradient arithmetic, stores the dgi/dgh rows and puts the three
//      hidden-side gate gradients of its unit into LDS;
//   2. every thread accumulates  sum_{own rows} W[row][k] * dgh[row]  for its two k and its row part (768 FMAs,
//      dgh broadcast from LDS), the four row parts are summed with two DPP exchanges: a PARTIAL of
//      (W_hh^T dgh)[k] over this half's rows, for all 256 k;
//   3. the partials of the k that belong to the PEER's units travel as 8-byte {tag, value} granules (same
//      recipe and buffer layout as gru_pair_kernel); the partials of the own k are completed with the peer's
//      granules and become next step's carried dh.
// No atomics, fixed summation order: bit-reproducible.  Bias-gradient partial sums as in gru_bwd_kernel.
#pragma once
#include "gru_pair_kernel.h"

constexpr int GBP_G4 = 24;                     // groups of 4 own rows per row part (96 rows), for each of the thread's two k
constexpr int GBP_REG1 = 6;                    // groups of the second k kept in registers (the first k: all 24) -> 120 VGPRs
constexpr int GBP_LDS4 = GBP_G4 - GBP_REG1;    // groups of the second k kept in LDS (18 x 8 KB)
// dgh of the own rows in LDS: [utterance][row part][GBP_RPS].  A ds_read_b128 of the product loop serves 16 lanes = 4 row parts x 4
// lanes reading the same 16 bytes; with the parts 96 floats = 384 B apart (round 3) parts 0 / 2 and 1 / 3 sat on the same banks
// (384 = 128 mod 256): a 2-way conflict on all 96 reads per thread and step (PMC r03: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE =
// 0.45, the LDS pipe busy 57 % of the kernel).  100 floats = 400 B puts the four parts at byte 0 / 144 / 32 / 176 of the 256-byte
// bank line: four disjoint 16-byte slots.
constexpr int GBP_RPS = 100;
constexpr int gbp_gs(bool pad) { return pad ? 4 * GBP_RPS : 384; }      // floats per utterance
constexpr size_t GBP_LDS_BYTES = ((size_t)GBP_LDS4 * GP_THREADS * 4 + GP_BW * gbp_gs(true) + GP_BW * GP_UH) * 4;

// KNOCK (timing knock-outs, SIR_BPTT_KNOCK; results invalid unless 0): bit 0 = no exchange with the peer (no granule store, no
// poll), bit 1 = no product loop, bit 2 = product loop without its dgh reads from LDS, bit 3 = product loop without the LDS-resident
// weights (the register-resident ones stand in).  PAD = the padded dgh image above (false: round 3's), PK = the product loop on
// v_pk_fma_f32 (even / odd partial sums per accumulator, added once per step).
template <int KNOCK = 0, bool PAD = true, bool PK = false>
__global__ __launch_bounds__(GP_THREADS) void gru_bwd_pair_kernel(
    const float* __restrict__ dy, const float* __restrict__ gates, const float* __restrict__ y, const float* __restrict__ whh0,
    const float* __restrict__ whh1, float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ bsum_i,
    float* __restrict__ bsum_h, int B, int S, float* xbuf, unsigned int* status, unsigned epoch) {
    extern __shared__ __attribute__((aligned(16))) float blds[];
    gp_f4* wl4 = reinterpret_cast<gp_f4*>(blds);                          // [GBP_LDS4][threads] float4 (4 consecutive own rows)
    float* gsh = blds + (size_t)GBP_LDS4 * GP_THREADS * 4;                // dgh of the own rows: [utterance][384]
    constexpr int GS = gbp_gs(PAD), RPS = PAD ? GBP_RPS : 96;
    float* dhs = gsh + GP_BW * gbp_gs(true);                              // (W_hh^T dgh) of the own units: [utterance][128]
    // the two halves of a pair on ONE XCD (workgroup L = x + gridDim.x y runs on XCD L % 8; gridDim.x = 2 pairs): x, x + 8 of a block of 16
    const int npairs_ = gridDim.x >> 1;
    const int dir = blockIdx.y;
    const int half = (npairs_ & 7) == 0 ? (blockIdx.x >> 3) & 1 : blockIdx.x & 1;
    const int pair = (npairs_ & 7) == 0 ? (blockIdx.x & 7) + 8 * (blockIdx.x >> 4) : blockIdx.x >> 1;
    const int b0 = pair * GP_BW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* __restrict__ whh = dir ? whh1 : whh0;
    unsigned long long* xg = reinterpret_cast<unsigned long long*>(xbuf) + (size_t)(pair * 2 + dir) * 2 * 2 * GP_BW * GP_UH;

    // ---- matvec role: k pair kp = 16 wave + lane / 4 (k0 = kp, k1 = kp + 128), row part rp = lane & 3 ----------
    const int kp = (wv << 4) | (lane >> 2), rp = lane & 3;
    gp_f4 wr0[GBP_G4], wr1[GBP_REG1];
    {
        auto grow = [&](int ro) { return (ro >> 7) * 256 + half * GP_UH + (ro & 127); };     // own row -> row of W_hh
#pragma unroll
        for (int i = 0; i < GBP_G4; ++i) {
            const int ro = rp * 96 + 4 * i;
            const float* w0 = whh + (size_t)grow(ro) * 256, *w1 = whh + (size_t)grow(ro + 1) * 256;
            const float* w2 = whh + (size_t)grow(ro + 2) * 256, *w3 = whh + (size_t)grow(ro + 3) * 256;
            gp_f4 a, c;
            a.x = w0[kp]; a.y = w1[kp]; a.z = w2[kp]; a.w = w3[kp];
            c.x = w0[kp + 128]; c.y = w1[kp + 128]; c.z = w2[kp + 128]; c.w = w3[kp + 128];
            wr0[i] = a;
            if (i < GBP_REG1) wr1[i < GBP_REG1 ? i : 0] = c;
            else wl4[(size_t)(i - GBP_REG1) * GP_THREADS + tid] = c;
        }
    }
    // ---- gate role: unit ul = tid & 127 of this half, utterance bq = tid >> 7 ----------------------------
    const int ul = tid & 127, bq = tid >> 7, u = half * GP_UH + ul;
    const bool bvalid = (b0 + bq) < B;
    // LDS slot of own row ro = 128 g + ul (gate g): row part ro / 96, position ro % 96 inside it
    int gslot[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) { const int ro = 128 * g + ul; gslot[g] = bq * GS + (ro / 96) * RPS + ro % 96; }
    for (int i = tid; i < GP_BW * GP_UH; i += GP_THREADS) dhs[i] = 0.0f;
    float dhz = 0.0f;
    float sum_r = 0.f, sum_z = 0.f, sum_n = 0.f, sum_nr = 0.f;
    __syncthreads();

    // the six per-step inputs of the gate role are fetched one step ahead (their latency hides behind a whole step)
    float in_r = 0.f, in_z = 0.f, in_n = 0.f, in_hn = 0.f, in_hp = 0.f, in_dy = 0.f;
    auto fetch = [&](int it_) {
        if (!bvalid || it_ >= S) return;
        const int step_ = S - 1 - it_;
        const int t_ = dir ? (S - 1 - step_) : step_;
        const int tp_ = dir ? t_ + 1 : t_ - 1;
        const size_t row_ = (size_t)(b0 + bq) * S + t_;
        const float* gs = gates + (row_ * 2 + dir) * 1024;
        in_r = gs[u]; in_z = gs[256 + u]; in_n = gs[512 + u]; in_hn = gs[768 + u];
        in_hp = (step_ > 0) ? y[((size_t)(b0 + bq) * S + tp_) * 512 + dir * 256 + u] : 0.0f;
        in_dy = dy[row_ * 512 + dir * 256 + u];
    };
    fetch(0);
    for (int it = 0; it < S; ++it) {
        const int step = S - 1 - it;
        const int t = dir ? (S - 1 - step) : step;                // time index processed now
        float drp = 0.f, dzp = 0.f, dnp = 0.f, dnr = 0.f, dhz_new = 0.f;
        const float r = in_r, zg = in_z, nn = in_n, hn = in_hn, hprev = in_hp, dyv = in_dy;
        fetch(it + 1);
        if (bvalid) {
            const size_t row = (size_t)(b0 + bq) * S + t;
            const float dh = dyv + dhz + dhs[bq * GP_UH + ul];
            const float dn = dh * (1.0f - zg);
            const float dz = dh * (hprev - nn);
            dnp = dn * (1.0f - nn * nn);
            drp = dnp * hn * r * (1.0f - r);
            dzp = dz * zg * (1.0f - zg);
            dnr = dnp * r;
            dhz_new = dh * zg;
            float* gi_o = dgi + row * 1536 + dir * 768;
            float* gh_o = dgh + row * 1536 + dir * 768;
            gi_o[u] = drp; gi_o[256 + u] = dzp; gi_o[512 + u] = dnp;
            gh_o[u] = drp; gh_o[256 + u] = dzp; gh_o[512 + u] = dnr;
            sum_r += drp; sum_z += dzp; sum_n += dnp; sum_nr += dnr;
        }
        dhz = dhz_new;
        gsh[gslot[0]] = drp; gsh[gslot[1]] = dzp; gsh[gslot[2]] = dnr;
        __syncthreads();                                          // gsh complete; everyone has consumed dhs

        // ---- partial of W_hh^T dgh over the own rows ----------------------------------------------------
        float acc0[GP_BW], acc1[GP_BW];
        const float* gpart = gsh + rp * RPS;
        if constexpr (!PK) {
#pragma unroll
            for (int bb = 0; bb < GP_BW; ++bb) { acc0[bb] = 0.0f; acc1[bb] = 0.0f; }
            auto fma8 = [&](const gp_f4& a, const gp_f4& c, int i) {
#pragma unroll
                for (int bb = 0; bb < GP_BW; ++bb) {
                    gp_f4 g4;
                    if (KNOCK & 4) g4 = a;
                    else g4 = *reinterpret_cast<const gp_f4*>(gpart + bb * GS + 4 * i);
                    acc0[bb] = fmaf(a.x, g4.x, acc0[bb]); acc0[bb] = fmaf(a.y, g4.y, acc0[bb]);
                    acc0[bb] = fmaf(a.z, g4.z, acc0[bb]); acc0[bb] = fmaf(a.w, g4.w, acc0[bb]);
                    acc1[bb] = fmaf(c.x, g4.x, acc1[bb]); acc1[bb] = fmaf(c.y, g4.y, acc1[bb]);
                    acc1[bb] = fmaf(c.z, g4.z, acc1[bb]); acc1[bb] = fmaf(c.w, g4.w, acc1[bb]);
                }
            };
            if (!(KNOCK & 2)) {
#pragma unroll
                for (int i = 0; i < GBP_REG1; ++i) fma8(wr0[i], wr1[i], i);
                asm volatile("" ::: "memory");                        // keep the LDS-resident weights in LDS
#pragma unroll
                for (int i = GBP_REG1; i < GBP_G4; ++i)
                    fma8(wr0[i], (KNOCK & 8) ? wr0[GBP_G4 - 1 - i] : wl4[(size_t)(i - GBP_REG1) * GP_THREADS + tid], i);
            }
        } else {
            // packed form: (even, odd) partial sums per accumulator -- two v_pk_fma_f32 per four products
            gp_f2 p0[GP_BW], p1[GP_BW];
#pragma unroll
            for (int bb = 0; bb < GP_BW; ++bb) { p0[bb] = gp_f2{0.0f, 0.0f}; p1[bb] = gp_f2{0.0f, 0.0f}; }
            auto fma8 = [&](const gp_f4& a, const gp_f4& c, int i) {
#pragma unroll
                for (int bb = 0; bb < GP_BW; ++bb) {
                    gp_f4 g4;
                    if (KNOCK & 4) g4 = a;
                    else g4 = *reinterpret_cast<const gp_f4*>(gpart + bb * GS + 4 * i);
                    p0[bb] = __builtin_elementwise_fma(a.xy, g4.xy, p0[bb]);
                    p0[bb] = __builtin_elementwise_fma(a.zw, g4.zw, p0[bb]);
                    p1[bb] = __builtin_elementwise_fma(c.xy, g4.xy, p1[bb]);
                    p1[bb] = __builtin_elementwise_fma(c.zw, g4.zw, p1[bb]);
                }
            };
            if (!(KNOCK & 2)) {
#pragma unroll
                for (int i = 0; i < GBP_REG1; ++i) fma8(wr0[i], wr1[i], i);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = GBP_REG1; i < GBP_G4; ++i)
                    fma8(wr0[i], (KNOCK & 8) ? wr0[GBP_G4 - 1 - i] : wl4[(size_t)(i - GBP_REG1) * GP_THREADS + tid], i);
            }
#pragma unroll
            for (int bb = 0; bb < GP_BW; ++bb) { acc0[bb] = p0[bb].x + p0[bb].y; acc1[bb] = p1[bb].x + p1[bb].y; }
        }
#pragma unroll
        for (int bb = 0; bb < GP_BW; ++bb) {                      // the four row parts: lanes 4j .. 4j + 3
            acc0[bb] += gp_quad_xor1(acc0[bb]); acc0[bb] += gp_quad_xor2(acc0[bb]);
            acc1[bb] += gp_quad_xor1(acc1[bb]); acc1[bb] += gp_quad_xor2(acc1[bb]);
        }
        // lane rp finishes utterance rp: one value for the own unit kp (k = kp + 128 half), one for the peer's
        float s0 = acc0[0], s1 = acc1[0];
#pragma unroll
        for (int bb = 1; bb < GP_BW; ++bb) { s0 = (rp == bb) ? acc0[bb] : s0; s1 = (rp == bb) ? acc1[bb] : s1; }
        const float vown = half ? s1 : s0, vpeer = half ? s0 : s1;
        const unsigned tagv = (epoch << 16) | (unsigned)(it + 1);     // {launch epoch of this buffer, step + 1}: stale granules never match
        if (it + 1 < S) {
            unsigned long long* gmine = xg + ((size_t)(it & 1) * 2 + half) * GP_BW * GP_UH;
            const unsigned long long* gpeer = xg + ((size_t)(it & 1) * 2 + (half ^ 1)) * GP_BW * GP_UH;
            unsigned long long pv = ((unsigned long long)tagv << 32) | __float_as_uint(vpeer);
            if (!(KNOCK & 1)) {
                __hip_atomic_store(gmine + rp * GP_UH + kp, pv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while ((unsigned)((pv = __hip_atomic_load(gpeer + rp * GP_UH + kp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != tagv) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > GP_SPIN_LIMIT) { __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
            }
            dhs[rp * GP_UH + kp] = vown + __uint_as_float((unsigned)pv);     // (dhs was last read before the barrier above)
        }
        __syncthreads();                                          // dhs of the next step complete
    }
    if (bvalid) {
        float* bi = bsum_i + (size_t)(b0 + bq) * 1536 + dir * 768;
        float* bh = bsum_h + (size_t)(b0 + bq) * 1536 + dir * 768;
        bi[u] = sum_r; bi[256 + u] = sum_z; bi[512 + u] = sum_n;
        bh[u] = sum_r; bh[256 + u] = sum_z; bh[512 + u] = sum_nr;
    }
}

// ------------------------------------------------------------------------------------------
// Second layout of the same kernel (round 4): FOUR CONSECUTIVE k per thread and EIGHT row parts.  The knock-outs of the two-k kernel
// (profiles/r04/ab_bptt.txt) put the product loop at 57 of its 125 us, 45 of them the dgh broadcast reads from LDS: 96
// ds_read_b128 per thread and step for 768 FMAs -- the loop is bound by LDS read issue, not by the FMAs.  Here thread
// (k quad kq = 8 wave + lane / 8, row part rp = lane & 7) holds W[row][4 kq + j], j = 0..3, for 48 of the 384 own rows: the same
// 192 weights (30 float4 in registers, 18 in LDS) and 768 FMAs, but HALF the dgh reads (48 per thread and step).  The eight row
// parts are summed with three DPP steps (quad xor 1, xor 2, row_half_mirror).  A wave's 32 k are consecutive and lie in ONE half
// of the hidden units (waves 0-3: units of half 0, waves 4-7: of half 1), which splits the exchange by wave: the four waves whose k
// are the PEER's units only store granules, the four whose k are the own units only poll; lane rp finishes utterance rp & 3 of the
// two k with j = 2 (rp >> 2) + {0, 1}, i.e. two neighbouring granules = one 16-byte write-through store / poll, and a wave
// instruction covers four runs of 256 contiguous bytes (whole cache lines: a first four-k version with k = kq + 64 j touched
// eight half lines per instruction and lost 19 us to the exchange).  dgh image: [utterance][8 parts][52] floats -- the parts 208 B
// apart sit on eight disjoint 16-byte slots of the 256-byte bank line.
// ------------------------------------------------------------------------------------------
constexpr int GB4_RPS = 52, GB4_GS = 8 * GB4_RPS;          // floats per row part / per utterance
constexpr int GB4_G = 12;                                   // groups of 4 own rows per row part (48 rows)
constexpr int GB4_REGJ = 3;                                 // groups of k index 2 and 3 kept in registers (k index 0, 1: all 12) -> 30 float4
constexpr int GB4_LDS4 = 2 * (GB4_G - GB4_REGJ);            // 18 float4 per thread in LDS
constexpr size_t GB4_LDS_BYTES = ((size_t)GB4_LDS4 * GP_THREADS * 4 + GP_BW * GB4_GS + GP_BW * GP_UH) * 4;
typedef unsigned int gb4_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float gp_half_mirror(float v) {   // lane i <-> lane 7 - i within each group of 8
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));
}

template <int KNOCK = 0>
__global__ __launch_bounds__(GP_THREADS) void gru_bwd_pair_k4_kernel(
    const float* __restrict__ dy, const float* __restrict__ gates, const float* __restrict__ y, const float* __restrict__ whh0,
    const float* __restrict__ whh1, float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ bsum_i,
    float* __restrict__ bsum_h, int B, int S, float* xbuf, unsigned int* status, unsigned epoch) {
    extern __shared__ __attribute__((aligned(16))) float blds[];
    gp_f4* wl4 = reinterpret_cast<gp_f4*>(blds);                          // [GB4_LDS4][threads] float4 (4 consecutive own rows)
    float* gsh = blds + (size_t)GB4_LDS4 * GP_THREADS * 4;                // dgh of the own rows: [utterance][8][52]
    float* dhs = gsh + GP_BW * GB4_GS;                                    // (W_hh^T dgh) of the own units: [utterance][128]
    const int npairs_ = gridDim.x >> 1;
    const int dir = blockIdx.y;
    const int half = (npairs_ & 7) == 0 ? (blockIdx.x >> 3) & 1 : blockIdx.x & 1;
    const int pair = (npairs_ & 7) == 0 ? (blockIdx.x & 7) + 8 * (blockIdx.x >> 4) : blockIdx.x >> 1;
    const int b0 = pair * GP_BW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* __restrict__ whh = dir ? whh1 : whh0;
    unsigned long long* xg = reinterpret_cast<unsigned long long*>(xbuf) + (size_t)(pair * 2 + dir) * 2 * 2 * GP_BW * GP_UH;

    // ---- matvec role ---------------------------------------------------------------------------------------
    const int kq = (wv << 3) | (lane >> 3), rp = lane & 7;                // k = 4 kq + j
    gp_f4 wa[2][GB4_G], wb[2][GB4_REGJ];                                  // k index 0, 1: all groups; k index 2, 3: the first GB4_REGJ
    {
        auto grow = [&](int ro) { return (ro >> 7) * 256 + half * GP_UH + (ro & 127); };     // own row -> row of W_hh
#pragma unroll
        for (int i = 0; i < GB4_G; ++i) {
            const int ro = rp * 48 + 4 * i;
            const float4 r0 = *reinterpret_cast<const float4*>(whh + (size_t)grow(ro) * 256 + 4 * kq);
            const float4 r1 = *reinterpret_cast<const float4*>(whh + (size_t)grow(ro + 1) * 256 + 4 * kq);
            const float4 r2 = *reinterpret_cast<const float4*>(whh + (size_t)grow(ro + 2) * 256 + 4 * kq);
            const float4 r3 = *reinterpret_cast<const float4*>(whh + (size_t)grow(ro + 3) * 256 + 4 * kq);
            const gp_f4 c[4] = {{r0.x, r1.x, r2.x, r3.x}, {r0.y, r1.y, r2.y, r3.y}, {r0.z, r1.z, r2.z, r3.z}, {r0.w, r1.w, r2.w, r3.w}};
            wa[0][i] = c[0];
            wa[1][i] = c[1];
            if (i < GB4_REGJ) { wb[0][i < GB4_REGJ ? i : 0] = c[2]; wb[1][i < GB4_REGJ ? i : 0] = c[3]; }
            else {
                wl4[(size_t)(i - GB4_REGJ) * GP_THREADS + tid] = c[2];
                wl4[(size_t)((GB4_G - GB4_REGJ) + i - GB4_REGJ) * GP_THREADS + tid] = c[3];
            }
        }
    }
    // ---- gate role: unit ul = tid & 127 of this half, utterance bq = tid >> 7 ----------------------------
    const int ul = tid & 127, bq = tid >> 7, u = half * GP_UH + ul;
    const bool bvalid = (b0 + bq) < B;
    int gslot[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) { const int ro = 128 * g + ul; gslot[g] = bq * GB4_GS + (ro / 48) * GB4_RPS + ro % 48; }
    for (int i = tid; i < GP_BW * GP_UH; i += GP_THREADS) dhs[i] = 0.0f;
    float dhz = 0.0f;
    float sum_r = 0.f, sum_z = 0.f, sum_n = 0.f, sum_nr = 0.f;
    __syncthreads();

    float in_r = 0.f, in_z = 0.f, in_n = 0.f, in_hn = 0.f, in_hp = 0.f, in_dy = 0.f;
    auto fetch = [&](int it_) {
        if (!bvalid || it_ >= S) return;
        const int step_ = S - 1 - it_;
        const int t_ = dir ? (S - 1 - step_) : step_;
        const int tp_ = dir ? t_ + 1 : t_ - 1;
        const size_t row_ = (size_t)(b0 + bq) * S + t_;
        const float* gs = gates + (row_ * 2 + dir) * 1024;
        in_r = gs[u]; in_z = gs[256 + u]; in_n = gs[512 + u]; in_hn = gs[768 + u];
        in_hp = (step_ > 0) ? y[((size_t)(b0 + bq) * S + tp_) * 512 + dir * 256 + u] : 0.0f;
        in_dy = dy[row_ * 512 + dir * 256 + u];
    };
    fetch(0);
    const int fb = rp & 3, fj = rp >> 2;                       // this lane finishes utterance fb of k index 2 fj and 2 fj + 1
    const bool own_wave = (wv >> 2) == half;                   // the wave's 32 k are this half's own units (else the peer's)
    const int kl = (4 * kq + 2 * fj) & 127;                    // unit index (inside its half) of the first of the lane's two k
    for (int it = 0; it < S; ++it) {
        const int step = S - 1 - it;
        const int t = dir ? (S - 1 - step) : step;
        float drp = 0.f, dzp = 0.f, dnp = 0.f, dnr = 0.f, dhz_new = 0.f;
        const float r = in_r, zg = in_z, nn = in_n, hn = in_hn, hprev = in_hp, dyv = in_dy;
        fetch(it + 1);
        if (bvalid) {
            const size_t row = (size_t)(b0 + bq) * S + t;
            const float dh = dyv + dhz + dhs[bq * GP_UH + ul];
            const float dn = dh * (1.0f - zg);
            const float dz = dh * (hprev - nn);
            dnp = dn * (1.0f - nn * nn);
            drp = dnp * hn * r * (1.0f - r);
            dzp = dz * zg * (1.0f - zg);
            dnr = dnp * r;
            dhz_new = dh * zg;
            float* gi_o = dgi + row * 1536 + dir * 768;
            float* gh_o = dgh + row * 1536 + dir * 768;
            gi_o[u] = drp; gi_o[256 + u] = dzp; gi_o[512 + u] = dnp;
            gh_o[u] = drp; gh_o[256 + u] = dzp; gh_o[512 + u] = dnr;
            sum_r += drp; sum_z += dzp; sum_n += dnp; sum_nr += dnr;
        }
        dhz = dhz_new;
        gsh[gslot[0]] = drp; gsh[gslot[1]] = dzp; gsh[gslot[2]] = dnr;
        __syncthreads();                                          // gsh complete; everyone has consumed dhs

        float acc[4][GP_BW];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int bb = 0; bb < GP_BW; ++bb) acc[j][bb] = 0.0f;
        const float* gpart = gsh + rp * GB4_RPS;
        auto fma16 = [&](const gp_f4& w0, const gp_f4& w1, const gp_f4& w2, const gp_f4& w3, int i) {
#pragma unroll
            for (int bb = 0; bb < GP_BW; ++bb) {
                gp_f4 g4;
                if (KNOCK & 4) g4 = w0;
                else g4 = *reinterpret_cast<const gp_f4*>(gpart + bb * GB4_GS + 4 * i);
                const gp_f4* ws[4] = {&w0, &w1, &w2, &w3};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j][bb] = fmaf(ws[j]->x, g4.x, acc[j][bb]); acc[j][bb] = fmaf(ws[j]->y, g4.y, acc[j][bb]);
                    acc[j][bb] = fmaf(ws[j]->z, g4.z, acc[j][bb]); acc[j][bb] = fmaf(ws[j]->w, g4.w, acc[j][bb]);
                }
            }
        };
        if (!(KNOCK & 2)) {
#pragma unroll
            for (int i = 0; i < GB4_REGJ; ++i) fma16(wa[0][i], wa[1][i], wb[0][i], wb[1][i], i);
            asm volatile("" ::: "memory");                        // keep the LDS-resident weights in LDS
#pragma unroll
            for (int i = GB4_REGJ; i < GB4_G; ++i)
                fma16(wa[0][i], wa[1][i], wl4[(size_t)(i - GB4_REGJ) * GP_THREADS + tid],
                      wl4[(size_t)((GB4_G - GB4_REGJ) + i - GB4_REGJ) * GP_THREADS + tid], i);
        }
        // this lane's two values: k index 2 fj, 2 fj + 1 of utterance fb, summed over the eight row parts (lanes 8 m .. 8 m + 7)
        float v0 = 0.0f, v1 = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int bb = 0; bb < GP_BW; ++bb) {
                float v = acc[j][bb];
                v += gp_quad_xor1(v); v += gp_quad_xor2(v); v += gp_half_mirror(v);
                if ((j & 1) == 0) v0 = (fb == bb && fj == (j >> 1)) ? v : v0;
                else v1 = (fb == bb && fj == (j >> 1)) ? v : v1;
            }
        const unsigned tagv = (epoch << 16) | (unsigned)(it + 1);
        if (it + 1 < S) {
            // [parity][writer half][utterance][unit of the READER's half] granules {value, tag}; the lane's two are neighbours
            unsigned long long* gmine = xg + ((size_t)(it & 1) * 2 + half) * GP_BW * GP_UH + fb * GP_UH + kl;
            const unsigned long long* gpeer = xg + ((size_t)(it & 1) * 2 + (half ^ 1)) * GP_BW * GP_UH + fb * GP_UH + kl;
            if (!own_wave) {                                      // the peer's units: publish the partials, nothing to wait for
                if (!(KNOCK & 1)) {
                    const gb4_u32x4 g2 = {__float_as_uint(v0), tagv, __float_as_uint(v1), tagv};
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(gmine), "v"(g2) : "memory");
                }
            } else {                                              // own units: complete them with the peer's partials
                float p0 = 0.0f, p1 = 0.0f;
                if (!(KNOCK & 1)) {
                    unsigned spins = 0;
                    for (;;) {
                        gb4_u32x4 g2;
                        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g2) : "v"(gpeer) : "memory");
                        p0 = __uint_as_float(g2.x); p1 = __uint_as_float(g2.z);
                        if (g2.y == tagv && g2.w == tagv) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > GP_SPIN_LIMIT) { __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    }
                }
                dhs[fb * GP_UH + kl] = v0 + p0;                   // (dhs was last read before the barrier above)
                dhs[fb * GP_UH + kl + 1] = v1 + p1;
            }
        }
        __syncthreads();                                          // dhs of the next step complete
    }
    if (bvalid) {
        float* bi = bsum_i + (size_t)(b0 + bq) * 1536 + dir * 768;
        float* bh = bsum_h + (size_t)(b0 + bq) * 1536 + dir * 768;
        bi[u] = sum_r; bi[256 + u] = sum_z; bi[512 + u] = sum_n;
        bh[u] = sum_r; bh[256 + u] = sum_z; bh[512 + u] = sum_nr;
    }
}
