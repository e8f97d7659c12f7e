// Paired-workgroup GRU recurrence: W_hh never leaves the chip during the sequence.
//
// W_hh of one direction is 768 x 256 fp32 = 786 KB -- more than one CU can hold (160 KB LDS +
// 512 KB registers), which is why gru_recurrence_kernel re-streams ~2/3 of it from L2 every step
// and is bound by L2 bandwidth (PMC: 70 % of wave time in s_waitcnt).  Here TWO workgroups share
// one group of 4 utterances: workgroup `half` owns hidden units [128*half, 128*half+128), i.e. 384
// gate rows = 393 KB, of which 10/16 live in registers and 6/16 in LDS for all S steps.  After each
// step the two halves exchange their 128 x 4 new h values (2 KB) through global memory:
//   every value is one naturally aligned 8-byte {tag = step+1, value} granule written by ONE agent-scope
//   relaxed atomic store (write-through, sc1); the consumer re-reads its granule with agent-scope
//   relaxed loads until the tag matches -- recipe R2 of cdna_hip_programming.md G16 ("the data IS the
//   flag": no separate flag, no fence; measured 1 hop ~1 us vs ~5 us for store+flag+poll+load).
// Granules are double-buffered by step parity and zeroed by a memset before every launch, the pair
// sits in adjacent blockIdx (dispatched together), spins are bounded and report a timeout word.
// Thread layout: lane = k-part (4 parts of 64 k) + 4 * (unit & 15), 8 waves x 16 units = 128 units;
// the 4 k-parts of a unit are summed with two __shfl_xor steps (no LDS partial sums), and lane kp
// then finishes utterance kp of its unit.
#pragma once
#include "model_kernels.h"

constexpr int GP_BW = 4;                 // utterances per pair
constexpr int GP_UH = 128;               // hidden units per workgroup
constexpr int GP_THREADS = 512;          // 128 units x 4 k-parts: 2 waves per SIMD, up to 256 VGPRs each
constexpr int GP_KP = 4;                 // k-parts (64 k each)
constexpr int GP_K4 = 64 / GP_KP;        // float4 groups per gate per thread (16)
constexpr int GP_REG4 = 10;              // ... of which kept in registers (120 VGPRs)
constexpr int GP_LDS4 = GP_K4 - GP_REG4; // ... kept in LDS (6 x 24 KB)
constexpr int GP_HP = 64 + 4;             // padded length of one 64-wide k-part of h: the 4 k-parts a wave reads at once
constexpr int GP_HB = GP_KP * GP_HP;      // land in different banks (unpadded they are 4-way conflicting)
constexpr size_t GP_LDS_BYTES = (size_t)(GP_LDS4 * 3 * GP_THREADS * 4 + GP_BW * GP_HB) * 4;
__device__ __forceinline__ int gp_hidx(int b, int k) { return b * GP_HB + (k >> 6) * GP_HP + (k & 63); }
constexpr unsigned GP_SPIN_LIMIT = 1u << 22;

typedef __attribute__((address_space(1))) unsigned int gp_gu32;

typedef float gp_f2 __attribute__((ext_vector_type(2)));
typedef float gp_f4 __attribute__((ext_vector_type(4)));

// acc2[b] += (w.x, w.y) * (h.x, h.y) + (w.z, w.w) * (h.z, h.w): two v_pk_fma_f32 per 4 MACs.  At 2 waves per
// SIMD this kernel is bound by VALU ISSUE (PMC: every instruction costs its wave 4 cycles, the two waves
// barely overlap), so halving the instruction count is what counts; the even/odd partial sums are added once
// per step.
template <int NB>
__device__ __forceinline__ void gp_pkfma4(gp_f2 (&acc)[NB], const gp_f4 w, const gp_f4 (&h4)[NB]) {
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
        acc[bb] = __builtin_elementwise_fma(w.xy, h4[bb].xy, acc[bb]);
        acc[bb] = __builtin_elementwise_fma(w.zw, h4[bb].zw, acc[bb]);
    }
}

__device__ __forceinline__ float gp_quad_xor1(float v) {     // lane ^ 1 within each quad
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float gp_quad_xor2(float v) {     // lane ^ 2 within each quad
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ void gp_store_sc1(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // global_store_dword ... sc1
}
__device__ __forceinline__ float gp_load_sc1(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_load_dword ... sc1
}

// xbuf  [npairs*2 dirs][2 parity][2 halves][GP_BW][128] 8-byte {tag, value} granules, zeroed before every launch
// status[0] is set to 1 if a spin times out (results are then invalid; never happens when all pairs are resident)
template <bool SAVE>
__global__ __launch_bounds__(GP_THREADS) void gru_pair_kernel(
    const float* __restrict__ gi, const float* __restrict__ whh0, const float* __restrict__ whh1,
    const float* __restrict__ bhh0, const float* __restrict__ bhh1, float* __restrict__ y, int B, int S,
    float* __restrict__ gates, float* xbuf, unsigned int* flags, unsigned int* status, int dbg_nowait = 0) {
    extern __shared__ __attribute__((aligned(16))) float plds[];
    gp_f4* wl4 = reinterpret_cast<gp_f4*>(plds);                       // [GP_LDS4][3][threads] float4
    float* hs = plds + GP_LDS4 * 3 * GP_THREADS * 4;                           // h[b][256]
    const int dir = blockIdx.y, pair = blockIdx.x >> 1, half = blockIdx.x & 1;
    const int b0 = pair * GP_BW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int kp = lane & (GP_KP - 1), ul = (wv << 4) | (lane >> 2);      // unit within this half
    const int u = half * GP_UH + ul;                                     // hidden unit (0..255)
    const float* __restrict__ whh = dir ? whh1 : whh0;                   // original [768][256] layout
    const float* __restrict__ bhh = dir ? bhh1 : bhh0;
    const int pd = pair * 2 + dir;
    unsigned long long* xg = reinterpret_cast<unsigned long long*>(xbuf) + (size_t)pd * 2 * 2 * GP_BW * GP_UH;   // granules

    // this thread's weights: gate rows g*256+u, k in [64*kp, 64*kp+64): resident for the whole sequence
    gp_f4 wr[GP_REG4][3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const gp_f4* src = reinterpret_cast<const gp_f4*>(whh + (size_t)(g * 256 + u) * 256 + kp * (GP_K4 * 4));
#pragma unroll
        for (int i = 0; i < GP_REG4; ++i) wr[i][g] = src[i];
#pragma unroll
        for (int i = 0; i < GP_LDS4; ++i) wl4[(i * 3 + g) * GP_THREADS + tid] = src[GP_REG4 + i];
    }
    const float bh_r = bhh[u], bh_z = bhh[256 + u], bh_n = bhh[512 + u];
    for (int i = tid; i < GP_BW * GP_HB; i += GP_THREADS) hs[i] = 0.0f;
    const int bme = kp;                                      // lane kp finishes utterance kp of unit u
    const bool finisher = true;
    const bool bvalid = finisher && (b0 + bme) < B;
    float hprev = 0.0f;
    __syncthreads();

    // gate pre-activations are fetched one step ahead: their HBM/L2 latency hides behind a whole step
    float gr_n = 0.f, gz_n = 0.f, gn_n = 0.f;
    if (bvalid) {
        const float* g = gi + ((size_t)(b0 + bme) * S + (dir ? S - 1 : 0)) * 1536 + dir * 768;
        gr_n = g[u]; gz_n = g[256 + u]; gn_n = g[512 + u];
    }
    for (int step = 0; step < S; ++step) {
        const int t = dir ? (S - 1 - step) : step;
        const float gr = gr_n, gz = gz_n, gn = gn_n;
        if (bvalid && step + 1 < S) {
            const int tn = dir ? (S - 2 - step) : step + 1;
            const float* g = gi + ((size_t)(b0 + bme) * S + tn) * 1536 + dir * 768;
            gr_n = g[u]; gz_n = g[256 + u]; gn_n = g[512 + u];
        }
        gp_f2 acc2[3][GP_BW];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int bb = 0; bb < GP_BW; ++bb) acc2[g][bb] = (gp_f2)(0.0f, 0.0f);
#pragma unroll
        for (int i = 0; i < GP_REG4; ++i) {                      // register-resident weights
            gp_f4 h4[GP_BW];
#pragma unroll
            for (int bb = 0; bb < GP_BW; ++bb) h4[bb] = *reinterpret_cast<const gp_f4*>(hs + bb * GP_HB + kp * GP_HP + i * 4);
            gp_pkfma4(acc2[0], wr[i][0], h4); gp_pkfma4(acc2[1], wr[i][1], h4); gp_pkfma4(acc2[2], wr[i][2], h4);
        }
        // keep the LDS-resident weights IN LDS (do not let the compiler hoist these loop-invariant loads)
        asm volatile("" ::: "memory");
#pragma unroll 2
        for (int i = 0; i < GP_LDS4; ++i) {                      // LDS-resident weights
            gp_f4 h4[GP_BW];
#pragma unroll
            for (int bb = 0; bb < GP_BW; ++bb)
                h4[bb] = *reinterpret_cast<const gp_f4*>(hs + bb * GP_HB + kp * GP_HP + (GP_REG4 + i) * 4);
#pragma unroll
            for (int g = 0; g < 3; ++g) gp_pkfma4(acc2[g], wl4[(i * 3 + g) * GP_THREADS + tid], h4);
        }
        // sum even/odd partial sums, then the 4 k-parts of each unit (lanes differing in bits 0..1)
        float acc[3][GP_BW];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int bb = 0; bb < GP_BW; ++bb) {
                float v = acc2[g][bb].x + acc2[g][bb].y;
                v += gp_quad_xor1(v);          // DPP quad_perm: no LDS crossbar trip
                v += gp_quad_xor2(v);
                acc[g][bb] = v;
            }
        float hr = bh_r, hz = bh_z, hn = bh_n;
#pragma unroll
        for (int bb = 0; bb < GP_BW; ++bb)
            if (bb == bme) { hr += acc[0][bb]; hz += acc[1][bb]; hn += acc[2][bb]; }
        const float r = sigmoidf_(gr + hr);
        const float zg = sigmoidf_(gz + hz);
        const float nn = tanhf(gn + r * hn);
        const float hnew = (1.0f - zg) * nn + zg * hprev;
        // exchange (R2 of the guide's hand-off recipe: the data IS the flag): every value travels as one
        // naturally aligned 8-byte {tag = step+1, value} granule written by ONE write-through store; the
        // consumer re-reads its granule until the tag matches.  One hop instead of store+flag+poll+load.
        unsigned long long* gslot = xg + ((size_t)(step & 1) * 2 + half) * GP_BW * GP_UH;
        const unsigned long long* gpeer = xg + ((size_t)(step & 1) * 2 + (half ^ 1)) * GP_BW * GP_UH;
        hprev = hnew;
        __hip_atomic_store(gslot + bme * GP_UH + ul, ((unsigned long long)(unsigned)(step + 1) << 32) | __float_as_uint(hnew),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bvalid) {
            y[((size_t)(b0 + bme) * S + t) * 512 + dir * 256 + u] = hnew;
            if (SAVE) {
                float* gs = gates + (((size_t)(b0 + bme) * S + t) * 2 + dir) * 1024;
                gs[u] = r; gs[256 + u] = zg; gs[512 + u] = nn; gs[768 + u] = hn;
            }
        }
        __syncthreads();                                                  // every wave is done reading hs
        hs[gp_hidx(bme, u)] = hnew;                                       // own half of the new h
        unsigned long long pv;
        unsigned spins = 0;
        while (((pv = __hip_atomic_load(gpeer + bme * GP_UH + ul, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) !=
               (unsigned long long)(unsigned)(step + 1)) {
            if (dbg_nowait) break;                                        // timing experiment only (wrong results)
            __builtin_amdgcn_s_sleep(1);
            if (++spins > GP_SPIN_LIMIT) { __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        hs[gp_hidx(bme, (half ^ 1) * GP_UH + ul)] = __uint_as_float((unsigned)pv);
        __syncthreads();
    }
}
