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

