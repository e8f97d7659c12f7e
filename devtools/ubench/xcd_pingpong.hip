// Hand-off latency between two workgroups, by the scope of the accesses -- is an exchange through the XCD's own L2
// (partners placed on the same XCD, sc0 = L1-bypassing loads, plain write-through stores) cheaper than the agent-scope
// relaxed atomics the GRU recurrence kernels use (which go out to the fabric)?
//   * 64 workgroups; partners are (i, i + stride) with stride 8 (same XCD if workgroups go round robin over 8 XCDs) or
//     stride 1 (neighbouring XCDs);
//   * each workgroup reports its XCC_ID (s_getreg HW_REG_XCC_ID) so the placement assumption is checked, not assumed;
//   * a ping-pong of N tagged 8-byte granules ("the data is the flag"), timed with s_memrealtime (100 MHz).
// Every spin loop has a limit: an incoherent path shows up as "TIMEOUT", not as a hang.
//   build: hipcc -O3 --offload-arch=gfx950 xcd_pingpong.hip -o xcd_pingpong
// Measured (MI355X, idle chip, round 2): workgroup i runs on XCD i % 8 (all 64 ids as expected); agent-scope round trip
// 0.95 us between neighbouring XCDs, 0.82 us on one XCD -- placement buys 14 %, so the recurrences' 2.1 us hand-off is
// load (each workgroup write-through-stores 8 KB and polls 24 KB per step), not distance.  The sc0 path never sees the
// partner's store on gfx950 (TIMEOUT on the same XCD too): no cheaper-than-agent scope is available for this.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__device__ __forceinline__ void put(unsigned long long* p, unsigned long long v) {
    if (MODE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
}
template <int MODE>
__device__ __forceinline__ unsigned long long get(const unsigned long long* p) {
    if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int MODE>
__global__ void pingpong(unsigned long long* box, int stride, int rounds, unsigned* xcc, unsigned long long* ticks, unsigned* timeouts) {
    const int id = blockIdx.x;
    const int pair_lo = (id / (2 * stride)) * (2 * stride) + id % stride;      // lower member of this workgroup's pair
    const bool lo = id == pair_lo;
    if (threadIdx.x == 0) xcc[id] = __builtin_amdgcn_s_getreg(20 | (3 << 11));  // HW_REG_XCC_ID, bits 3:0
    if (threadIdx.x != 0) return;
    unsigned long long* mine = box + (size_t)id * 16;                          // 128-byte apart
    const unsigned long long* theirs = box + (size_t)(lo ? id + stride : id - stride) * 16;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = wall_clock64();
    unsigned to = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (lo) put<MODE>(mine, (unsigned long long)r);
        unsigned spins = 0;
        while (get<MODE>(theirs) != (unsigned long long)r) {
            if (++spins > (1u << 20)) { to = 1; break; }
        }
        if (to) break;
        if (!lo) put<MODE>(mine, (unsigned long long)r);
    }
    (void)t0;
    ticks[id] = wall_clock64() - r0;
    timeouts[id] = to;
}

template <int MODE>
static int run(const char* name, int stride, int nwg, int rounds) {
    unsigned long long *box, *ticks;
    unsigned *xcc, *to;
    CK(hipMalloc(&box, (size_t)nwg * 128)); CK(hipMemset(box, 0, (size_t)nwg * 128));
    CK(hipMalloc(&ticks, nwg * 8)); CK(hipMalloc(&xcc, nwg * 4)); CK(hipMalloc(&to, nwg * 4));
    hipLaunchKernelGGL(pingpong<MODE>, dim3(nwg), dim3(64), 0, 0, box, stride, rounds, xcc, ticks, to);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> ht(nwg);
    std::vector<unsigned> hx(nwg), hto(nwg);
    CK(hipMemcpy(ht.data(), ticks, nwg * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hx.data(), xcc, nwg * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hto.data(), to, nwg * 4, hipMemcpyDeviceToHost));
    int same = 0, pairs = 0, touts = 0;
    double sum = 0, mx = 0;
    for (int id = 0; id < nwg; ++id) {
        const int lo = (id / (2 * stride)) * (2 * stride) + id % stride;
        if (id != lo) continue;
        ++pairs;
        same += hx[id] == hx[id + stride];
        touts += hto[id] | hto[id + stride];
        const double us = ht[id] / 100.0 / rounds;               // 100 MHz wall clock -> us per round trip
        sum += us; mx = us > mx ? us : mx;
    }
    printf("%-28s stride %d: %d/%d pairs on one XCD, round trip %.2f us avg / %.2f max (one-way hand-off = half)%s\n", name, stride, same, pairs,
           sum / pairs, mx, touts ? "   TIMEOUT" : "");
    printf("    XCC ids of workgroups 0..15:");
    for (int i = 0; i < 16 && i < nwg; ++i) printf(" %u", hx[i]);
    printf("\n");
    (void)hipFree(box); (void)hipFree(ticks); (void)hipFree(xcc); (void)hipFree(to);
    return 0;
}

int main() {
    const int nwg = 64, rounds = 2000;
    if (run<0>("agent-scope relaxed atomics", 1, nwg, rounds)) return 1;
    if (run<0>("agent-scope relaxed atomics", 8, nwg, rounds)) return 1;
    if (run<1>("sc0 load / sc0 store (L2)", 8, nwg, rounds)) return 1;
    if (run<1>("sc0 load / sc0 store (L2)", 1, nwg, rounds)) return 1;   // different XCDs: expected to time out or crawl
    return 0;
}
