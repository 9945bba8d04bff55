// What an LDS atomic costs on MI355X (r5): the v5 dedup's topic pass issues ~8 returning 64-bit LDS atomics per 64 candidates and two
// variants that removed its global round trips did not get faster — is it bound by the LDS atomic unit?  Per mode: 1024 blocks x 512
// threads (4 per CU, the topic pass's shape), every wave issues ITER dependent (returning) or independent (fire-and-forget) operations on
// pseudo-random slots of a 32 KiB table; reported: ns per wave-instruction per CU (all 32 waves of a CU issuing) and the same in cycles
// at the measured clock.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_bench.hip -o tools/lds_atomic_bench && tools/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kSlots = 4096, kIter = 2048;
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// MODE 0: atomicCAS u64 returning   1: atomicMin u64 returning   2: atomicMin u64 no return   3: atomicOr u32 returning
//      4: atomicOr u32 no return    5: ds_read_b64               6: ds_write_b64              7: atomicCAS u32 returning
//      8: atomicAdd u32 returning   9: ds_read_b32
template <int MODE> __global__ __launch_bounds__(512) void k(uint32_t active_lanes, uint64_t* sink) {
    __shared__ unsigned long long tab[kSlots];
    uint32_t* tab32 = reinterpret_cast<uint32_t*>(tab);
    for (uint32_t i = threadIdx.x; i < kSlots; i += 512) tab[i] = ~0ull;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    uint32_t x = mix(blockIdx.x * 512 + threadIdx.x + 1);
    unsigned long long acc = 0;
    if (lane < active_lanes) {
        for (int it = 0; it < kIter; ++it) {
            const uint32_t s = x & (kSlots - 1), s32 = x & (2 * kSlots - 1);
            unsigned long long r = 0;
            if (MODE == 0) r = atomicCAS(&tab[s], ~0ull, (unsigned long long)x << 32 | it);
            else if (MODE == 1) r = atomicMin(&tab[s], (unsigned long long)x << 32 | it);
            else if (MODE == 2) atomicMin(&tab[s], (unsigned long long)x << 32 | it);
            else if (MODE == 3) r = atomicOr(&tab32[s32], 1u << (x >> 27));
            else if (MODE == 4) atomicOr(&tab32[s32], 1u << (x >> 27));
            else if (MODE == 5) r = tab[s];
            else if (MODE == 6) tab[s] = x;
            else if (MODE == 7) r = atomicCAS(&tab32[s32], ~0u, x);
            else if (MODE == 8) r = atomicAdd(&tab32[s32], 1u);
            else if (MODE == 9) r = tab32[s32];
            acc += r;
            x = mix(x + uint32_t(r & 1));        // returning modes: the next address depends on the result (a probing loop's chain)
        }
    }
    if (acc == 0x123456789ull) sink[0] = acc + tab[threadIdx.x];
}

template <int MODE> int run(const char* name, uint32_t lanes, uint64_t* sink) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<MODE><<<1024, 512>>>(lanes, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) k<MODE><<<1024, 512>>>(lanes, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= 5;
    // per CU: 4 blocks x 8 waves x kIter wave-instructions
    const double per_cu = 4.0 * 8 * kIter;
    printf("%-28s lanes %2u  %8.3f ms   %7.2f ns per wave-instruction per CU   (%6.1f cycles at 2.4 GHz)\n", name, lanes, ms, ms * 1e6 / per_cu, ms * 1e6 / per_cu * 2.4);
    return 0;
}

int main() {
    uint64_t* sink;
    CK(hipMalloc(&sink, 64));
    for (uint32_t lanes : {64u, 16u, 4u, 1u}) {
        run<0>("atomicCAS u64 rtn", lanes, sink);
        run<1>("atomicMin u64 rtn", lanes, sink);
        run<2>("atomicMin u64 no rtn", lanes, sink);
        run<7>("atomicCAS u32 rtn", lanes, sink);
        run<3>("atomicOr u32 rtn", lanes, sink);
        run<4>("atomicOr u32 no rtn", lanes, sink);
        run<8>("atomicAdd u32 rtn", lanes, sink);
        run<5>("ds_read_b64", lanes, sink);
        run<9>("ds_read_b32", lanes, sink);
        run<6>("ds_write_b64", lanes, sink);
    }
    return 0;
}
