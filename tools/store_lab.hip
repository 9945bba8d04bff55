// Store-stream lab (r5t): which block geometry / position mapping gets a 12-byte-tuple store stream closest to the chip's write ceiling?
// The IDS24 expansion writes 6.2 TB/s when it runs alone (profiles/r05k_*: 3 B x 2^30 hits in 0.519 ms) with the SAME store instruction
// shape as the tuple expansion (one nontemporal dwordx3 per lane, consecutive lanes on consecutive 12-byte groups), which writes 5.5-5.7.
// Every kernel here reads one 8-byte entry per tuple from an L2-resident run (like the expansion at config 3) and writes N 12-byte tuples:
//   strided   block = THREADS x PER tuples, lane's j-th tuple at j * THREADS + tid      (expand_kernel: 1024 x 2)
//   wavecont  the same tile, but a wave owns PER * 64 CONSECUTIVE tuples: its j-th store follows its (j-1)-th
//   hipcc --offload-arch=gfx950 -O3 tools/store_lab.hip -o tools/store_lab && tools/store_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct T3 { uint32_t a, b, c; };

__device__ uint4* g_recs;        // MAP 3: one 16-byte record per tile (tiles_kernel's TileRec): the block's first load, everything else depends on it
template <int THREADS, int PER, int MAP, bool NT>
__global__ __launch_bounds__(THREADS) void k_store(T3* __restrict__ out, const uint2* __restrict__ src, uint64_t n, uint32_t srcmask) {
    constexpr int TILE = THREADS * PER;
    // MAP 2: strided positions, and the blocks of one XCD (blockIdx % 8) own one CONTIGUOUS eighth of the output
    const uint32_t nb = gridDim.x, bid = MAP == 2 ? (blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const uint64_t base = uint64_t(bid) * TILE;
    uint32_t s0 = uint32_t(blockIdx.x * 2654435761u) & srcmask;          // the tile's run: somewhere in the hot pool
    if (MAP == 3) { const uint4 r = g_recs[blockIdx.x]; s0 = r.y & srcmask; }
    uint64_t pos[PER];
    uint2 e[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const uint32_t rel = (MAP != 1) ? uint32_t(j) * THREADS + threadIdx.x : (threadIdx.x >> 6) * (PER * 64) + uint32_t(j) * 64 + (threadIdx.x & 63);
        pos[j] = base + rel;
        e[j] = src[(s0 + rel) & srcmask];
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        if (pos[j] < n) {
            if (NT) {
                __builtin_nontemporal_store(uint32_t(blockIdx.x), &out[pos[j]].a);
                __builtin_nontemporal_store(e[j].x, &out[pos[j]].b);
                __builtin_nontemporal_store(e[j].y, &out[pos[j]].c);
            } else {
                T3 t; t.a = blockIdx.x; t.b = e[j].x; t.c = e[j].y;
                out[pos[j]] = t;
            }
        }
    }
}

template <int THREADS, int PER, int MAP, bool NT>
int run(const char* name, T3* out, const uint2* src, uint64_t n, hipEvent_t a, hipEvent_t b, uint32_t srcmask = (1u << 20) - 1) {
    const uint32_t grid = uint32_t((n + THREADS * PER - 1) / (THREADS * PER));
    k_store<THREADS, PER, MAP, NT><<<grid, THREADS>>>(out, src, n, srcmask);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 4; ++i) k_store<THREADS, PER, MAP, NT><<<grid, THREADS>>>(out, src, n, srcmask);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= 4;
    printf("%-34s tile %5d  pool %4u KiB  %8.3f ms  %8.1f GB/s\n", name, THREADS * PER, (srcmask + 1) / 128, ms, n * 12.0 / ms / 1e6);
    return 0;
}

int main() {
    const uint64_t n = 1ull << 30;                 // tuples: 12 GiB, one headline window
    T3* out; uint2* src;
    CK(hipMalloc(&out, n * 12)); CK(hipMalloc(&src, (1u << 20) * 8));
    CK(hipMemset(src, 1, (1u << 20) * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
#define R(T, P, M, N) if (run<T, P, M, N>(#T " x " #P " map" #M " nt=" #N, out, src, n, a, b)) return 1
    R(1024, 2, 0, true); R(1024, 2, 1, true); R(1024, 2, 0, false);
    R(512, 4, 0, true);  R(512, 4, 1, true);
    R(256, 8, 0, true);  R(256, 8, 1, true);
    R(256, 4, 0, true);  R(256, 4, 1, true);
    R(256, 2, 0, true);  R(512, 2, 0, true);  R(1024, 1, 0, true);
    R(128, 16, 0, true); R(128, 16, 1, true); R(128, 8, 1, true);
    R(64, 32, 1, true);  R(64, 16, 1, true);
    R(1024, 4, 0, true); R(1024, 4, 1, true); R(512, 8, 1, true);
    // XCD-contiguous block order; a small (certainly L2-resident) source pool; larger tiles
    R(1024, 2, 2, true); R(1024, 4, 2, true); R(256, 8, 2, true);
#define RS(T, P, M, N, MASK) if (run<T, P, M, N>(#T " x " #P " map" #M " nt=" #N, out, src, n, a, b, MASK)) return 1
    RS(1024, 2, 0, true, (1u << 16) - 1); RS(1024, 4, 0, true, (1u << 16) - 1); RS(256, 8, 0, true, (1u << 16) - 1); RS(256, 2, 0, true, (1u << 16) - 1);
    {   // records: 2^30 / 2048 = 524 288 tiles x 16 B = 8 MiB (a 2^30-hit window's), .y = the run's start
        uint4* recs; const uint32_t nt = uint32_t(n / 512);
        CK(hipMalloc(&recs, size_t(nt) * 16));
        uint4* h = new uint4[nt];
        for (uint32_t i = 0; i < nt; ++i) h[i] = make_uint4(i, i * 2654435761u, i, 0);
        CK(hipMemcpy(recs, h, size_t(nt) * 16, hipMemcpyHostToDevice));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_recs), &recs, sizeof(recs)));
        delete[] h;
    }
    RS(1024, 2, 3, true, (1u << 16) - 1); RS(1024, 2, 0, true, (1u << 16) - 1); RS(256, 8, 3, true, (1u << 16) - 1); RS(512, 4, 3, true, (1u << 16) - 1); RS(512, 4, 0, true, (1u << 16) - 1);
    RS(1024, 2, 3, true, (1u << 18) - 1); RS(1024, 2, 0, true, (1u << 18) - 1); RS(1024, 2, 0, true, (1u << 19) - 1);
    RS(1024, 8, 0, true, (1u << 16) - 1); RS(1024, 8, 0, true, (1u << 20) - 1); RS(1024, 16, 0, true, (1u << 20) - 1); RS(512, 16, 0, true, (1u << 20) - 1);
    return 0;
}
