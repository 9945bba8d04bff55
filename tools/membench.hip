// Store/stream bandwidth ceilings on MI355X for the expand kernel's access pattern, and known-byte-count
// read kernels that calibrate rocprofv3's FETCH_SIZE for the access patterns of the product's kernels
// (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern").
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench && tools/membench
//   calibration: rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -o pmc -- tools/membench calib
//                python tools/pmc_calibrate.py DIR   (-> profiles/pmc_calibration.json)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct T3 { uint32_t a, b, c; };

template <int MODE> __global__ __launch_bounds__(256) void k_store(T3* __restrict__ out, const uint2* __restrict__ src, uint64_t n, uint32_t srcmask) {
    const uint64_t base = uint64_t(blockIdx.x) * 2048;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint64_t p = base + j * 256 + threadIdx.x;
        if (p < n) {
            T3 t;
            if (MODE == 0 || MODE == 2) { t.a = uint32_t(p >> 11); t.b = uint32_t(p); t.c = 1; }
            else { const uint2 s = src[p & srcmask]; t.a = uint32_t(p >> 11); t.b = s.x; t.c = s.y; }
            if (MODE == 2 || MODE == 3) {
                __builtin_nontemporal_store(t.a, &out[p].a); __builtin_nontemporal_store(t.b, &out[p].b); __builtin_nontemporal_store(t.c, &out[p].c);
            } else out[p] = t;
        }
    }
}
__global__ __launch_bounds__(256) void k_store16(uint4* __restrict__ out, uint64_t n16) {
    const uint64_t base = uint64_t(blockIdx.x) * 2048;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint64_t p = base + j * 256 + threadIdx.x;
        if (p < n16) out[p] = make_uint4(uint32_t(p), 1, 2, 3);
    }
}
__global__ __launch_bounds__(256) void k_copy16(uint4* __restrict__ out, const uint4* __restrict__ in, uint64_t n16) {
    const uint64_t base = uint64_t(blockIdx.x) * 2048;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint64_t p = base + j * 256 + threadIdx.x;
        if (p < n16) out[p] = in[p];
    }
}

// ---- FETCH_SIZE calibration: every kernel reads a KNOWN number of bytes exactly once from a buffer far larger
// than L2 + Infinity Cache (or, for the *_hot variant, far smaller) ------------------------------------------
template <class V> __global__ __launch_bounds__(256) void cal_stream(const V* __restrict__ in, uint64_t n, uint32_t* __restrict__ sink) {
    const uint64_t base = uint64_t(blockIdx.x) * 2048;
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint64_t p = base + j * 256 + threadIdx.x;
        if (p < n) { const V v = in[p]; acc ^= v.x ^ v.y; }
    }
    if (acc == 0x12345u) sink[0] = acc;     // never true: keeps the loads alive
}
// walk_kernel's pattern: one 32-byte record (two 16-byte halves) per lane at a pseudo-random slot of a table
// of `mask + 1` records
__global__ __launch_bounds__(256) void cal_gather32(const uint4* __restrict__ tab, uint64_t mask, uint64_t n, uint32_t* __restrict__ sink) {
    const uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    uint64_t h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    const uint4* r = tab + 2 * (h & mask);
    const uint4 a = r[0], b = r[1];
    if ((a.x ^ b.y) == 0x12345u) sink[0] = a.x;
}
// expand_kernel's pattern: 8 bytes per lane, consecutive lanes read consecutive entries of a run that starts at a
// pseudo-random place of a pool; run length `run` entries (one wave = 64 consecutive entries when run >= 64)
__global__ __launch_bounds__(256) void cal_runs8(const uint2* __restrict__ pool, uint64_t pool_mask, uint32_t run, uint64_t n, uint32_t* __restrict__ sink) {
    const uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    uint64_t h = (i / run) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    const uint2 v = pool[((h & pool_mask) & ~uint64_t(run - 1)) + i % run];
    if ((v.x ^ v.y) == 0x12345u) sink[0] = v.x;
}

static int calib() {
    const uint64_t bytes = 16ull << 30;                       // 16 GiB: 64x the Infinity Cache
    void* buf; uint32_t* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, bytes));
    CK(hipDeviceSynchronize());
    const uint64_t n16 = bytes / 16, n8 = bytes / 8;
    // name, algorithmic bytes read (printed for tools/pmc_calibrate.py)
    cal_stream<uint4><<<uint32_t((n16 + 2047) / 2048), 256>>>((const uint4*)buf, n16, sink);
    printf("CAL cal_stream<uint4> %llu\n", (unsigned long long)bytes);
    cal_stream<uint2><<<uint32_t((n8 + 2047) / 2048), 256>>>((const uint2*)buf, n8, sink);
    printf("CAL cal_stream<uint2> %llu\n", (unsigned long long)bytes);
    const uint64_t recs = bytes / 32, ng = 1ull << 28;       // 2^28 gathers of 32 B from 2^29 records
    cal_gather32<<<uint32_t(ng / 256), 256>>>((const uint4*)buf, recs - 1, ng, sink);
    printf("CAL cal_gather32 %llu\n", (unsigned long long)(ng * 32));
    const uint64_t nr = 1ull << 30;
    for (uint32_t run : {16u, 64u, 1024u}) {
        cal_runs8<<<uint32_t(nr / 256), 256>>>((const uint2*)buf, n8 - 1, run, nr, sink);
        printf("CAL cal_runs8/%u %llu\n", run, (unsigned long long)(nr * 8));
    }
    // hot pool: 64 MiB of runs re-read 128 times (fits the Infinity Cache, not L2)
    cal_runs8<<<uint32_t(nr / 256), 256>>>((const uint2*)buf, (64ull << 20) / 8 - 1, 1024, nr, sink);
    printf("CAL cal_runs8/hot64MiB %llu\n", (unsigned long long)(nr * 8));
    CK(hipDeviceSynchronize());
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'c') return calib();
    const uint64_t n = 1ull << 28;                 // tuples (3 GiB)
    T3* out; uint2* src; uint4* in16;
    CK(hipMalloc(&out, n * 12)); CK(hipMalloc(&src, (1u << 20) * 8)); CK(hipMalloc(&in16, n * 12));
    CK(hipMemset(src, 1, (1u << 20) * 8)); CK(hipMemset(in16, 1, n * 12));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const uint32_t grid = uint32_t((n + 2047) / 2048);
    auto time = [&](const char* name, auto launch, double bytes) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(a); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("%-44s %8.3f ms  %8.1f GB/s\n", name, ms, bytes / ms / 1e6);
    };
    time("store dwordx3 (12 B/lane)", [&] { k_store<0><<<grid, 256>>>(out, src, n, 0xFFFFF); }, n * 12.0);
    time("L2-read 8 B + store dwordx3", [&] { k_store<1><<<grid, 256>>>(out, src, n, 0xFFFFF); }, n * 12.0);
    time("store 3x dword nontemporal", [&] { k_store<2><<<grid, 256>>>(out, src, n, 0xFFFFF); }, n * 12.0);
    time("L2-read 8 B + 3x dword nontemporal", [&] { k_store<3><<<grid, 256>>>(out, src, n, 0xFFFFF); }, n * 12.0);
    const uint64_t n16 = n * 12 / 16; const uint32_t g16 = uint32_t((n16 + 2047) / 2048);
    time("store dwordx4 (16 B/lane)", [&] { k_store16<<<g16, 256>>>((uint4*)out, n16); }, n * 12.0);
    time("copy dwordx4 (read+write bytes)", [&] { k_copy16<<<g16, 256>>>((uint4*)out, in16, n16); }, n * 24.0);
    return 0;
}
