// Store/stream bandwidth ceilings on MI355X for the expand kernel's access pattern.
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o /tmp/membench && /tmp/membench
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

int main() {
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
