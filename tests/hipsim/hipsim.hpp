// TEST INFRASTRUCTURE ONLY — runs the SOURCE of a HIP kernel on the host, one OS thread per GPU thread.
//
// tests/hipsim compiles selected kernel sources of rmqtt_amd/csrc (the .inc files kernels.hip includes) for x86 with the ROCm clang
// (vector types and builtins as in device code) behind the small set of shims below, and runs a launch block by block: 256 threads of a
// block are 256 OS threads, __syncthreads() is a pthread barrier, __shared__ is a function-local static, cross-lane reads of a wave
// (RGR_LANES_PUT / RGR_LANES_GET in the kernel source: ds_bpermute on the device) go through a per-wave mirror.  What this checks on a
// machine without a GPU: the index arithmetic, the control flow around partial / multi-run / straddling cases, the LDS protocol (a
// missing barrier is a data race ThreadSanitizer reports: tools/hipsim_sanitizers.sh).  It is NOT a CPU fallback: the product library neither
// contains nor links any of this, and nothing here is timed.
#pragma once
#include <pthread.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define RGR_HIPSIM 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

namespace hipsim {
struct Dim { unsigned x; };
constexpr int kWave = 64;
constexpr int kMaxWaves = 16;
struct Wave {
    pthread_barrier_t bar;
    uint32_t mirror[4][kWave];
};
struct BlockCtx {
    pthread_barrier_t bar;
    Wave waves[kMaxWaves];
};
inline BlockCtx* g_block = nullptr;
inline thread_local unsigned t_tid = 0;
inline void syncthreads() { pthread_barrier_wait(&g_block->bar); }
inline void wave_sync() { pthread_barrier_wait(&g_block->waves[t_tid / kWave].bar); }
// lane's value published to the wave: convergent (every lane of the wave), like the instruction it stands for
inline void lanes_put(int slot, uint32_t v) {
    Wave& w = g_block->waves[t_tid / kWave];
    w.mirror[slot][t_tid % kWave] = v;
    pthread_barrier_wait(&w.bar);
}
inline uint32_t lanes_get(int slot, uint32_t lane) { return g_block->waves[t_tid / kWave].mirror[slot][lane % kWave]; }

}  // namespace hipsim

inline thread_local hipsim::Dim threadIdx{0}, blockIdx{0};
#define __syncthreads() hipsim::syncthreads()

inline uint32_t min(uint32_t a, uint32_t b) { return a < b ? a : b; }
inline uint32_t max(uint32_t a, uint32_t b) { return a > b ? a : b; }

namespace hipsim {
template <class F> void run(unsigned nblocks, unsigned nthreads, F kernel) {
    BlockCtx ctx;
    pthread_barrier_init(&ctx.bar, nullptr, nthreads);
    const unsigned nw = (nthreads + kWave - 1) / kWave;
    for (unsigned w = 0; w < nw; ++w) {
        const unsigned lanes = (w + 1) * kWave <= nthreads ? kWave : nthreads - w * kWave;
        pthread_barrier_init(&ctx.waves[w].bar, nullptr, lanes);
        std::memset(ctx.waves[w].mirror, 0, sizeof(ctx.waves[w].mirror));
    }
    g_block = &ctx;
    std::vector<std::thread> th;
    th.reserve(nthreads);
    for (unsigned t = 0; t < nthreads; ++t)
        th.emplace_back([&, t] {
            t_tid = t;
            threadIdx.x = t;
            for (unsigned b = 0; b < nblocks; ++b) {
                blockIdx.x = b;
                kernel();
                pthread_barrier_wait(&ctx.bar);          // the next block reuses the statics that stand for LDS
            }
        });
    for (auto& t : th) t.join();
    for (unsigned w = 0; w < nw; ++w) pthread_barrier_destroy(&ctx.waves[w].bar);
    pthread_barrier_destroy(&ctx.bar);
    g_block = nullptr;
}
}  // namespace hipsim
