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
#include <time.h>

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
// A barrier that gives up: a cross-lane read executed by only some lanes of a wave (the hardware would hand the readers ZERO for
// every source lane that is switched off — ds_bpermute honours EXEC on both sides) shows up here as lanes that never arrive.  After
// kGiveUpMs the wait fails, the run is flagged (hipsim::diverged()) and every later barrier lets its callers through.
constexpr int kGiveUpMs = 4000;
inline bool g_diverged = false;
struct TimedBarrier {
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    pthread_cond_t cv = PTHREAD_COND_INITIALIZER;
    unsigned n = 0, waiting = 0, generation = 0;
    void init(unsigned count) { n = count; waiting = 0; generation = 0; }
    void wait() {
        pthread_mutex_lock(&mu);
        if (g_diverged) { pthread_mutex_unlock(&mu); return; }
        const unsigned gen = generation;
        if (++waiting == n) {
            waiting = 0; ++generation;
            pthread_cond_broadcast(&cv);
        } else {
            timespec ts;
            clock_gettime(CLOCK_REALTIME, &ts);
            ts.tv_sec += kGiveUpMs / 1000; ts.tv_nsec += long(kGiveUpMs % 1000) * 1000000L;
            if (ts.tv_nsec >= 1000000000L) { ts.tv_sec++; ts.tv_nsec -= 1000000000L; }
            while (gen == generation && !g_diverged)
                if (pthread_cond_timedwait(&cv, &mu, &ts) != 0 && gen == generation) { g_diverged = true; pthread_cond_broadcast(&cv); }
        }
        pthread_mutex_unlock(&mu);
    }
};
struct Wave {
    TimedBarrier bar;
    uint32_t mirror[4][kWave];
};
struct BlockCtx {
    TimedBarrier bar;
    Wave waves[kMaxWaves];
};
inline BlockCtx* g_block = nullptr;
inline thread_local unsigned t_tid = 0;
inline bool diverged() { return g_diverged; }
inline void syncthreads() { g_block->bar.wait(); }
inline void wave_sync() { g_block->waves[t_tid / kWave].bar.wait(); }
// lane's value published to the wave: convergent (every lane of the wave), like the instruction it stands for
inline void lanes_put(int slot, uint32_t v) {
    Wave& w = g_block->waves[t_tid / kWave];
    w.mirror[slot][t_tid % kWave] = v;
    w.bar.wait();
}
// lane `lane`'s published value.  EVERY lane of the wave must execute the read (the barrier checks it): see TimedBarrier
inline uint32_t lanes_get(int slot, uint32_t lane) {
    Wave& w = g_block->waves[t_tid / kWave];
    w.bar.wait();
    return w.mirror[slot][lane % kWave];
}
}  // namespace hipsim

inline thread_local hipsim::Dim threadIdx{0}, blockIdx{0};
inline hipsim::Dim gridDim{1};          // (set by hipsim::run)

// LDS / global atomics of the kernels (every "GPU thread" is an OS thread: real atomics)
template <class T> inline T atomicCAS(T* p, T expected, T desired) {
    __atomic_compare_exchange_n(p, &expected, desired, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
    return expected;
}
template <class T> inline T atomicMin(T* p, T v) {
    T cur = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (v < cur && !__atomic_compare_exchange_n(p, &cur, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return cur;
}
template <class T> inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }

// wave-level votes and shuffles (convergent: every lane of the wave, like the instructions)
inline unsigned long long __ballot(int pred) {
    hipsim::lanes_put(2, pred ? 1u : 0u);
    unsigned long long m = 0;
    for (unsigned l = 0; l < hipsim::kWave; ++l) m |= static_cast<unsigned long long>(hipsim::g_block->waves[hipsim::t_tid / hipsim::kWave].mirror[2][l] & 1u) << l;
    hipsim::wave_sync();
    return m;
}
inline uint32_t __shfl(uint32_t v, int src_lane, int /*width*/) {
    hipsim::lanes_put(3, v);
    const uint32_t r = hipsim::g_block->waves[hipsim::t_tid / hipsim::kWave].mirror[3][unsigned(src_lane) % hipsim::kWave];
    hipsim::wave_sync();
    return r;
}
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
#define __syncthreads() hipsim::syncthreads()

inline uint32_t min(uint32_t a, uint32_t b) { return a < b ? a : b; }
inline uint32_t max(uint32_t a, uint32_t b) { return a > b ? a : b; }

namespace hipsim {
template <class F> bool run(unsigned nblocks, unsigned nthreads, F kernel) {
    BlockCtx* ctx = new BlockCtx();
    ctx->bar.init(nthreads);
    const unsigned nw = (nthreads + kWave - 1) / kWave;
    for (unsigned w = 0; w < nw; ++w) {
        const unsigned lanes = (w + 1) * kWave <= nthreads ? kWave : nthreads - w * kWave;
        ctx->waves[w].bar.init(lanes);
        std::memset(ctx->waves[w].mirror, 0, sizeof(ctx->waves[w].mirror));
    }
    g_diverged = false;
    g_block = ctx;
    gridDim.x = nblocks;
    std::vector<std::thread> th;
    th.reserve(nthreads);
    for (unsigned t = 0; t < nthreads; ++t)
        th.emplace_back([&, t] {
            t_tid = t;
            threadIdx.x = t;
            for (unsigned b = 0; b < nblocks && !g_diverged; ++b) {
                blockIdx.x = b;
                kernel();
                ctx->bar.wait();          // the next block reuses the statics that stand for LDS
            }
        });
    for (auto& t : th) t.join();
    g_block = nullptr;
    delete ctx;
    return !g_diverged;            // false: some lanes skipped a convergent operation (barrier, cross-lane read)
}
}  // namespace hipsim
