// Small HIP runtime helpers (RAII device / pinned buffers, error plumbing).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <mutex>
#include <vector>
#include <stdexcept>
#include <string>

namespace rgr {

struct HipError : std::runtime_error {
    hipError_t code;
    HipError(hipError_t c, const char* what, const char* file, int line)
        : std::runtime_error(std::string(what) + ": " + hipGetErrorString(c) + " (" + file + ":" + std::to_string(line) + ")"), code(c) {}
};

#define RGR_HIP(expr)                                                        \
    do {                                                                     \
        hipError_t _e = (expr);                                              \
        if (_e != hipSuccess) throw ::rgr::HipError(_e, #expr, __FILE__, __LINE__); \
    } while (0)

// Grow-only device buffer.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
    }
    // Ensure capacity; contents are NOT preserved on growth.
    void ensure(size_t n) {
        if (n <= bytes) return;
        release();
        size_t want = n + n / 8 + 256;
        RGR_HIP(hipMalloc(&p, want));
        bytes = want;
    }
    // Ensure capacity keeping the first `keep` bytes.
    void ensure_preserve(size_t n, size_t keep) {
        if (n <= bytes) return;
        void* old = p;
        const size_t want = n + n / 2 + 256;
        void* np = nullptr;
        RGR_HIP(hipMalloc(&np, want));
        if (old && keep) RGR_HIP(hipMemcpy(np, old, keep, hipMemcpyDeviceToDevice));
        if (old) (void)hipFree(old);
        p = np;
        bytes = want;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

struct PinnedBuf {
    void* p = nullptr;
    size_t bytes = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() { release(); }
    void release() {
        if (p) { (void)hipHostFree(p); p = nullptr; bytes = 0; }
    }
    void ensure(size_t n) {
        if (n <= bytes) return;
        release();
        size_t want = n + n / 8 + 256;
        RGR_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
        bytes = want;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// Recycled pinned host blocks for the results of the host-buffer entry points: hipHostMalloc costs
// milliseconds per call (it pins pages), so a result takes its tuple block from here and hands it back at
// rgr_result_free.  Shared (shared_ptr) between the handle and every outstanding result, so a result may
// outlive its handle.  Blocks are powers of two >= 64 KiB; at most `keep_bytes` stay cached.
class PinnedPool {
   public:
    explicit PinnedPool(size_t keep_bytes = size_t(16) << 30) : keep_(keep_bytes) {}
    PinnedPool(const PinnedPool&) = delete;
    PinnedPool& operator=(const PinnedPool&) = delete;
    ~PinnedPool() { for (auto& b : free_) (void)hipHostFree(b.p); }
    void* take(size_t bytes, size_t* cap) {
        size_t want = size_t(64) << 10;
        if (bytes <= (size_t(1) << 30)) { while (want < bytes) want <<= 1; }
        else want = (bytes + (size_t(256) << 20) - 1) & ~((size_t(256) << 20) - 1);      // big blocks: 256 MiB steps, not powers of two
        {
            std::lock_guard<std::mutex> g(mu_);
            size_t best = free_.size();
            for (size_t i = 0; i < free_.size(); ++i)
                if (free_[i].cap >= want && free_[i].cap <= want + want / 4 && (best == free_.size() || free_[i].cap < free_[best].cap)) best = i;
            if (best != free_.size()) {
                void* p = free_[best].p;
                *cap = free_[best].cap;
                held_ -= *cap;
                free_[best] = free_.back(); free_.pop_back();
                return p;
            }
        }
        void* p = nullptr;
        RGR_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
        *cap = want;
        return p;
    }
    void give(void* p, size_t cap) {
        if (!p) return;
        {
            std::lock_guard<std::mutex> g(mu_);
            if (held_ + cap <= keep_) { free_.push_back(Block{p, cap}); held_ += cap; return; }
        }
        (void)hipHostFree(p);
    }

   private:
    struct Block { void* p; size_t cap; };
    std::mutex mu_;
    std::vector<Block> free_;
    size_t held_ = 0, keep_;
};

}  // namespace rgr
