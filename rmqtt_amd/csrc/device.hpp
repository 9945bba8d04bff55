// Small HIP runtime helpers (RAII device / pinned buffers, error plumbing).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
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

}  // namespace rgr
