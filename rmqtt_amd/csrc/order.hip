// Walk order of a device-resident publish batch (rgr_batch_set_order, r6).
//
// walk_kernel gives every publish topic one lane and 256 consecutive topics one block; each lane then reads one 32-byte edge record per trie node
// it visits.  In the caller's order neighbouring lanes have nothing in common and every record is a separate sector from HBM; with the batch sorted
// by its leading tokens the lanes of a wave walk the same upper trie levels and the same hot subscriber runs at the same time — measured through
// the product API on topics sorted on the host (tools/walk_order_lab.py, profiles/r06g_*): walk -16 % at BASELINE configs[1] (1 997 -> 2 263 M
// topics/s), walk -24 %, count / compact -29 %, expansion -3.6 % at configs[2] (3/10 scale).  Here the library does the reordering itself:
// keys = the first six level tokens of every topic, three stable radix sorts of (key, batch index) pairs (rocPRIM through hipCUB — a library primitive for a
// once-per-batch preprocessing step; nothing of the hot path), then the token arrays are gathered into that order.  Tuples still name the
// caller's topic index (the permutation rides on the topic-id indirection of the compaction: no cost per hit); windows enumerate topics in walk
// order and rgr_window.d_topic_order says which.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdlib>

#include "kernels.hpp"

namespace rgr {
namespace {

// key of topic idx_in[k] (or of topic k when idx_in is null): levels `first` and `first + 1` as token ids, high word first
__global__ __launch_bounds__(256) void order_keys_kernel(const uint32_t* __restrict__ tokens, const uint64_t* __restrict__ tok_off, const uint8_t* __restrict__ tflags,
                                                         uint32_t n, uint32_t first, const uint32_t* __restrict__ idx_in, unsigned long long* __restrict__ keys,
                                                         uint32_t* __restrict__ idx) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const uint32_t t = idx_in ? idx_in[k] : k;
    const uint64_t o = tok_off[t];
    const uint32_t L = uint32_t(tok_off[t + 1] - o);
    // invalid topics (no tokens) sort behind everything else; '$'-topics keep their own first tokens (they walk other records anyway)
    unsigned long long key = first ? 0ull : ~0ull;
    if (!(tflags[t] & kTopicInvalid) && L > first) key = (static_cast<unsigned long long>(tokens[o + first]) << 32) | (L > first + 1 ? tokens[o + first + 1] : 0u);
    keys[k] = key;
    if (idx) idx[k] = t;
}

__global__ __launch_bounds__(256) void order_len_kernel(const uint32_t* __restrict__ perm, const uint64_t* __restrict__ tok_off, uint32_t n, uint32_t* __restrict__ len) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) { const uint32_t t = perm[k]; len[k] = uint32_t(tok_off[t + 1] - tok_off[t]); }
}

__global__ __launch_bounds__(256) void order_gather_kernel(const uint32_t* __restrict__ perm, uint32_t n, const uint64_t* __restrict__ off_old, const uint32_t* __restrict__ tok_old,
                                                           const uint8_t* __restrict__ fl_old, const uint64_t* __restrict__ off_new, uint32_t* __restrict__ tok_new,
                                                           uint8_t* __restrict__ fl_new) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const uint32_t t = perm[k];
    const uint64_t a = off_old[t], b = off_new[k];
    const uint32_t L = uint32_t(off_old[t + 1] - a);
    for (uint32_t i = 0; i < L; ++i) tok_new[b + i] = tok_old[a + i];
    fl_new[k] = fl_old[t];
}

__global__ __launch_bounds__(256) void order_compose_kernel(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ ids, uint32_t n, uint32_t* __restrict__ out) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) out[k] = ids[perm[k]];
}

__global__ __launch_bounds__(256) void order_compose_kernel64(const uint32_t* __restrict__ perm, const unsigned long long* __restrict__ in, uint32_t n, unsigned long long* __restrict__ out) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) out[k] = in[perm[k]];
}

}  // namespace

size_t order_sort_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, static_cast<const unsigned long long*>(nullptr), static_cast<unsigned long long*>(nullptr),
                                             static_cast<const uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), int(n), 0, 64, hipStream_t(nullptr));
    return bytes;
}

// perm[k] = batch index of the topic walked k-th: topics sorted by their first SIX level tokens, ties in batch order — three stable radix sorts,
// least significant pair of levels first (two levels: 2.02 -> 2.17 G topics/s at BASELINE configs[1]; sorted by the whole string on the host: 2.26 G,
// profiles/r06g_*, r06h_*).  keys / keys_tmp: [n] u64 scratch, idx_tmp: [n] u32 scratch, temp: order_sort_temp_bytes(n).
int launch_order_sort(const uint32_t* tokens, const uint64_t* tok_off, const uint8_t* tflags, uint32_t n, unsigned long long* keys, unsigned long long* keys_tmp,
                      uint32_t* idx_tmp, uint32_t* perm, void* temp, size_t temp_bytes, void* stream) {
    if (!n) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t nb = (n + 255) / 256;
    // RGR_ORDER_LEVELS (A/B switch, read per sort): how many leading levels order the batch: one stable sort per pair of levels, the least significant
    // pair first.  Measured (profiles/r06n_*): 4 / 6 / 8 levels -> walk 5.58 / 4.89 / 4.94 ms per 10 M topics at BASELINE configs[2], configs[1] 2.29 / 2.45 /
    // 2.43 G topics/s, runs 946 / 999 / 993 M; tuples and ids24 within +-1 %.  Six.
    uint32_t levels = 6;
    if (const char* e = std::getenv("RGR_ORDER_LEVELS")) { const int v = std::atoi(e); if (v >= 2 && v <= 16) levels = uint32_t(v) & ~1u; }
    bool first = true;
    for (uint32_t lv = levels; lv >= 2; lv -= 2) {
        // keys of levels lv - 2, lv - 1: of topic k on the first pass (idx_tmp = identity), of the topic at position k afterwards (idx_tmp = perm so far)
        order_keys_kernel<<<nb, 256, 0, s>>>(tokens, tok_off, tflags, n, lv - 2, first ? nullptr : perm, keys, idx_tmp);
        const int rc = int(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, keys_tmp, idx_tmp, perm, int(n), 0, 64, s));
        if (rc) return rc;
        first = false;
    }
    return 0;
}

void launch_order_len(const uint32_t* perm, const uint64_t* tok_off, uint32_t n, uint32_t* len, void* stream) {
    if (n) order_len_kernel<<<(n + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(perm, tok_off, n, len);
}

void launch_order_gather(const uint32_t* perm, uint32_t n, const uint64_t* off_old, const uint32_t* tok_old, const uint8_t* fl_old, const uint64_t* off_new, uint32_t* tok_new,
                         uint8_t* fl_new, void* stream) {
    if (n) order_gather_kernel<<<(n + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(perm, n, off_old, tok_old, fl_old, off_new, tok_new, fl_new);
}

// publish attributes (8 bytes each) into walk order
void launch_order_gather_attrs(const uint32_t* perm, const PublishAttr* in, uint32_t n, PublishAttr* out, void* stream) {
    static_assert(sizeof(PublishAttr) == 8, "gathered as 64-bit words");
    if (n) order_compose_kernel64<<<(n + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(perm, reinterpret_cast<const unsigned long long*>(in), n, reinterpret_cast<unsigned long long*>(out));
}

void launch_order_compose(const uint32_t* perm, const uint32_t* ids, uint32_t n, uint32_t* out, void* stream) {
    if (n) order_compose_kernel<<<(n + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(perm, ids, n, out);
}

}  // namespace rgr
