// Hand-written HIP kernels for gfx950 (MI355X / CDNA4): publish-topic matching.
//
// Integer / indexing work only — HBM- and latency-bound, no MFMA.  Wave = 64 lanes.
//
//   walk_kernel     one lane per publish topic; depth-first walk of the subscription trie in
//                   exactly TopicTree::matches' order (rmqtt/src/trie.rs:327-375): '#'-child
//                   item, then the '+' subtree, then the exact-child subtree; at path end the
//                   node's own filter, then its '#' child ("parent match").  '$'-topics skip
//                   the wildcard steps at the root (trie.rs:342-346).  The block's '/'-tokenised
//                   topics are staged into LDS with coalesced loads; the per-lane DFS stack
//                   lives in a second LDS window addressed like the tokens (one word per
//                   level) and spills to HBM scratch beyond the window, so topic depth is
//                   unbounded.  One 32-byte edge record is read per visited trie node.
//   tok_count/tok_fill  device tokeniser: Topic::from_str + dictionary lookup, lane per topic.
//   retain_*        RetainTree::matches as level-synchronous, load-balanced frontier rounds
//                   over a preorder-numbered trie of retained topics.
//   count/scan/compact  per-topic hit counts -> exclusive scans -> dense list of
//                   (topic, subscriber-run) pairs with their output offsets (block-cooperative
//                   for pair lists longer than kBigPairs).
//   tiles_kernel    for every output tile, the first pair that intersects it.
//   expand_kernel   load-balanced expansion: each block owns kTile consecutive output
//                   positions, finds the owning pair of each position by binary search in LDS
//                   and streams (topic_idx, sub_id, qos) tuples out fully coalesced with
//                   nontemporal 12-byte stores.
//   scatter_*       incremental epoch update: patch dirty edge records / filter descriptors.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "kernels.hpp"
#include "match_core.hpp"

namespace rgr {

namespace {

#ifndef RGR_WALK_THREADS
#define RGR_WALK_THREADS 256
#endif
#ifndef RGR_WALK_WINDOW
#define RGR_WALK_WINDOW 2560             // LDS words per array (tokens, stack): 2 x 10 KiB per 256 topics
#endif
constexpr int kWalkThreads = RGR_WALK_THREADS;
// the overflow re-walk handles the few topics with more than slot_cap matched filters — the heaviest walks of the chunk.
// One wave per block spreads them over the CUs instead of packing 256 of them into each of a handful of blocks.
constexpr int kOvfThreads = 64;
constexpr int kWalkWindow = RGR_WALK_WINDOW;
#ifndef RGR_EXPAND_THREADS
#define RGR_EXPAND_THREADS 1024
#endif
#ifndef RGR_EXPAND_PER_THREAD
#define RGR_EXPAND_PER_THREAD 2
#endif
// Timing diagnostics for the delivery variant of the expansion (tools/deliver_sweep.sh builds the library once per switch; the results
// of such a build are WRONG by construction — they only tell where the time goes): RGR_DIAG_NO_ATTRS (no attribute gather: the sub id stands
// in for the client index), RGR_DIAG_NO_CAND_STORE.  (RGR_DIAG_NO_PAIR_COUNTS of profiles/r03h_deliver_sweep.txt became the product: r3j.)
#ifndef RGR_EXPAND_NT
#define RGR_EXPAND_NT 1          // nontemporal tuple stores: the output is write-once, keep L2 for the subscriber runs
#endif
constexpr int kExpandThreads = RGR_EXPAND_THREADS;
constexpr int kExpandPerThread = RGR_EXPAND_PER_THREAD;
constexpr int kTile = kExpandThreads * kExpandPerThread;   // 2048 hits = 24 KiB of tuples
constexpr int kScanThreads = 256;
constexpr int kScanPerThread = 8;
constexpr int kScanBlock = kScanThreads * kScanPerThread;  // 2048 topics per scan block

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// --------------------------------------------------------------------------- epoch delta
__global__ __launch_bounds__(256) void scatter_edges_kernel(EdgeEntry* __restrict__ dst, const uint32_t* __restrict__ slots,
                                                            const EdgeEntry* __restrict__ recs, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4* s = reinterpret_cast<const uint4*>(recs + i);
    uint4* d = reinterpret_cast<uint4*>(dst + slots[i]);
    d[0] = s[0]; d[1] = s[1];
}
__global__ __launch_bounds__(256) void scatter_desc_kernel(FilterDesc* __restrict__ dst, const uint32_t* __restrict__ fids,
                                                           const FilterDesc* __restrict__ recs, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[fids[i]] = recs[i];
}

// --------------------------------------------------------------------------- tokeniser
// Device-side Topic::from_str + dictionary lookup: one lane per topic, one pass over its bytes
// per kernel (count, then fill after the exclusive scan of the level counts).
__global__ __launch_bounds__(256) void tok_count_kernel(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offs, uint32_t n,
                                                        uint32_t* __restrict__ level_cnt, uint8_t* __restrict__ tflags,
                                                        const uint8_t* __restrict__ force_invalid, uint32_t lane_shift) {
    // (lane_shift: a batch that cannot fill the chip uses every 2^shift-th lane — a wave runs as long as its longest topic; see launch_walk)
    if (threadIdx.x & ((1u << lane_shift) - 1u)) return;
    const uint32_t t = (blockIdx.x * 256 + threadIdx.x) >> lane_shift;
    if (t >= n) return;
    if (force_invalid && force_invalid[t]) { level_cnt[t] = 0; tflags[t] = kTopicInvalid; return; }
    bool meta;
    const int64_t nl = scan_topic_at(WordBytes(blob + offs[t]), offs[t + 1] - offs[t], &meta, [](int64_t, uint64_t, uint32_t, uint64_t, int) {});
    level_cnt[t] = nl < 0 ? 0u : uint32_t(nl);
    tflags[t] = nl < 0 ? kTopicInvalid : meta ? kTopicMeta : 0;
}

// (r7g) Two phases per lane.  One pass that looks a level up the moment its '/' is met makes the 64 lanes of a wave — whose levels end at different bytes —
// run the dictionary probe (dependent loads: slot, entry, the stored string's bytes) once per DISTINCT level-end position of the wave, ~30 times for
// topics of 7 levels: 171 us for a batch of 2 600 topics (profiles/r07e_*), the largest kernel of a micro-batch pass.  Now the byte scan only records
// each level's (start, length, hash) in LDS, and the probes run level by level, all lanes on their k-th level together.
constexpr int kTokFillThreads = 128, kTokFillLevels = 16;          // levels held in LDS per lane (32 KiB per block); deeper levels are looked up in the scan
__global__ __launch_bounds__(kTokFillThreads) void tok_fill_kernel(DictView d, const uint8_t* __restrict__ blob, const uint64_t* __restrict__ offs,
                                                                   uint32_t n, const uint64_t* __restrict__ tok_off,
                                                                   const uint8_t* __restrict__ tflags, uint32_t* __restrict__ tokens, uint32_t lane_shift) {
    __shared__ uint32_t s_seg[kTokFillLevels][kTokFillThreads], s_len[kTokFillLevels][kTokFillThreads];
    __shared__ uint64_t s_hash[kTokFillLevels][kTokFillThreads];
    if (threadIdx.x & ((1u << lane_shift) - 1u)) return;
    const uint32_t t = (blockIdx.x * kTokFillThreads + threadIdx.x) >> lane_shift;
    if (t >= n || (tflags[t] & kTopicInvalid)) return;
    const uint8_t* s = blob + offs[t];
    uint32_t* out = tokens + tok_off[t];
    bool meta;
    const int64_t nlev = scan_topic_at(WordBytes(s), offs[t + 1] - offs[t], &meta, [&](int64_t idx, uint64_t seg, uint32_t sl, uint64_t h, int kind) {
        if (kind) out[idx] = kind == 1 ? kTokPlus : kTokHash;
        else if (idx < kTokFillLevels) { s_seg[idx][threadIdx.x] = uint32_t(seg); s_len[idx][threadIdx.x] = sl; s_hash[idx][threadIdx.x] = h; }
        else out[idx] = dict_find(d, s + seg, sl, h);
        if (kind && idx < kTokFillLevels) s_len[idx][threadIdx.x] = 0xFFFFFFFFu;      // (a wildcard level: nothing to look up)
    });
    const int held = nlev < kTokFillLevels ? int(nlev) : kTokFillLevels;
    for (int k = 0; k < held; ++k) {
        const uint32_t sl = s_len[k][threadIdx.x];
        if (sl != 0xFFFFFFFFu) out[k] = dict_find(d, s + s_seg[k][threadIdx.x], sl, s_hash[k][threadIdx.x]);
    }
}

// PUBLISH packets: lane per packet (match_core.hpp publish_scan); then the topic-name fields are gathered into a
// dense blob for the tokeniser.
__global__ __launch_bounds__(256) void publish_scan_kernel(const uint8_t* __restrict__ pkts, const uint64_t* __restrict__ offs, uint32_t n, int version,
                                                           PubInfo* __restrict__ info, uint32_t* __restrict__ topic_len, uint8_t* __restrict__ bad,
                                                           const uint32_t* __restrict__ from_ids, PublishAttr* __restrict__ attrs) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    PubInfo pi;
    publish_scan(pkts, offs[t], offs[t + 1] - offs[t], version, pi);
    info[t] = pi;
    topic_len[t] = pi.topic_len;
    bad[t] = pi.error;
    if (attrs) attrs[t] = PublishAttr{from_ids ? from_ids[t] : kNone, uint32_t(pi.qos) | (uint32_t(pi.retain) << 2)};
}
__global__ __launch_bounds__(256) void publish_topics_kernel(const uint8_t* __restrict__ pkts, const PubInfo* __restrict__ info, uint32_t n,
                                                             const uint64_t* __restrict__ topic_offs, uint8_t* __restrict__ blob) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const PubInfo pi = info[t];
    const uint8_t* s = pkts + pi.topic_off;
    uint8_t* d = blob + topic_offs[t];
    for (uint32_t i = 0; i < pi.topic_len; ++i) d[i] = s[i];
}

// exclusive scan u32 -> u64 (three phases, same block shape as the chunk scan)
__global__ __launch_bounds__(kScanThreads) void scan1_reduce_kernel(const uint32_t* __restrict__ in, uint32_t n, uint64_t* block_tmp) {
    __shared__ unsigned long long s_w[kScanThreads / 64];
    const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * kScanPerThread;
    unsigned long long a = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) if (base + k < n) a += in[base + k];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long r = 0; for (int i = 0; i < kScanThreads / 64; ++i) r += s_w[i]; block_tmp[blockIdx.x] = r; }
}
__global__ __launch_bounds__(kScanThreads) void scan1_down_kernel(const uint32_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t n,
                                                                  const uint64_t* block_tmp, uint32_t nblocks) {
    __shared__ unsigned long long s_w[kScanThreads / 64];
    const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * kScanPerThread;
    unsigned long long la[kScanPerThread], a = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) { la[k] = a; if (base + k < n) a += in[base + k]; }
    unsigned long long xa = a;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned long long y = __shfl_up(xa, o, 64); if (lane >= o) xa += y; }
    if (lane == 63) s_w[w] = xa;
    __syncthreads();
    unsigned long long pre = 0;
    for (int i = 0; i < w; ++i) pre += s_w[i];
    const unsigned long long t0 = block_tmp[blockIdx.x] + pre + (xa - a);
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) if (base + k < n) out[base + k] = t0 + la[k];
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = block_tmp[nblocks];
}

// --------------------------------------------------------------------------- walk
template <bool OVF>
__global__ __launch_bounds__(kWalkThreads) void walk_kernel(TrieView tv, WalkArgs a) {
    __shared__ uint32_t s_tok[OVF ? 1 : kWalkWindow];
    __shared__ uint32_t s_path[OVF ? 1 : kWalkWindow];
    __shared__ unsigned long long s_visited;         // (statistics only) the block's visited nodes: ONE global atomic per block, not per wave

    const int tid = threadIdx.x;
    uint32_t tl;            // chunk-local topic index
    bool active;
    uint64_t win_base = 0;  // token index of the first staged token
    uint32_t staged = 0;    // tokens held in LDS

    if (OVF) {
        const uint32_t i = blockIdx.x * kOvfThreads + tid;
        active = i < min(*a.ovf_count, a.n);
        tl = active ? a.ovf_list[i] : 0;
    } else {
        const uint32_t per_block = uint32_t(kWalkThreads) >> a.lane_shift;
        const uint32_t t0 = blockIdx.x * per_block;
        tl = t0 + (uint32_t(tid) >> a.lane_shift);
        active = (uint32_t(tid) & ((1u << a.lane_shift) - 1u)) == 0 && tl < a.n;
        const uint32_t t1 = min(t0 + per_block, a.n);
        win_base = a.tok_off[a.topic_base + t0];
        const uint64_t win_end = a.tok_off[a.topic_base + t1];
        const uint64_t span = win_end - win_base;
        staged = span < uint64_t(kWalkWindow) ? uint32_t(span) : uint32_t(kWalkWindow);
        for (uint32_t i = tid; i < staged; i += kWalkThreads) s_tok[i] = a.tokens[win_base + i];   // coalesced
        if (tid == 0) s_visited = 0;
        __syncthreads();
    }

    uint32_t cnt = 0;
    uint32_t visited = 0;
    if (active) {
        const uint32_t gt = a.topic_base + tl;
        const uint64_t off0 = a.tok_off[gt];
        const uint32_t L = uint32_t(a.tok_off[gt + 1] - off0);
        const uint8_t fl = a.tflags[gt];
        const uint64_t rel = off0 - win_base;          // position of level 0 inside the window
        const uint64_t arena_base = OVF ? a.ovf_base[tl] : 0;

        auto emit = [&](uint32_t fid) {
            if (OVF) { if (arena_base + cnt < a.ovf_arena_cap) a.ovf_arena[arena_base + cnt] = fid; }
            else if (cnt < a.slot_cap) a.slots[uint64_t(cnt) * a.n + tl] = fid;
            cnt++;
        };
        const EdgeEntry* edges = tv.edges;
        auto load = [&](uint32_t slot, U4& e0, U4& e1) {
            const uint4* ep = reinterpret_cast<const uint4*>(edges + slot);
            const uint4 a0 = ep[0], a1 = ep[1];
            e0 = U4{a0.x, a0.y, a0.z, a0.w};
            e1 = U4{a1.x, a1.y, a1.z, a1.w};
        };
        if (!(fl & kTopicInvalid)) {
            // Tokens and DFS stack: the LDS window when the WHOLE topic lies inside it (the common case: 10 words per topic
            // are staged), HBM otherwise.  The choice is made once per topic and the walk is instantiated twice, so the LDS
            // instance reads with ds_read: a per-access `i < staged ? s_tok[i] : a.tokens[..]` is lowered to flat_load, which
            // sends LDS-resident reads through the vector-memory path (r2: DESIGN.md §10).
            const uint32_t rel32 = uint32_t(rel);
            if (!OVF && rel + L <= uint64_t(staged)) {
                visited = walk_topic(
                    tv.root, tv.mask, L, (fl & kTopicMeta) != 0, [&](uint32_t d) { return s_tok[rel32 + d]; },
                    [&](uint32_t d) { return s_path[rel32 + d]; }, [&](uint32_t d, uint32_t v) { s_path[rel32 + d] = v; }, emit, load);
            } else {
                const uint32_t* gtok = a.tokens + off0;
                uint32_t* gpath = a.path_scratch + off0;
                visited = walk_topic(
                    tv.root, tv.mask, L, (fl & kTopicMeta) != 0, [&](uint32_t d) { return gtok[d]; }, [&](uint32_t d) { return gpath[d]; },
                    [&](uint32_t d, uint32_t v) { gpath[d] = v; }, emit, load);
            }
        }
        if (!OVF) {
            a.pair_cnt[tl] = cnt;
            if (cnt > a.slot_cap) {
                const uint32_t i = atomicAdd(a.ovf_count, 1u);
                a.ovf_list[i] = tl;
                a.ovf_base[tl] = atomicAdd(a.ovf_cursor, (unsigned long long)cnt);
            }
        }
    }
    if (a.visited && !OVF) {
        // (r6: per WAVE this was 156 k atomics on one address per 10 M topics — 0.7 ms of a 5.6 ms walk when the caller asks for the statistic)
        const unsigned long long v = wave_sum(visited);
        if ((tid & 63) == 0 && v) atomicAdd(&s_visited, v);
        __syncthreads();
        if (tid == 0 && s_visited) atomicAdd(a.visited, s_visited);
    }
}

// --------------------------------------------------------------------------- retain (RetainTree::matches)
// Level-synchronous, load-balanced frontier expansion over the preorder-numbered trie of retained
// topics.  Per level: retain_step_kernel (one lane per frontier item: probe / child-range /
// descriptors), exclusive scans, retain_expand_kernel (every block owns kTile consecutive slots
// of the NEXT frontier and finds their producing item by binary search in LDS — a '+' over a
// 65 k-child node is spread over many blocks instead of serialising one lane), and
// retain_emit_kernel (descriptors appended per filter, in order).  The emitted descriptor lists
// feed the shared count / scan / compact / tiles / expand kernels.
__device__ __forceinline__ uint32_t retain_probe(const REdge* edges, uint32_t mask, uint32_t parent, uint32_t token) {
    for (uint32_t s = edge_hash(parent, token) & mask;; s = (s + 1) & mask) {
        const uint4 e = *reinterpret_cast<const uint4*>(edges + s);
        if (e.x == kEdgeEmpty) return kNone;
        if (e.x == parent && e.y == token) return e.z;
    }
}

__global__ __launch_bounds__(256) void retain_step_kernel(RetainView rv, RetainRound r) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= r.m) return;
    const uint32_t f = r.f_filter ? r.f_filter[i] : i;
    const uint32_t node = r.f_filter ? r.f_node[i] : 0u;
    const uint32_t gt = r.topic_base + f;
    RetainStep st{0, 0, kNone, kNone};
    if (!(r.tflags[gt] & kTopicInvalid)) {
        const uint64_t off0 = r.tok_off[gt];
        const uint32_t L = uint32_t(r.tok_off[gt + 1] - off0);
        const uint32_t d = r.fdepth[f];
        const uint32_t tok = d < L ? r.tokens[off0 + d] : 0u;
        const uint32_t tok_next = d + 1 < L ? r.tokens[off0 + d + 1] : 0u;
        const REdge* edges = rv.edges;
        const uint32_t mask = rv.mask;
        const GcEdge* gc = rv.gc_edges;
        const uint32_t gmask = rv.gc_mask;
        st = retain_step(
            rv, node, d, L, tok, tok_next, [&](uint32_t p, uint32_t t) { return retain_probe(edges, mask, p, t); },
            [&](uint32_t g, uint32_t t, uint32_t& begin, uint32_t& count) {
                for (uint32_t s = edge_hash(g, t) & gmask;; s = (s + 1) & gmask) {
                    const uint4 e = *reinterpret_cast<const uint4*>(gc + s);
                    if (e.x == kEdgeEmpty) { begin = 0; count = 0; return; }
                    if (e.x == g && e.y == t) { begin = e.z; count = e.w; return; }
                }
            });
    }
    r.cnt[i] = st.cnt; r.payload[i] = st.payload;
    r.e0[i] = st.e0; r.e1[i] = st.e1;
    r.ecnt[i] = (st.e0 != kNone) + (st.e1 != kNone);
}

constexpr uint32_t kRetainSmall = 32;   // expansions up to this size are written by their own lane

// Next frontier, part 1: every item writes its own contribution at out_off[i] when it is small;
// big expansions ('+' over a node with many children) are queued for retain_big_kernel.
__global__ __launch_bounds__(256) void retain_scatter_kernel(RetainView rv, RetainRound r, const uint64_t* __restrict__ out_off,
                                                             uint32_t* __restrict__ nf_filter, uint32_t* __restrict__ nf_node,
                                                             uint32_t* __restrict__ big_list, uint32_t* __restrict__ big_count) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= r.m) return;
    const uint32_t c = r.cnt[i];
    if (!c) return;
    if (c > kRetainSmall) { big_list[atomicAdd(big_count, 1u)] = i; return; }
    const uint32_t f = r.f_filter ? r.f_filter[i] : i;
    const uint32_t pay = r.payload[i];
    const uint64_t o = out_off[i];
    for (uint32_t k = 0; k < c; ++k) { nf_filter[o + k] = f; nf_node[o + k] = retain_child(rv, pay, k); }
}

// Next frontier, part 2: one block per queued big expansion, coalesced copy of its child list.
// Every big item owns the slot range [out_off[i], out_off[i]+cnt) so the result does not depend
// on the queue order.
__global__ __launch_bounds__(256) void retain_big_kernel(RetainView rv, RetainRound r, const uint64_t* __restrict__ out_off,
                                                         uint32_t* __restrict__ nf_filter, uint32_t* __restrict__ nf_node,
                                                         const uint32_t* __restrict__ big_list, const uint32_t* __restrict__ big_count) {
    const uint32_t nb = *big_count;
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const uint32_t i = big_list[b];
        const uint32_t c = r.cnt[i], pay = r.payload[i];
        const uint32_t f = r.f_filter ? r.f_filter[i] : i;
        const uint64_t o = out_off[i];
        for (uint32_t k = threadIdx.x; k < c; k += 256) { nf_filter[o + k] = f; nf_node[o + k] = retain_child(rv, pay, k); }
    }
}

// Descriptors of this round appended to the arena in item order.  A filter's items are
// contiguous in the frontier and it emits only in its final round, so the exclusive scan value
// at its first item is the base of its descriptor list and the value after its last item the
// end: no atomics.  Earlier rounds write base == end; the final round overwrites both.
__global__ __launch_bounds__(256) void retain_emit_kernel(RetainRound r, const uint64_t* __restrict__ epos, uint64_t g_base,
                                                          uint32_t* __restrict__ arena, uint64_t* __restrict__ ovf_base,
                                                          uint64_t* __restrict__ ovf_end) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= r.m) return;
    const uint32_t f = r.f_filter ? r.f_filter[i] : i;
    uint64_t p = g_base + epos[i];
    const uint32_t n = r.ecnt[i];
    if (i == 0 || (r.f_filter ? r.f_filter[i - 1] : i - 1) != f) ovf_base[f] = p;
    if (i == r.m - 1 || (r.f_filter ? r.f_filter[i + 1] : i + 1) != f) ovf_end[f] = p + n;
    if (!n) return;
    if (r.e0[i] != kNone) arena[p++] = r.e0[i];
    if (r.e1[i] != kNone) arena[p++] = r.e1[i];
}

__global__ __launch_bounds__(256) void retain_advance_kernel(RetainRound r, uint32_t n, uint32_t* __restrict__ fdepth) {
    const uint32_t f = blockIdx.x * 256 + threadIdx.x;
    if (f >= n) return;
    const uint32_t gt = r.topic_base + f;
    const uint64_t off0 = r.tok_off[gt];
    const uint32_t L = uint32_t(r.tok_off[gt + 1] - off0);
    const uint32_t d = fdepth[f];
    if (d > L) return;                                  // finished long ago
    bool jump = false;
    if (d < L && !(r.tflags[gt] & kTopicInvalid))
        jump = retain_jumps(r.tokens[off0 + d], d + 1 < L, d + 1 < L ? r.tokens[off0 + d + 1] : 0u);
    fdepth[f] = d + (jump ? 2u : 1u);
}

__global__ __launch_bounds__(256) void retain_finish_kernel(uint32_t n, const uint64_t* __restrict__ ovf_base,
                                                            const uint64_t* __restrict__ ovf_end, uint32_t* __restrict__ pair_cnt) {
    const uint32_t f = blockIdx.x * 256 + threadIdx.x;
    if (f < n) pair_cnt[f] = uint32_t(ovf_end[f] - ovf_base[f]);
}

// --------------------------------------------------------------------------- dense matched-filter list
// REPS: instead of the filter id, the sub id of the filter's FIRST subscriber (kNone when it has none): a host that keeps
// its own relations map keyed by filter string (the reference's AllRelationsMap) finds the filter through that relation
// and expands from its own map, as router.rs:194-231 does — no filter-id bookkeeping on the host, shard-independent.
template <bool REPS>
__global__ __launch_bounds__(256) void pairs_dense_kernel(TrieView tv, ChunkArrays c, const uint64_t* __restrict__ off, uint32_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= c.n) return;
    const uint32_t cnt = c.pair_cnt[t];
    const uint64_t o = off[t];
    for (uint32_t j = 0; j < cnt; ++j) {
        const uint32_t fid = pair_fid(c, t, cnt, j);
        if (REPS) { const FilterDesc fd = tv.filt[fid]; out[o + j] = fd.count ? tv.subs[fd.begin].sub_id : kNone; }
        else out[o + j] = fid;
    }
}

// --------------------------------------------------------------------------- count
__global__ __launch_bounds__(256) void count_kernel(TrieView tv, ChunkArrays c) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= c.n) return;
    if (c.pair_cnt[t] > kBigPairs) { c.big_list[atomicAdd(c.big_count, 1u)] = t; return; }   // -> count_big_kernel
    count_topic(tv, c, t);
}

// Long pair lists (a retained-path '+' over a wide node emits one descriptor per child; a hot
// publish topic can match hundreds of filters): one block per topic instead of one lane.
// (r6) eight descriptors per thread and step, all in flight together: a retained-path filter like `+/+/+/#` owns a list of 10^5 - 10^6 descriptors, walked by
// ONE block — at one gather per thread and step its 4 000 dependent rounds were the whole kernel (1.78 ms per launch at BASELINE configs[4]).
constexpr int kBigPer = 8;
__global__ __launch_bounds__(256) void count_big_kernel(TrieView tv, ChunkArrays c) {
    __shared__ unsigned long long s_a[4], s_b[4];
    const uint32_t nb = *c.big_count;
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const uint32_t t = c.big_list[b];
        const uint32_t cnt = c.pair_cnt[t];
        if (c.ovf_base[t] + cnt > c.ovf_arena_cap && cnt > c.slot_cap) {
            if (threadIdx.x == 0) { *c.error_flag = 1; c.hit_cnt[t] = 0; c.pair_live[t] = 0; }
            continue;
        }
        unsigned long long hits = 0, live = 0;
        for (uint32_t j0 = threadIdx.x; j0 < cnt; j0 += 256 * kBigPer) {
            uint32_t n[kBigPer];
#pragma unroll
            for (int k = 0; k < kBigPer; ++k) { const uint32_t j = j0 + uint32_t(k) * 256; n[k] = j < cnt ? tv.filt[pair_fid(c, t, cnt, j)].count : 0u; }
#pragma unroll
            for (int k = 0; k < kBigPer; ++k) { hits += n[k]; live += n[k] != 0; }
        }
        hits = wave_sum(hits); live = wave_sum(live);
        if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = hits; s_b[threadIdx.x >> 6] = live; }
        __syncthreads();
        if (threadIdx.x == 0) {
            c.hit_cnt[t] = uint32_t(s_a[0] + s_a[1] + s_a[2] + s_a[3]);
            c.pair_live[t] = uint32_t(s_b[0] + s_b[1] + s_b[2] + s_b[3]);
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------- scan
// Exclusive scans of hit_cnt -> hit_off (u64) and pair_live -> pair_base (u64), three phases.
struct U2 { unsigned long long a, b; };

__device__ __forceinline__ U2 block_reduce2(unsigned long long a, unsigned long long b, U2* s_w) {
    a = wave_sum(a); b = wave_sum(b);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_w[w].a = a; s_w[w].b = b; }
    __syncthreads();
    U2 r{0, 0};
    for (int i = 0; i < kScanThreads / 64; ++i) { r.a += s_w[i].a; r.b += s_w[i].b; }
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(kScanThreads) void scan_reduce_kernel(ChunkArrays c, uint64_t* block_tmp) {
    __shared__ U2 s_w[kScanThreads / 64];
    const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * kScanPerThread;
    unsigned long long a = 0, b = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) {
        const uint32_t t = base + k;
        if (t < c.n) { a += c.hit_cnt[t]; b += c.pair_live[t]; }
    }
    const U2 r = block_reduce2(a, b, s_w);
    if (threadIdx.x == 0) { block_tmp[2 * blockIdx.x] = r.a; block_tmp[2 * blockIdx.x + 1] = r.b; }
}

// Exclusive scan of the per-block sums by ONE block of 1024 threads (each thread owns a contiguous
// slice; block-wide scan of the slice totals).  `stride` interleaved sequences are scanned at once.
template <int STRIDE>
__global__ __launch_bounds__(1024) void scan_spine_kernel(uint64_t* block_tmp, uint32_t nblocks) {
    __shared__ unsigned long long s_w[STRIDE][16];
    const uint32_t per = (nblocks + 1023) / 1024;
    const uint32_t lo = min(threadIdx.x * per, nblocks), hi = min(lo + per, nblocks);
    unsigned long long tot[STRIDE];
#pragma unroll
    for (int k = 0; k < STRIDE; ++k) tot[k] = 0;
    for (uint32_t i = lo; i < hi; ++i)
#pragma unroll
        for (int k = 0; k < STRIDE; ++k) tot[k] += block_tmp[STRIDE * i + k];
    unsigned long long x[STRIDE];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < STRIDE; ++k) {
        x[k] = tot[k];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned long long y = __shfl_up(x[k], o, 64); if (lane >= o) x[k] += y; }
        if (lane == 63) s_w[k][w] = x[k];
    }
    __syncthreads();
    unsigned long long run[STRIDE], all[STRIDE];
#pragma unroll
    for (int k = 0; k < STRIDE; ++k) {
        unsigned long long pre = 0, sum = 0;
        for (int i = 0; i < 16; ++i) { if (i < w) pre += s_w[k][i]; sum += s_w[k][i]; }
        run[k] = pre + (x[k] - tot[k]);
        all[k] = sum;
    }
    for (uint32_t i = lo; i < hi; ++i)
#pragma unroll
        for (int k = 0; k < STRIDE; ++k) { const unsigned long long v = block_tmp[STRIDE * i + k]; block_tmp[STRIDE * i + k] = run[k]; run[k] += v; }
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < STRIDE; ++k) block_tmp[STRIDE * nblocks + k] = all[k];
}

__global__ __launch_bounds__(kScanThreads) void scan_down_kernel(ChunkArrays c, const uint64_t* block_tmp, uint32_t nblocks) {
    __shared__ U2 s_w[kScanThreads / 64];
    __shared__ U2 s_pre[kScanThreads / 64];
    const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * kScanPerThread;
    unsigned long long la[kScanPerThread], lb[kScanPerThread];
    unsigned long long a = 0, b = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) {
        const uint32_t t = base + k;
        la[k] = a; lb[k] = b;
        if (t < c.n) { a += c.hit_cnt[t]; b += c.pair_live[t]; }
    }
    // exclusive scan of the per-thread totals across the block
    unsigned long long xa = a, xb = b;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long ya = __shfl_up(xa, o, 64), yb = __shfl_up(xb, o, 64);
        if (lane >= o) { xa += ya; xb += yb; }
    }
    if (lane == 63) { s_w[w].a = xa; s_w[w].b = xb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long pa = 0, pb = 0;
        for (int i = 0; i < kScanThreads / 64; ++i) { s_pre[i].a = pa; s_pre[i].b = pb; pa += s_w[i].a; pb += s_w[i].b; }
    }
    __syncthreads();
    const unsigned long long ta = block_tmp[2 * blockIdx.x] + s_pre[w].a + (xa - a);
    const unsigned long long tb = block_tmp[2 * blockIdx.x + 1] + s_pre[w].b + (xb - b);
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) {
        const uint32_t t = base + k;
        if (t < c.n) { c.hit_off[t] = ta + la[k]; c.pair_base[t] = tb + lb[k]; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { c.hit_off[c.n] = block_tmp[2 * nblocks]; c.pair_base[c.n] = block_tmp[2 * nblocks + 1]; }
}

// --------------------------------------------------------------------------- compact
// Dense (topic, subscriber-run) pairs of the chunk, in topic order.  One WAVE per 64 consecutive topics: their pairs
// occupy ONE contiguous range [pair_base[t0], pair_base[t0+64]) of the pair arrays, so the wave stages them in LDS
// (kCompactStage pairs per round; a lane deposits its topic's pairs at their final index relative to the round) and
// writes every round out with fully coalesced stores.  (r3: the lane-per-topic version wrote 3 x 4-8 bytes per pair at
// addresses ~16 pairs apart between neighbouring lanes — 1.84 ms per 2 M-topic chunk at config 3, bound by those
// scattered stores, profiles/r02g_bench_config3_kernel_stats_rocprofv3.txt.)
// Topics with more than kBigPairs matched filters are left to compact_big_kernel, which runs AFTER this kernel on the
// same stream: whatever this kernel's write-out leaves in their ranges is overwritten there.
#ifndef RGR_COMPACT_STAGE
#define RGR_COMPACT_STAGE 512
#endif
constexpr int kCompactStage = RGR_COMPACT_STAGE;
constexpr int kCompactWave = 64;
__global__ __launch_bounds__(kCompactWave) void compact_kernel(TrieView tv, ChunkArrays c, uint32_t topic_base) {
    __shared__ uint64_t s_off[kCompactStage];
    __shared__ uint32_t s_src[kCompactStage];
    __shared__ uint32_t s_topic[kCompactStage];
    __shared__ uint8_t s_qr[kCompactStage];
    const uint32_t t0 = blockIdx.x * kCompactWave;
    const uint32_t t = t0 + threadIdx.x;
    const uint32_t t1 = min(t0 + uint32_t(kCompactWave), c.n);
    const uint64_t P0 = c.pair_base[t0], P1 = c.pair_base[t1];
    const bool in = t < c.n;
    const uint32_t cnt = in ? c.pair_cnt[t] : 0u;
    const bool mine = in && cnt <= kBigPairs && c.pair_live[t] != 0;
    uint64_t p = in ? c.pair_base[t] : P1;
    uint64_t o = in ? c.hit_off[t] : 0;
    uint32_t j = 0;
    const uint32_t topic_val = !in ? 0u : c.topic_ids ? c.topic_ids[topic_base + t] : topic_base + t;
    const uint8_t qr = (in && c.pair_qr) ? uint8_t(c.pub[topic_base + t].qos_retain) : uint8_t(0);
    if (in && t == c.n - 1) c.pair_off[c.pair_base[c.n]] = c.hit_off[c.n];   // sentinel: total hits of the chunk
    for (uint64_t w0 = P0; w0 < P1; w0 += kCompactStage) {
        const uint64_t w1 = (P1 - w0) < uint64_t(kCompactStage) ? P1 : w0 + kCompactStage;
        if (mine) {
            while (j < cnt && p < w1) {                      // p >= w0: earlier rounds consumed everything below
                const FilterDesc fd = tv.filt[pair_fid(c, t, cnt, j)];
                ++j;
                if (fd.count) {
                    const uint32_t i = uint32_t(p - w0);
                    s_src[i] = fd.begin; s_topic[i] = topic_val; s_off[i] = o;
                    if (c.pair_qr) s_qr[i] = qr;
                    ++p; o += fd.count;
                }
            }
        }
        __syncthreads();
        const uint32_t m = uint32_t(w1 - w0);
        for (uint32_t i = threadIdx.x; i < m; i += kCompactWave) {
            c.pair_src[w0 + i] = s_src[i];
            c.pair_topic[w0 + i] = s_topic[i];
            c.pair_off[w0 + i] = s_off[i];
            if (c.pair_qr) c.pair_qr[w0 + i] = s_qr[i];
        }
        __syncthreads();
    }
}

// Order-preserving compaction of one long pair list by a whole block: 256 x kBigPer pairs per step (a thread owns kBigPer CONSECUTIVE pairs, their
// descriptors gathered together), block exclusive scan of the threads' (live, count) totals, carried across steps.  (r6: one pair per thread and step —
// two barriers and a dependent gather per 256 pairs — made the 10^5 - 10^6-descriptor lists of the retained path 3.98 ms per launch.)
__global__ __launch_bounds__(256) void compact_big_kernel(TrieView tv, ChunkArrays c, uint32_t topic_base) {
    __shared__ unsigned long long s_l[4], s_c[4];
    const uint32_t nb = *c.big_count;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const uint32_t t = c.big_list[b];
        const uint32_t cnt = c.pair_cnt[t];
        if (c.pair_live[t] == 0) continue;      // nothing to emit — also every topic count_big_kernel refused (list beyond the arena)
        uint64_t p0 = c.pair_base[t], o0 = c.hit_off[t];
        const uint32_t topic_val = c.topic_ids ? c.topic_ids[topic_base + t] : topic_base + t;
        const uint8_t qr = c.pair_qr ? uint8_t(c.pub[topic_base + t].qos_retain) : uint8_t(0);
        for (uint32_t j0 = 0; j0 < cnt; j0 += 256 * kBigPer) {
            const uint32_t jt = j0 + threadIdx.x * kBigPer;
            FilterDesc fd[kBigPer];
#pragma unroll
            for (int k = 0; k < kBigPer; ++k) fd[k] = jt + k < cnt ? tv.filt[pair_fid(c, t, cnt, jt + k)] : FilterDesc{0, 0};
            unsigned long long ml = 0, mc = 0;                    // this thread's live pairs / hits
#pragma unroll
            for (int k = 0; k < kBigPer; ++k) { ml += fd[k].count != 0; mc += fd[k].count; }
            unsigned long long xl = ml, xc = mc;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned long long yl = __shfl_up(xl, o, 64), yc = __shfl_up(xc, o, 64);
                if (lane >= o) { xl += yl; xc += yc; }
            }
            if (lane == 63) { s_l[w] = xl; s_c[w] = xc; }
            __syncthreads();
            unsigned long long pl = 0, pc = 0, tl = 0, tc = 0;
            for (int i = 0; i < 4; ++i) { if (i < w) { pl += s_l[i]; pc += s_c[i]; } tl += s_l[i]; tc += s_c[i]; }
            uint64_t p = p0 + pl + (xl - ml), o = o0 + pc + (xc - mc);
#pragma unroll
            for (int k = 0; k < kBigPer; ++k)
                if (fd[k].count) {
                    c.pair_src[p] = fd[k].begin;
                    c.pair_topic[p] = topic_val;
                    if (c.pair_qr) c.pair_qr[p] = qr;
                    c.pair_off[p] = o;
                    ++p; o += fd[k].count;
                }
            p0 += tl; o0 += tc;
            __syncthreads();
        }
    }
}

#include "prep_batched.inc"

// --------------------------------------------------------------------------- tiles
__global__ __launch_bounds__(256) void tiles_kernel(ChunkArrays c, uint64_t pair_lo, uint64_t pair_hi, uint64_t hit_lo, TileRec* __restrict__ tile_first) {
    const uint64_t p = pair_lo + uint64_t(blockIdx.x) * 256 + threadIdx.x;
    if (p >= pair_hi) return;
    tiles_pair_rec(c, p, pair_lo, hit_lo, kTile, tile_first);
}

// --------------------------------------------------------------------------- run descriptors for the exchange step
// One 16-byte descriptor per (topic, subscriber-run) pair of a window: what crosses xGMI in the run gather instead of the
// 12-byte tuples themselves (at config-3 fan-out ~20 runs = 320 B per publish instead of 178 KB of tuples).
__global__ __launch_bounds__(256) void pack_runs_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ topic,
                                                        const uint64_t* __restrict__ off, uint64_t n, uint32_t shard, RunDesc* __restrict__ out) {
    const uint64_t r = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    if (r < n) out[r] = RunDesc{shard, src[r], uint32_t(off[r + 1] - off[r]), topic[r]};
}

#include "expand_tuple.inc"

// packed side array of the subscriber entries (TrieView::subs_packed)
__global__ __launch_bounds__(256) void pack_subs_kernel(const SubEntry* __restrict__ subs, uint64_t n, uint32_t* __restrict__ packed) {
    const uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) { const SubEntry e = subs[i]; packed[i] = e.sub_id | ((e.qos_flags & 3u) << 30); }
}

#include "expand_compact.inc"

#include "dedup.inc"

// --------------------------------------------------------------------------- delivery results grouped by node
// SubRelationsMap is keyed by node (types.rs:486-497; router.rs:258-261 builds it from one collector per node): the host glue
// wants, per publish, one contiguous slice of delivery tuples per node.  The node index rides in bits 16-31 of the delivery
// word; this is a stable partition of every topic's tuples by that index (filter order, and sub-id order inside a filter, are
// preserved inside a node) — one radix-256 pass per byte of the largest node index in the table, i.e. one pass for up to 256
// nodes, none at all for a single-node broker.  One WAVE per topic: histogram of the digit in LDS, exclusive scan of the 256
// buckets across the lanes, then the scatter with ballot ranks (stable).
constexpr int kNodeWaves = 4;
// LDS traffic of ONE wave is issued in order, but the compiler must not move it across the phases of the wave-level
// algorithm below (the waves of a block work on different topics with different trip counts: no __syncthreads here)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__global__ __launch_bounds__(64 * kNodeWaves) void node_partition_kernel(const Tuple* __restrict__ in, Tuple* __restrict__ out,
                                                                         const uint64_t* __restrict__ hit_off, uint64_t hit_lo, uint32_t nt, uint32_t shift) {
    __shared__ uint32_t s_cnt[kNodeWaves][256];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* cnt = s_cnt[wave];
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t t = blockIdx.x * kNodeWaves + wave; t < nt; t += gridDim.x * kNodeWaves) {
        const uint64_t h0 = hit_off[t] - hit_lo, h1 = hit_off[t + 1] - hit_lo;
        if (h1 == h0) continue;
        if (h1 - h0 == 1) { if (lane == 0) out[h0] = in[h0]; continue; }
        wave_lds_sync();
        for (uint32_t i = lane; i < 256; i += 64) cnt[i] = 0;
        wave_lds_sync();
        for (uint64_t p0 = h0; p0 < h1; p0 += 64) {                       // histogram
            const uint64_t p = p0 + lane;
            if (p < h1) atomicAdd(&cnt[(in[p].qos_flags >> shift) & 0xFFu], 1u);
        }
        wave_lds_sync();
        // exclusive scan of the 256 bucket counts: a lane owns 4 consecutive buckets
        const uint32_t c0 = cnt[lane * 4], c1 = cnt[lane * 4 + 1], c2 = cnt[lane * 4 + 2], c3 = cnt[lane * 4 + 3];
        uint32_t x = c0 + c1 + c2 + c3;
        const uint32_t own = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o, 64); if (int(lane) >= o) x += y; }
        const uint32_t base = x - own;
        cnt[lane * 4] = base; cnt[lane * 4 + 1] = base + c0; cnt[lane * 4 + 2] = base + c0 + c1; cnt[lane * 4 + 3] = base + c0 + c1 + c2;
        wave_lds_sync();
        for (uint64_t p0 = h0; p0 < h1; p0 += 64) {                       // stable scatter
            const uint64_t p = p0 + lane;
            const bool live = p < h1;
            Tuple tp{};
            if (live) tp = in[p];
            const uint32_t d = (tp.qos_flags >> shift) & 0xFFu;
            uint32_t dst = 0;
            unsigned long long rest = __ballot(live);
            while (rest) {                                                // one round per distinct digit present in the chunk
                const int leader = __ffsll(static_cast<long long>(rest)) - 1;
                const uint32_t dl = __shfl(d, leader, 64);
                const unsigned long long m = __ballot(live && d == dl);
                const uint32_t b = cnt[dl];                               // (read by every lane before the leader moves the bucket on)
                wave_lds_sync();
                if (live && d == dl) dst = b + uint32_t(__popcll(m & below));
                if (int(lane) == leader) cnt[dl] = b + uint32_t(__popcll(m));
                wave_lds_sync();
                rest &= ~m;
            }
            if (live) out[h0 + dst] = tp;
        }
    }
}

// Directory of the partitioned window: per topic the number of node groups, then (after the scan) one entry per group.
// is_start(p): first hit of its topic, or another node than its predecessor.
__global__ __launch_bounds__(64 * kNodeWaves) void node_groups_kernel(const Tuple* __restrict__ tuples, const uint64_t* __restrict__ hit_off, uint64_t hit_lo,
                                                                      uint32_t nt, uint32_t* __restrict__ group_cnt, const uint64_t* __restrict__ group_off,
                                                                      uint32_t* __restrict__ group_node, uint64_t* __restrict__ group_begin, uint64_t begin_bias) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t t = blockIdx.x * kNodeWaves + wave; t < nt; t += gridDim.x * kNodeWaves) {
        const uint64_t h0 = hit_off[t] - hit_lo, h1 = hit_off[t + 1] - hit_lo;
        uint32_t groups = 0;
        uint32_t prev = kNone;                                            // node of the position before the chunk
        for (uint64_t p0 = h0; p0 < h1; p0 += 64) {
            const uint64_t p = p0 + lane;
            const bool live = p < h1;
            const uint32_t node = live ? tuples[p].qos_flags >> 16 : 0u;
            uint32_t left = __shfl_up(node, 1, 64);
            if (lane == 0) left = prev;
            const bool start = live && (p == h0 || node != left);
            const unsigned long long m = __ballot(start);
            if (group_node && start) {
                const uint64_t g = group_off[t] + groups + uint32_t(__popcll(m & below));
                group_node[g] = node; group_begin[g] = begin_bias + p;
            }
            groups += uint32_t(__popcll(m));
            prev = __shfl(node, 63, 64);
        }
        if (!group_node && lane == 0) group_cnt[t] = groups;
    }
}

}  // namespace

// ------------------------------------------------------------------------------ launchers
uint32_t expand_tile_hits() { return kTile; }
const char* expand_tuple_kernel_name() { return "expand_kernel"; }

uint32_t scan_block_topics() { return kScanBlock; }

// Lane-per-item kernels whose wave runs as long as its longest item (walk, tokeniser): a batch that cannot fill the chip uses 2^-shift of the lanes, up to
// 2 048 waves.  RGR_WALK_LANE_SHIFT (0 .. 6; read per launch) overrides the choice.
static uint32_t small_batch_lane_shift(uint32_t n, uint64_t lane_budget = 131072ull) {
    const char* e = std::getenv("RGR_WALK_LANE_SHIFT");
    if (e) return uint32_t(std::min(6, std::max(0, std::atoi(e))));
    uint32_t shift = 0;
    while (shift < 6u && (uint64_t(n) << (shift + 1)) <= lane_budget) ++shift;
    return shift;
}
// (the tokeniser's kernels gain less from it — their waves are short either way — and lose a little once the batch has a few hundred waves of its own:
// 300 topics 0.140 -> 0.110 ms of tokenising, 2 600 0.163 -> 0.152, 20 000 0.189 -> 0.202 with the walk's budget: profiles/r07p_*)
constexpr uint64_t kTokLaneBudget = 32768;

void launch_scatter_edges(EdgeEntry* dst, const uint32_t* slots, const EdgeEntry* recs, uint32_t n, void* stream) {
    if (n) scatter_edges_kernel<<<(n + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(dst, slots, recs, n);
}
void launch_scatter_desc(FilterDesc* dst, const uint32_t* fids, const FilterDesc* recs, uint32_t n, void* stream) {
    if (n) scatter_desc_kernel<<<(n + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(dst, fids, recs, n);
}

void launch_tok_count(const uint8_t* blob, const uint64_t* offs, uint32_t n, uint32_t* level_cnt, uint8_t* tflags, void* stream,
                      const uint8_t* force_invalid) {
    if (!n) return;
    const uint32_t sh = small_batch_lane_shift(n, kTokLaneBudget);
    tok_count_kernel<<<uint32_t(((uint64_t(n) << sh) + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(blob, offs, n, level_cnt, tflags, force_invalid, sh);
}
void launch_publish_scan(const uint8_t* pkts, const uint64_t* pkt_offs, uint32_t n, int version, PubInfo* info, uint32_t* topic_len, uint8_t* bad,
                         const uint32_t* from_ids, PublishAttr* attrs, void* stream) {
    if (n) publish_scan_kernel<<<(n + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(pkts, pkt_offs, n, version, info, topic_len, bad, from_ids, attrs);
}
void launch_publish_topics(const uint8_t* pkts, const PubInfo* info, uint32_t n, const uint64_t* topic_offs, uint8_t* blob, void* stream) {
    if (n) publish_topics_kernel<<<(n + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(pkts, info, n, topic_offs, blob);
}

void launch_scan_u32(const uint32_t* in, uint64_t* out, uint32_t n, uint64_t* block_tmp, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t nb = (n + kScanBlock - 1) / kScanBlock;
    if (nb == 0) { (void)hipMemsetAsync(out, 0, 8, s); return; }
    scan1_reduce_kernel<<<nb, kScanThreads, 0, s>>>(in, n, block_tmp);
    scan_spine_kernel<1><<<1, 1024, 0, s>>>(block_tmp, nb);
    scan1_down_kernel<<<nb, kScanThreads, 0, s>>>(in, out, n, block_tmp, nb);
}

void launch_tok_fill(const DictView& d, const uint8_t* blob, const uint64_t* offs, uint32_t n, const uint64_t* tok_off,
                     const uint8_t* tflags, uint32_t* tokens, void* stream) {
    if (!n) return;
    const uint32_t sh = small_batch_lane_shift(n, kTokLaneBudget);
    tok_fill_kernel<<<uint32_t(((uint64_t(n) << sh) + kTokFillThreads - 1) / kTokFillThreads), kTokFillThreads, 0, static_cast<hipStream_t>(stream)>>>(d, blob, offs, n, tok_off, tflags, tokens, sh);
}

void launch_walk(const TrieView& t, const WalkArgs& a, bool overflow_pass, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a.n == 0) return;
    // The overflow pass does not know the overflow count on the host: it is launched over the
    // whole chunk and every block beyond *ovf_count exits at once.
    if (!overflow_pass) {
        // (r7m) A micro-batch — a few thousand publishes of the host router's batcher — was 41 waves of 64 walks: a wave runs as long as its LONGEST walk
        // (every step a dependent 32-byte gather) while 1 000 SIMDs idle beside it.  A chunk that cannot fill the chip spreads its walks over more
        // waves — 2^-shift of the lanes walk, up to 2 048 waves: 2 600 topics in 116 us with 64 walks per wave, 89 with 16, 64 with 4
        // (profiles/r07m_*, r07n_*; small_batch_lane_shift above — the tokeniser's kernels use it too).
        WalkArgs w = a;
        w.lane_shift = small_batch_lane_shift(a.n);
        const uint32_t per_block = uint32_t(kWalkThreads) >> w.lane_shift;
        walk_kernel<false><<<(a.n + per_block - 1) / per_block, kWalkThreads, 0, s>>>(t, w);
    }
    else walk_kernel<true><<<(a.n + kOvfThreads - 1) / kOvfThreads, kOvfThreads, 0, s>>>(t, a);
}

void launch_retain_step(const RetainView& t, const RetainRound& r, void* stream) {
    if (r.m) retain_step_kernel<<<(r.m + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(t, r);
}

void launch_retain_advance(const RetainRound& r, uint32_t n, uint32_t* fdepth, void* stream) {
    if (n) retain_advance_kernel<<<(n + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(r, n, fdepth);
}

void launch_retain_next(const RetainView& t, const RetainRound& r, const uint64_t* out_off, uint32_t* nf_filter, uint32_t* nf_node,
                        uint32_t* big_list, uint32_t* big_count, void* stream) {
    if (!r.m) return;
    hipStream_t s = static_cast<hipStream_t>(stream);
    retain_scatter_kernel<<<(r.m + 255) / 256, 256, 0, s>>>(t, r, out_off, nf_filter, nf_node, big_list, big_count);
    retain_big_kernel<<<1024, 256, 0, s>>>(t, r, out_off, nf_filter, nf_node, big_list, big_count);
}

void launch_retain_emit(const RetainRound& r, const uint64_t* epos, uint64_t g_base, uint32_t* arena, uint64_t* ovf_base,
                        uint64_t* ovf_end, void* stream) {
    if (r.m) retain_emit_kernel<<<(r.m + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(r, epos, g_base, arena, ovf_base, ovf_end);
}

void launch_retain_finish(uint32_t n, const uint64_t* ovf_base, const uint64_t* ovf_end, uint32_t* pair_cnt, void* stream) {
    if (n) retain_finish_kernel<<<(n + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(n, ovf_base, ovf_end, pair_cnt);
}

void launch_pairs_dense(const TrieView& t, const ChunkArrays& c, const uint64_t* off, uint32_t* out, bool reps, void* stream) {
    if (!c.n) return;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (reps) pairs_dense_kernel<true><<<(c.n + 255) / 256, 256, 0, s>>>(t, c, off, out);
    else pairs_dense_kernel<false><<<(c.n + 255) / 256, 256, 0, s>>>(t, c, off, out);
}

// count / compact with their gathers in batches (prep_batched.inc) are the default since r5a: 7.7 -> 6.0 ms of preparation per 10 M topics at
// config 3, ids24 105.9 -> 108.1 M matches/s, packed 88.6 -> 90.1 M, tuples 30.39 -> 30.50 M, all digests equal
// (profiles/r05a_ab_prep.jsonl).  RGR_PREP_BATCH=0 (read per launch) selects the one-gather-at-a-time kernels.
static bool prep_batched() { const char* e = std::getenv("RGR_PREP_BATCH"); return !(e && e[0] == '0'); }

void launch_count(const TrieView& t, const ChunkArrays& c, void* stream) {
    if (c.n == 0) return;
    hipStream_t s = static_cast<hipStream_t>(stream);
    (void)hipMemsetAsync(c.big_count, 0, 4, s);
    if (prep_batched()) count_batched_kernel<<<(c.n + 255) / 256, 256, 0, s>>>(t, c);
    else count_kernel<<<(c.n + 255) / 256, 256, 0, s>>>(t, c);
    count_big_kernel<<<512, 256, 0, s>>>(t, c);
}

void launch_scan(const ChunkArrays& c, uint64_t* block_tmp, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t nb = (c.n + kScanBlock - 1) / kScanBlock;
    if (nb == 0) return;
    scan_reduce_kernel<<<nb, kScanThreads, 0, s>>>(c, block_tmp);
    scan_spine_kernel<2><<<1, 1024, 0, s>>>(block_tmp, nb);
    scan_down_kernel<<<nb, kScanThreads, 0, s>>>(c, block_tmp, nb);
}

void launch_compact(const TrieView& t, const ChunkArrays& c, uint32_t topic_base, void* stream) {
    if (c.n == 0) return;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (prep_batched()) compact_batched_kernel<<<(c.n + kCompactWave - 1) / kCompactWave, kCompactWave, 0, s>>>(t, c, topic_base);
    else compact_kernel<<<(c.n + kCompactWave - 1) / kCompactWave, kCompactWave, 0, s>>>(t, c, topic_base);
    compact_big_kernel<<<512, 256, 0, s>>>(t, c, topic_base);
}

void launch_tiles(const ChunkArrays& c, uint64_t pair_lo, uint64_t pair_hi, uint64_t hit_lo, TileRec* tile_first, void* stream) {
    if (pair_hi <= pair_lo) return;
    const uint64_t np = pair_hi - pair_lo;
    tiles_kernel<<<uint32_t((np + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(c, pair_lo, pair_hi, hit_lo, tile_first);
}

void launch_expand(const TrieView& t, const ChunkArrays& c, uint64_t pair_lo, uint64_t pair_hi, uint64_t hit_lo, uint64_t hit_hi,
                   const TileRec* tile_first, Tuple* out, void* stream, const DeliverArgs* deliver, bool hits8, int id_source) {
    if (hit_hi <= hit_lo) return;
    const uint32_t ntiles = uint32_t((hit_hi - hit_lo + kTile - 1) / kTile);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (deliver && hits8) {          // RGR_FORMAT_DELIVER8 (r6): the lean expansion writing {sub_id, word} per hit
        const char* g = std::getenv("RGR_DELIVER_LEAN");
        if (g && g[0] == '2') expand_deliver_lean_kernel<kTile / 8, 8, true><<<ntiles, kTile / 8, 0, s>>>(t.subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tile_first, ntiles, out, *deliver);
        else expand_deliver_lean_kernel<kTile / 4, 4, true><<<ntiles, kTile / 4, 0, s>>>(t.subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tile_first, ntiles, out, *deliver);
        return;
    }
    // the plain kernel runs 1024 x 2 (with the single-run fast path: +3 % over 512 x 4, profiles/r02f_sweep_*); the delivery
    // variant keeps 512 x 4 — its per-wave candidate bookkeeping was 19 % slower at 1024 x 2 (profiles/r02g_bench_config3_deliver_*)
    // the delivery variant with its loads issued early (expand_tuple.inc) is the default since r5a: 0.876 -> 0.860 ms per 2^28-hit window,
    // 13.82 -> 14.03 M matches/s, whole-window delivery parity ok (profiles/r05a_ab_deliver.jsonl).  RGR_DELIVER_EARLY=0 (read per
    // launch) selects expand_kernel<true>.
    const char* early = deliver ? std::getenv("RGR_DELIVER_EARLY") : nullptr;
    // r5g-r5i: the variant that compacts every wave's v5 hits before it runs the v5 path ONCE per wave (expand_deliver_lean_kernel,
    // expand_tuple.inc) is the default: 0.855 -> 0.727 ms per 2^28-hit window, delivery stage 14.59 -> 16.14 M matches/s on one table,
    // whole-window delivery parity ok (profiles/r05i_ab_deliver_lean_512x4_vs_256x8_and_2e30_windows.jsonl; SQ counters before / after:
    // r05c_*, r05h_*: 385 -> 228 vector and 253 -> 169 scalar instructions per wave).  256 threads x 8 positions (RGR_DELIVER_LEAN=2)
    // measures the same (0.736); RGR_DELIVER_LEAN=0 (read per launch) selects the kernels above.
    const char* lean = deliver ? std::getenv("RGR_DELIVER_LEAN") : nullptr;
    if (deliver && lean && lean[0] == '2') expand_deliver_lean_kernel<kTile / 8, 8><<<ntiles, kTile / 8, 0, s>>>(t.subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tile_first, ntiles, out, *deliver);
    else if (deliver && !(lean && lean[0] == '0')) expand_deliver_lean_kernel<kTile / 4, 4><<<ntiles, kTile / 4, 0, s>>>(t.subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tile_first, ntiles, out, *deliver);
    else if (deliver && !(early && early[0] == '0')) expand_deliver_early_kernel<kTile / 4, 4><<<ntiles, kTile / 4, 0, s>>>(t.subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tile_first, ntiles, out, *deliver);
    else if (deliver) expand_kernel<true, kTile / 4, 4><<<ntiles, kTile / 4, 0, s>>>(t.subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tile_first, ntiles, out, *deliver);
    else if (id_source == 2) expand_kernel<false, kExpandThreads, kExpandPerThread, 2><<<ntiles, kExpandThreads, 0, s>>>(t.subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tile_first, ntiles, out, DeliverArgs{}, nullptr);
    else if (id_source == 1 && t.subs_packed) expand_kernel<false, kExpandThreads, kExpandPerThread, 1><<<ntiles, kExpandThreads, 0, s>>>(t.subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tile_first, ntiles, out, DeliverArgs{}, t.subs_packed);
    else expand_kernel<false, kExpandThreads, kExpandPerThread><<<ntiles, kExpandThreads, 0, s>>>(t.subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tile_first, ntiles, out, DeliverArgs{});
}

// Tiles per block of the lane-held compact expansion (expand_compact_lp_kernel); 0 = the tile-per-block kernel.  Measured at config 3
// (profiles/r04n_ab_lane_held_ids24_packed.jsonl, one table, full passes): IDS24 98.97 M matches/s with the tile-per-block kernel,
// 105.6 / 104.1 / 105.0 M with 1 / 2 / 4 tiles per block lane-held (all 148 150 579 430 hits digested equal); PACKED 89.4 M against
// 85.4 / 83.0 / 81.0 M — its 16-byte stores already run at the store stream's rate, the search through ds_bpermute only adds to it.
// So IDS24 takes the lane-held kernel with one tile per block, PACKED stays.  RGR_COMPACT_LP (0, 1, 2, 4; read per launch) overrides
// both: the A/B switch of bench.py --ab-env and of the tests.
#ifndef RGR_COMPACT_LP_IDS24
#define RGR_COMPACT_LP_IDS24 1
#endif
#ifndef RGR_COMPACT_LP_PACKED
#define RGR_COMPACT_LP_PACKED 0
#endif
#define RGR_STR2(x) #x
#define RGR_STR(x) RGR_STR2(x)
#define RGR_COMPACT_LP_IDS24_NAME (RGR_COMPACT_LP_IDS24 ? "expand_compact_lp_kernel<ids24, " RGR_STR(RGR_COMPACT_LP_IDS24) ">" : "expand_compact_kernel<ids24>")
const char* expand_ids24_kernel_name() { return RGR_COMPACT_LP_IDS24_NAME; }
int compact_lp_tiles(int format) {
    const char* e = std::getenv("RGR_COMPACT_LP");
    const int v = e ? std::atoi(e) : format == kFmtIds24 ? RGR_COMPACT_LP_IDS24 : RGR_COMPACT_LP_PACKED;
    return v <= 0 ? 0 : v == 1 ? 1 : v < 4 ? 2 : 4;
}

bool launch_expand_compact(const TrieView& t, const ChunkArrays& c, uint64_t pair_lo, uint64_t pair_hi, uint64_t hit_lo, uint64_t hit_hi,
                           const TileRec* tile_first, int format, uint32_t* out_ids, uint8_t* out_qos, void* stream, const NextTiles* next) {
    if (hit_hi <= hit_lo) return false;
    const uint32_t ntiles = uint32_t((hit_hi - hit_lo + kTile - 1) / kTile);
    hipStream_t s = static_cast<hipStream_t>(stream);
    static const bool no_packed_reads = std::getenv("RGR_NO_PACKED_READS") != nullptr;       // A/B switch: 8-byte entry loads as in r3
    const uint32_t* pk = no_packed_reads ? nullptr : t.subs_packed;
    const uint32_t nb1 = (ntiles + kCompactTiles - 1) / kCompactTiles, nb24 = (ntiles + kIds24Tiles - 1) / kIds24Tiles;
    // RGR_COMPACT_LP=T (A/B switch, read per launch): PACKED / IDS24 through expand_compact_lp_kernel with T tiles per block (pairs held in
    // lanes, expand_compact.inc) instead of the tile-per-block kernel; needs the packed side array
    // RGR_IDS24_X4=1 (A/B switch, read per launch; not measured yet): IDS24 through 16-byte stores, two lane-held tiles per block
    if (format == kFmtIds24 && pk) {
        const char* x4 = std::getenv("RGR_IDS24_X4");
        if (x4 && x4[0] == '1') {
            expand_ids24_x4_kernel<<<(ntiles + 1) / 2, kCompactThreads, 0, s>>>(t.subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tile_first, ntiles, out_ids, out_qos, pk);
            return false;
        }
    }
    const int lp = compact_lp_tiles(format);
    if (next && next->out && lp == 1 && pk && format == kFmtIds24 && next->pair_hi > next->pair_lo) {
        // RGR_TILES_FUSED: the lane-held IDS24 expansion (one tile per block) followed, in the same grid, by the blocks that write the
        // NEXT window's tile records
        const uint32_t nb_tiles = uint32_t((next->pair_hi - next->pair_lo + kCompactThreads - 1) / kCompactThreads);
        expand_ids24_lp_tiles_kernel<<<ntiles + nb_tiles, kCompactThreads, 0, s>>>(t.subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tile_first, ntiles, out_ids, out_qos, pk, *next);
        return true;
    }
    if (lp && pk && (format == kFmtIds24 || format == kFmtPacked)) {
        const uint32_t nb = (ntiles + uint32_t(lp) - 1) / uint32_t(lp);
#define RGR_LP_LAUNCH(F, TT) expand_compact_lp_kernel<F, TT><<<nb, kCompactThreads, 0, s>>>(t.subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tile_first, ntiles, out_ids, out_qos, pk)
        if (format == kFmtIds24) { if (lp == 1) RGR_LP_LAUNCH(kFmtIds24, 1); else if (lp == 2) RGR_LP_LAUNCH(kFmtIds24, 2); else RGR_LP_LAUNCH(kFmtIds24, 4); }
        else { if (lp == 1) RGR_LP_LAUNCH(kFmtPacked, 1); else if (lp == 2) RGR_LP_LAUNCH(kFmtPacked, 2); else RGR_LP_LAUNCH(kFmtPacked, 4); }
#undef RGR_LP_LAUNCH
        return false;
    }
    if (format == kFmtIds24) expand_compact_kernel<kFmtIds24, kIds24Tiles><<<nb24, kCompactThreads, 0, s>>>(t.subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tile_first, ntiles, out_ids, out_qos, pk);
    else if (format == kFmtPacked) expand_compact_kernel<kFmtPacked, kCompactTiles><<<nb1, kCompactThreads, 0, s>>>(t.subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tile_first, ntiles, out_ids, out_qos, pk);
    else expand_compact_kernel<kFmtSoa, kCompactTiles><<<nb1, kCompactThreads, 0, s>>>(t.subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tile_first, ntiles, out_ids, out_qos, nullptr);
    return false;
}

void launch_pack_subs(const SubEntry* subs, uint64_t n, uint32_t* packed, void* stream) {
    if (n) pack_subs_kernel<<<uint32_t((n + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(subs, n, packed);
}

void launch_pack_runs(const uint32_t* src, const uint32_t* topic, const uint64_t* off, uint64_t n, uint32_t shard, RunDesc* out, void* stream) {
    if (n) pack_runs_kernel<<<uint32_t((n + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(src, topic, off, n, shard, out);
}

void launch_node_partition(const Tuple* in, Tuple* out, const uint64_t* hit_off, uint64_t hit_lo, uint32_t nt, uint32_t shift, void* stream) {
    if (!nt) return;
    const uint32_t blocks = std::min<uint32_t>((nt + kNodeWaves - 1) / kNodeWaves, 8192);
    node_partition_kernel<<<blocks, 64 * kNodeWaves, 0, static_cast<hipStream_t>(stream)>>>(in, out, hit_off, hit_lo, nt, shift);
}
void launch_node_groups(const Tuple* tuples, const uint64_t* hit_off, uint64_t hit_lo, uint32_t nt, uint32_t* group_cnt, const uint64_t* group_off,
                        uint32_t* group_node, uint64_t* group_begin, uint64_t begin_bias, void* stream) {
    if (!nt) return;
    const uint32_t blocks = std::min<uint32_t>((nt + kNodeWaves - 1) / kNodeWaves, 8192);
    node_groups_kernel<<<blocks, 64 * kNodeWaves, 0, static_cast<hipStream_t>(stream)>>>(tuples, hit_off, hit_lo, nt, group_cnt, group_off, group_node, group_begin,
                                                                                        begin_bias);
}

uint32_t dedup_topic_cap() { return kDedupTopicCap; }
uint32_t dedup_stat_slots() { return 2048; }

void launch_dedup(const Cand* cand, const uint32_t* tile_ncand, const uint32_t* tile_trange, uint32_t ntiles, HitWords tuples, uint32_t nt,
                  const uint64_t* hit_off, uint64_t hit_lo, DedupItem* items, uint32_t* item_counts, uint32_t parity, unsigned long long* stat, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!ntiles) { (void)hipMemsetAsync(item_counts + (parity ^ 1u), 0, 4, s); return; }      // (the next window's counter: what this window's launch would have zeroed)
    const uint32_t slots = uint32_t(kDedupTopicSlots);
    // tile pass + classification in one launch (dedup.inc); item_counts[parity] is this window's item counter
#ifdef RGR_DIAG_TILE_BLOCKS_PER        /* diagnostic builds (r5x): tiles per block of the tile pass instead of a fixed grid of 2 048 blocks */
    const uint32_t tile_blocks = std::min<uint32_t>(dedup_stat_slots(), std::max<uint32_t>(1u, (ntiles + RGR_DIAG_TILE_BLOCKS_PER - 1) / RGR_DIAG_TILE_BLOCKS_PER));
#else
    const uint32_t tile_blocks = std::min<uint32_t>(ntiles, 2048u);
#endif
    // RGR_DEDUP_TEST_SLOTS (tests only): a smaller table, so that parts overflow and the re-split path runs on ordinary inputs
    // (read on every launch — it is one getenv — so that a test can set it after other tests of the same process have launched)
    const uint32_t max_slots = [&] {
        const char* e = std::getenv("RGR_DEDUP_TEST_SLOTS");
        uint32_t v = e ? uint32_t(std::atoi(e)) : 0u, p2 = 64;
        while (p2 < v && p2 < slots) p2 <<= 1;
        return v ? p2 : slots;
    }();
    // RGR_DEDUP_SLOT_FACTOR (A/B, r6t): table slots per candidate of a part (2: load factor 0.25-0.5; 4: 0.125-0.25, capped by the 32 KiB table)
    const char* sf = std::getenv("RGR_DEDUP_SLOT_FACTOR");
    const uint32_t slot_factor = sf && std::atoi(sf) >= 2 && std::atoi(sf) <= 16 ? uint32_t(std::atoi(sf)) : 2u;
    dedup_tile_kernel<<<tile_blocks + (nt + 255) / 256, 256, 0, s>>>(cand, tile_ncand, tile_trange, ntiles, hit_off, hit_lo, nt, tuples, stat, tile_blocks, items, item_counts, parity, max_slots, slot_factor);
    uint32_t* item_count = item_counts + parity;
    // the item count stays on the device: a fixed grid of persistent blocks (4 per CU fit) strides over the items.
    // RGR_DEDUP_PROBE=0 (A/B switch, read per launch): linear probing and 8-byte table clears, the topic pass as it was until r5f
    // (0.396 -> 0.358 ms per 2^28-hit window with double hashing + 16-byte clears; template parameter — as a kernel argument the same
    // instructions cost 0.515 ms)
    const char* pe = std::getenv("RGR_DEDUP_PROBE");
    const uint32_t grid = kDedupTopicThreads >= 512 ? 1024 : 1280;
    // RGR_DEDUP_PROBE=3: the lists read 64 entries at a time, as until r6s (=7, the default: the first 256 entries of a tile's list in one request,
    // 0.1404 -> 0.1289 ms of dedup per window, profiles/r06s_*)
    if (pe && pe[0] == '0') dedup_topic_kernel<0><<<grid, kDedupTopicThreads, 0, s>>>(cand, tile_ncand, items, item_count, tuples);
    else if (pe && pe[0] == '3') dedup_topic_kernel<3><<<grid, kDedupTopicThreads, 0, s>>>(cand, tile_ncand, items, item_count, tuples);
    else dedup_topic_kernel<7><<<grid, kDedupTopicThreads, 0, s>>>(cand, tile_ncand, items, item_count, tuples);
}

}  // namespace rgr
