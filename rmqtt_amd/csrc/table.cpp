// Host-side subscription table compiler — see table.hpp.
#include "table.hpp"

#include <algorithm>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <sys/mman.h>
#include <thread>
#include <unordered_set>

#include "rmqtt_gpu_router.h"
#include "topic.hpp"

namespace rgr {

// ------------------------------------------------------------------ FlatArray helpers
void* flat_alloc(size_t bytes) {
    void* mem = nullptr;
    if (posix_memalign(&mem, bytes >= (2u << 20) ? (2u << 20) : 64, bytes) != 0) throw std::bad_alloc();
#ifdef MADV_HUGEPAGE
    if (bytes >= (2u << 20)) (void)madvise(mem, bytes, MADV_HUGEPAGE);   // fewer first-touch faults
#endif
    return mem;
}

unsigned flat_fill_threads(uint64_t n) {
    return unsigned(std::min<uint64_t>({uint64_t(std::max(1u, std::thread::hardware_concurrency())), n / (1u << 20), uint64_t(64)}));
}

// ------------------------------------------------------------------ StringDict
StringDict::StringDict() : slots_(1024, 0), mask_(1023) {}

uint64_t StringDict::hash(std::string_view s) {
    uint64_t h = 0xcbf29ce484222325ull;             // FNV-1a 64 + avalanche
    for (unsigned char c : s) { h ^= c; h *= 0x100000001b3ull; }
    h ^= h >> 32; h *= 0xd6e8feb86659fd93ull; h ^= h >> 32;
    return h;
}

uint32_t StringDict::find(std::string_view s) const {
    const uint64_t h = hash(s);
    for (uint64_t i = h & mask_;; i = (i + 1) & mask_) {
        const uint32_t v = slots_[i];
        if (!v) return kTokUnknown;
        const Entry& e = entries_[v - 1];
        if (e.hash == h && e.len == s.size() && (e.len == 0 || std::memcmp(arena_.data() + e.off, s.data(), e.len) == 0))
            return v - 1 + kTokFirst;
    }
}

void StringDict::grow() {
    std::vector<uint32_t> ns(slots_.size() * 2, 0);
    const uint64_t nm = ns.size() - 1;
    for (uint32_t v : slots_) {
        if (!v) continue;
        uint64_t i = entries_[v - 1].hash & nm;
        while (ns[i]) i = (i + 1) & nm;
        ns[i] = v;
    }
    slots_.swap(ns);
    mask_ = nm;
}

uint32_t StringDict::intern(std::string_view s) {
    uint32_t t = find(s);
    if (t != kTokUnknown) return t;
    if ((entries_.size() + 1) * 2 > slots_.size()) grow();
    const uint64_t h = hash(s);
    entries_.push_back(Entry{h, arena_.size(), uint32_t(s.size()), 0});
    arena_.insert(arena_.end(), s.begin(), s.end());
    uint64_t i = h & mask_;
    while (slots_[i]) i = (i + 1) & mask_;
    slots_[i] = uint32_t(entries_.size());
    return uint32_t(entries_.size()) - 1 + kTokFirst;
}

std::string_view StringDict::str(uint32_t tok) const {
    const Entry& e = entries_[tok - kTokFirst];
    return std::string_view(arena_.data() + e.off, e.len);
}

// ------------------------------------------------------------------ HostTable
static EdgeEntry empty_edge() { return EdgeEntry{kEdgeEmpty, 0, kNone, kNone, kNone, kNone, 0, 0}; }

HostTable::HostTable() {
    nodes_.push_back(Node{kNone, 0, kNone, 0, kNone, kNone, kNone});   // root = node 0
    edges_.assign(1024, empty_edge());
}

void HostTable::reserve(uint64_t n_filters_hint, uint64_t levels_hint) {
    uint64_t want = 1024;
    while (want < levels_hint * 2) want <<= 1;
    if (want > edges_.size()) rehash(want);
    // geometric: an exact reserve() reallocates (and moves tens of millions of records) on EVERY small bulk call
    // into a big table — 250 ms per SUBSCRIBE burst at 10 M subscriptions
    auto grow = [](auto& v, uint64_t need) { if (need > v.capacity()) v.reserve(std::max<uint64_t>(need, v.capacity() + v.capacity() / 2)); };
    grow(filters_, n_filters_hint);
    grow(nodes_, levels_hint + 1);
}

bool HostTable::tokenize_filter(std::string_view f, std::vector<uint32_t>& toks) {
    toks.clear();
    // validate first so that rejected filters do not pollute the dictionary
    int64_t n = for_each_level(f, [](int64_t, std::string_view, LevelKind) {});
    if (n < 0) return false;
    for_each_level(f, [&](int64_t, std::string_view seg, LevelKind k) {
        if (k == LevelKind::Plus) toks.push_back(kTokPlus);
        else if (k == LevelKind::Hash) toks.push_back(kTokHash);
        else toks.push_back(dict_.intern(seg));
    });
    return true;
}

uint8_t HostTable::tokenize_topic(std::string_view t, std::vector<uint32_t>& toks) const {
    const size_t mark = toks.size();
    uint8_t flags = 0;
    int64_t n = for_each_level(t, [&](int64_t idx, std::string_view seg, LevelKind k) {
        if (idx == 0 && k == LevelKind::Metadata) flags |= kTopicMeta;
        if (k == LevelKind::Plus) toks.push_back(kTokPlus);
        else if (k == LevelKind::Hash) toks.push_back(kTokHash);
        else toks.push_back(dict_.find(seg));
    });
    if (n < 0) { toks.resize(mark); return kTopicInvalid; }
    return flags;
}

uint32_t HostTable::find_slot(uint32_t parent, uint32_t token) const {
    const uint32_t mask = uint32_t(edges_.size() - 1);
    for (uint32_t i = edge_hash(parent, token) & mask;; i = (i + 1) & mask) {
        const EdgeEntry& e = edges_[i];
        if (e.parent == kEdgeEmpty) return kNone;
        if (e.parent == parent && e.token == token) return i;
    }
}

void HostTable::rehash(uint64_t new_cap) {
    EdgeArray old;
    old.swap(edges_);
    edges_.assign(new_cap, empty_edge());
    const uint32_t mask = uint32_t(new_cap - 1);
    for (const EdgeEntry& e : old) {
        if (e.parent == kEdgeEmpty || e.parent == kEdgeTomb) continue;
        uint32_t i = edge_hash(e.parent, e.token) & mask;
        while (edges_[i].parent != kEdgeEmpty) i = (i + 1) & mask;
        edges_[i] = e;
        nodes_[e.child].slot = i;
    }
    edge_used_ = edge_live_;
    delta_.relocated = true;
    // slots moved: re-point every header's plus_slot; the child-token bitmaps are rebuilt from the live edges (drops stale bits)
    for (EdgeEntry& e : edges_) {
        if (e.parent == kEdgeEmpty) continue;
        const uint32_t pc = nodes_[e.child].plus_child;
        e.plus_slot = pc == kNone ? kNone : nodes_[pc].slot;
    }
    const uint32_t rp = nodes_[0].plus_child;
    root_hdr_.plus_slot = rp == kNone ? kNone : nodes_[rp].slot;
    rebuild_child_bitmaps();
}

// The 64-bit child-token bitmap of every node header, exactly, from the live literal edges.
void HostTable::rebuild_child_bitmaps() {
    for (EdgeEntry& e : edges_) if (e.parent != kEdgeEmpty) e.lit_lo = e.lit_hi = 0;
    root_hdr_.lit_lo = root_hdr_.lit_hi = 0;
    for (const EdgeEntry& e : edges_) {
        if (e.parent == kEdgeEmpty || e.parent == kEdgeTomb || e.token < kTokFirst) continue;
        const uint32_t b = lit_bit(e.token), m = 1u << (b & 31u);
        if (e.parent == 0) ((b & 32u) ? root_hdr_.lit_hi : root_hdr_.lit_lo) |= m;
        else { EdgeEntry& p = edges_[nodes_[e.parent].slot]; ((b & 32u) ? p.lit_hi : p.lit_lo) |= m; }
    }
}

uint32_t HostTable::insert_edge(uint32_t parent, uint32_t token, uint32_t child) {
    if ((edge_used_ + 1) * 2 > edges_.size())
        rehash(edge_live_ * 4 > edges_.size() ? edges_.size() * 2 : edges_.size());
    const uint32_t mask = uint32_t(edges_.size() - 1);
    uint32_t i = edge_hash(parent, token) & mask;
    while (edges_[i].parent != kEdgeEmpty && edges_[i].parent != kEdgeTomb) i = (i + 1) & mask;
    if (edges_[i].parent == kEdgeEmpty) edge_used_++;
    edges_[i] = EdgeEntry{parent, token, child, kNone, kNone, kNone, 0, 0};
    edge_live_++;
    touch(i);
    return i;
}

uint32_t HostTable::new_node(uint32_t parent, uint32_t token) {
    uint32_t id;
    if (!free_nodes_.empty()) { id = free_nodes_.back(); free_nodes_.pop_back(); }
    else { id = uint32_t(nodes_.size()); nodes_.push_back(Node{}); }
    nodes_[id] = Node{parent, token, kNone, 0, kNone, kNone, kNone};
    const uint32_t slot = insert_edge(parent, token, id);   // may rehash (fixes every other slot)
    nodes_[id].slot = slot;
    nodes_[parent].nchild++;
    n_nodes_++;
    if (token == kTokPlus) { nodes_[parent].plus_child = id; set_plus_slot(parent, slot); }
    else if (token == kTokHash) nodes_[parent].hash_child = id;
    else literal_edge(parent, token, +1);
    return id;
}

void HostTable::set_plus_slot(uint32_t node, uint32_t slot) {
    if (node == 0) root_hdr_.plus_slot = slot; else { edges_[nodes_[node].slot].plus_slot = slot; touch(nodes_[node].slot); }
}
void HostTable::set_hash_fid(uint32_t node, uint32_t fid) {
    if (node == 0) root_hdr_.hash_fid = fid; else { edges_[nodes_[node].slot].hash_fid = fid; touch(nodes_[node].slot); }
}
// Child-token bitmap of `node`'s header.  A new literal edge sets its bit; a removed one leaves it (another child may share
// the bit, and a stale bit only costs the walk one probe) — rehash() and the bulk build recompute the bitmaps exactly.
void HostTable::literal_edge(uint32_t node, uint32_t token, int delta) {
    if (delta <= 0) return;
    const uint32_t b = lit_bit(token);
    uint32_t& w = node == 0 ? ((b & 32u) ? root_hdr_.lit_hi : root_hdr_.lit_lo)
                            : ((b & 32u) ? edges_[nodes_[node].slot].lit_hi : edges_[nodes_[node].slot].lit_lo);
    const uint32_t m = 1u << (b & 31u);
    if (w & m) return;
    w |= m;
    if (node != 0) touch(nodes_[node].slot);
}
void HostTable::set_term_fid(uint32_t node, uint32_t fid) {
    nodes_[node].term_fid = fid;
    if (node == 0) root_hdr_.term_fid = fid; else { edges_[nodes_[node].slot].term_fid = fid; touch(nodes_[node].slot); }
}
void HostTable::take_delta(Delta& d) {
    d.slots.swap(delta_.slots);
    d.fids.swap(delta_.fids);
    d.relocated = delta_.relocated;
    delta_ = Delta{};
}

uint32_t HostTable::walk_existing(const std::vector<uint32_t>& toks) const {
    uint32_t cur = 0;
    for (uint32_t t : toks) {
        if (t == kTokUnknown) return kNone;
        const uint32_t s = find_slot(cur, t);
        if (s == kNone) return kNone;
        cur = edges_[s].child;
    }
    return cur;
}

int32_t HostTable::filter_add(std::string_view f, uint32_t* fid) {
    std::vector<uint32_t> toks;
    if (!tokenize_filter(f, toks)) return RGR_EINVAL_TOPIC;
    uint32_t cur = 0;
    for (uint32_t t : toks) {
        const uint32_t s = find_slot(cur, t);
        cur = s != kNone ? edges_[s].child : new_node(cur, t);
    }
    if (nodes_[cur].term_fid != kNone) { *fid = nodes_[cur].term_fid; return RGR_OK; }
    uint32_t id;
    if (!free_fids_.empty()) { id = free_fids_.back(); free_fids_.pop_back(); }
    else { id = uint32_t(filters_.size()); filters_.emplace_back(); }
    filters_[id].node = cur;
    filters_[id].subs.clear();
    delta_.fids.push_back(id);
    set_term_fid(cur, id);
    if (nodes_[cur].token == kTokHash) set_hash_fid(nodes_[cur].parent, id);
    n_filters_++;
    *fid = id;
    return RGR_OK;
}

int32_t HostTable::filter_find(std::string_view f, uint32_t* fid) const {
    std::vector<uint32_t> toks;
    int64_t n = for_each_level(f, [&](int64_t, std::string_view seg, LevelKind k) {
        if (k == LevelKind::Plus) toks.push_back(kTokPlus);
        else if (k == LevelKind::Hash) toks.push_back(kTokHash);
        else toks.push_back(dict_.find(seg));
    });
    if (n < 0) return RGR_EINVAL_TOPIC;
    const uint32_t node = walk_existing(toks);
    if (node == kNone || nodes_[node].term_fid == kNone) return RGR_ENOENT;
    *fid = nodes_[node].term_fid;
    return RGR_OK;
}

int32_t HostTable::filter_remove(uint32_t fid) {
    if (fid >= filters_.size() || filters_[fid].node == kNone) return RGR_ENOENT;
    if (!filters_[fid].subs.empty()) return RGR_ESTATE;
    uint32_t t = filters_[fid].node;
    set_term_fid(t, kNone);
    if (nodes_[t].token == kTokHash) set_hash_fid(nodes_[t].parent, kNone);
    // trie.rs:141-143: prune children left with no values and no branches, bottom-up
    while (t != 0 && nodes_[t].term_fid == kNone && nodes_[t].nchild == 0) {
        const uint32_t p = nodes_[t].parent;
        edges_[nodes_[t].slot].parent = kEdgeTomb;
        touch(nodes_[t].slot);
        edge_live_--;
        if (nodes_[t].token == kTokPlus) { nodes_[p].plus_child = kNone; set_plus_slot(p, kNone); }
        else if (nodes_[t].token == kTokHash) nodes_[p].hash_child = kNone;
        else literal_edge(p, nodes_[t].token, -1);
        nodes_[p].nchild--;
        free_nodes_.push_back(t);
        n_nodes_--;
        t = p;
    }
    filters_[fid].node = kNone;
    delta_.fids.push_back(fid);
    std::vector<SubEntry>().swap(filters_[fid].subs);
    free_fids_.push_back(fid);
    n_filters_--;
    return RGR_OK;
}

int32_t HostTable::sub_add(uint32_t fid, uint32_t sub_id, uint8_t qos, uint8_t flags, uint16_t node_idx) {
    if (fid >= filters_.size() || filters_[fid].node == kNone) return RGR_ENOENT;
    auto& v = filters_[fid].subs;
    delta_.fids.push_back(fid);
    const SubEntry e{sub_id, uint32_t(qos) | (uint32_t(flags) << 8) | (uint32_t(node_idx) << 16)};
    const uint64_t is_v5 = (flags & kSubV5) ? 1 : 0;
    if (sub_id > max_sub_id_) max_sub_id_ = sub_id;
    if (node_idx > max_node_idx_) max_node_idx_ = node_idx;
    if (v.empty() || v.back().sub_id < sub_id) { v.push_back(e); n_subs_++; n_v5_ += is_v5; return RGR_OK; }
    auto it = std::lower_bound(v.begin(), v.end(), sub_id, [](const SubEntry& a, uint32_t b) { return a.sub_id < b; });
    if (it != v.end() && it->sub_id == sub_id) {                             // re-subscribe: options replaced
        n_v5_ += is_v5;
        n_v5_ -= ((it->qos_flags >> 8) & kSubV5) ? 1 : 0;
        *it = e;
        return RGR_OK;
    }
    v.insert(it, e);
    n_subs_++;
    n_v5_ += is_v5;
    return RGR_OK;
}

void HostTable::sub_set_attr(uint32_t sub_id, uint32_t owner_id, uint32_t client_idx) {
    if (sub_id >= attrs_.size()) attrs_.resize(size_t(sub_id) + 1 + attrs_.size() / 2, SubAttr{kNone, kNone});
    attrs_[sub_id] = SubAttr{owner_id, client_idx};
    has_attrs_ = true;
}

int32_t HostTable::sub_remove(uint32_t fid, uint32_t sub_id) {
    if (fid >= filters_.size() || filters_[fid].node == kNone) return RGR_ENOENT;
    auto& v = filters_[fid].subs;
    auto it = std::lower_bound(v.begin(), v.end(), sub_id, [](const SubEntry& a, uint32_t b) { return a.sub_id < b; });
    if (it == v.end() || it->sub_id != sub_id) return RGR_ENOENT;
    n_v5_ -= ((it->qos_flags >> 8) & kSubV5) ? 1 : 0;
    v.erase(it);
    delta_.fids.push_back(fid);
    n_subs_--;
    return RGR_OK;
}

void HostTable::subscribe_bulk(const uint8_t* blob, const uint64_t* offs, uint64_t n, const uint32_t* sub_ids, const uint8_t* qos,
                               const uint8_t* flags, uint32_t* fids_out, uint64_t* n_rejected, unsigned threads) {
    if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
    threads = unsigned(std::min<uint64_t>(threads, std::max<uint64_t>(1, n / 8192)));
    const bool prof = std::getenv("RGR_BULK_PROFILE") != nullptr;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = tnow();
    auto lap = [&](const char* what) { if (prof) { const double t = tnow(); std::fprintf(stderr, "[bulk] %-12s %.3f s\n", what, t - t_prev); t_prev = t; } };
    auto sv = [&](uint64_t i) { return std::string_view(reinterpret_cast<const char*>(blob) + offs[i], offs[i + 1] - offs[i]); };
    auto run = [&](auto&& fn) {
        if (threads == 1) { fn(0u); return; }
        std::vector<std::thread> th;
        for (unsigned k = 0; k < threads; ++k) th.emplace_back(fn, k);
        for (auto& t : th) t.join();
    };
    // ---- A: parallel parse against the current dictionary; unknown level strings are collected
    struct Part { std::vector<uint32_t> toks; std::vector<uint32_t> lens; std::vector<std::pair<uint64_t, std::string_view>> fix;
                  std::unordered_set<std::string_view> fresh; };
    std::vector<Part> parts(threads);
    run([&](unsigned k) {
        Part& p = parts[k];
        const uint64_t lo = n * k / threads, hi = n * (k + 1) / threads;
        p.lens.reserve(hi - lo); p.toks.reserve((hi - lo) * 9);
        for (uint64_t i = lo; i < hi; ++i) {
            const size_t mark = p.toks.size();
            int64_t L = for_each_level(sv(i), [&](int64_t, std::string_view seg, LevelKind kd) {
                if (kd == LevelKind::Plus) p.toks.push_back(kTokPlus);
                else if (kd == LevelKind::Hash) p.toks.push_back(kTokHash);
                else {
                    const uint32_t t = dict_.find(seg);
                    if (t == kTokUnknown) { p.fix.emplace_back(p.toks.size(), seg); p.fresh.insert(seg); }
                    p.toks.push_back(t);
                }
            });
            if (L < 0) {                                    // invalid filter: drop its partial tokens / fix-ups
                while (!p.fix.empty() && p.fix.back().first >= mark) p.fix.pop_back();
                p.toks.resize(mark);
                p.lens.push_back(kNone);
            } else p.lens.push_back(uint32_t(L));
        }
    });
    // ---- B: intern the new level strings (single thread), C: patch the unknown tokens
    for (auto& p : parts) for (auto& s : p.fresh) dict_.intern(s);
    run([&](unsigned k) { for (auto& f : parts[k].fix) parts[k].toks[f.first] = dict_.find(f.second); });
    lap("tokenise");
    // flat token arrays
    std::vector<uint64_t> toff(n + 1, 0);
    std::vector<uint8_t> valid(n, 1);
    uint64_t rejected = 0;
    {
        uint64_t i = 0;
        for (auto& p : parts) for (uint32_t L : p.lens) { valid[i] = L != kNone; rejected += L == kNone; toff[i + 1] = toff[i] + (L == kNone ? 0 : L); ++i; }
    }
    std::vector<uint32_t> toks(toff[n]);
    {
        uint64_t at = 0;
        for (auto& p : parts) { std::copy(p.toks.begin(), p.toks.end(), toks.begin() + at); at += p.toks.size(); std::vector<uint32_t>().swap(p.toks); }
    }
    // ---- order: lexicographic by token sequence (ties by arrival index: stable ids)
    std::vector<uint32_t> idx;
    idx.reserve(n - rejected);
    for (uint64_t i = 0; i < n; ++i) if (valid[i]) idx.push_back(uint32_t(i));
    auto less = [&](uint32_t a, uint32_t b) {
        const uint32_t* pa = toks.data() + toff[a]; const uint32_t* pb = toks.data() + toff[b];
        const uint64_t la = toff[a + 1] - toff[a], lb = toff[b + 1] - toff[b];
        const uint64_t m = std::min(la, lb);
        for (uint64_t k = 0; k < m; ++k) if (pa[k] != pb[k]) return pa[k] < pb[k];
        if (la != lb) return la < lb;
        return a < b;
    };
    {
        const size_t m = idx.size();
        std::vector<size_t> cut(threads + 1);
        for (unsigned k = 0; k <= threads; ++k) cut[k] = m * k / threads;
        run([&](unsigned k) { std::sort(idx.begin() + cut[k], idx.begin() + cut[k + 1], less); });
        for (unsigned width = 1; width < threads; width *= 2) {         // pairwise merges, parallel per round
            std::vector<std::thread> th;
            for (unsigned k = 0; k + width < threads; k += 2 * width)
                th.emplace_back([&, k] { std::inplace_merge(idx.begin() + cut[k], idx.begin() + cut[k + width], idx.begin() + cut[std::min(threads, k + 2 * width)], less); });
            for (auto& t : th) t.join();
        }
    }
    lap("sort");
    // ---- insert in sorted order: only the levels past the common prefix with the predecessor.
    // The sorted order also gives the exact number of trie nodes the batch can add, so the edge
    // table is sized once (no rehash while inserting).
    uint64_t new_nodes_total = 0;
    {
        uint64_t new_nodes = 0;
        const uint32_t* pv = nullptr; uint64_t pl = 0;
        for (uint32_t i : idx) {
            const uint32_t* cu = toks.data() + toff[i];
            const uint64_t L = toff[i + 1] - toff[i];
            uint64_t lcp = 0;
            while (lcp < L && lcp < pl && cu[lcp] == pv[lcp]) ++lcp;
            new_nodes += L - lcp;
            pv = cu; pl = L;
        }
        new_nodes_total = new_nodes;
    }
    // Empty table (the restore case): in sorted order a child exists iff it lies on the previous
    // filter's path, so no hash probe is needed while creating nodes; the edge table is
    // materialised once at the end (sequential pass with software prefetch) instead of 2 random
    // DRAM accesses per created node.
    const bool deferred = n_nodes_ == 1 && edge_live_ == 0;
    std::vector<uint64_t> lit_bits;       // per node: child-token bitmap (kernels.hpp lit_bit)
    if (deferred) {
        lit_bits.assign(nodes_.size(), 0);
        nodes_.reserve(nodes_.size() + new_nodes_total);
        lit_bits.reserve(nodes_.size() + new_nodes_total);
        filters_.reserve(n_filters_ + idx.size());
    } else {
        reserve(n_filters_ + idx.size(), n_nodes_ + 2 * new_nodes_total);
    }
    auto child_of = [&](uint32_t parent, uint32_t tok) -> uint32_t {
        if (!deferred) {
            const uint32_t s = find_slot(parent, tok);
            return s != kNone ? edges_[s].child : new_node(parent, tok);
        }
        const uint32_t id = uint32_t(nodes_.size());
        nodes_.push_back(Node{parent, tok, kNone, 0, kNone, kNone, kNone});
        lit_bits.push_back(0);
        nodes_[parent].nchild++;
        n_nodes_++;
        if (tok == kTokPlus) nodes_[parent].plus_child = id;
        else if (tok == kTokHash) nodes_[parent].hash_child = id;
        else lit_bits[parent] |= 1ull << lit_bit(tok);
        return id;
    };
    std::vector<uint32_t> path{0};                   // path[l] = node reached after l levels of the previous filter
    const uint32_t* prev = nullptr; uint64_t prev_len = 0;
    for (uint32_t i : idx) {
        const uint32_t* cur = toks.data() + toff[i];
        const uint64_t L = toff[i + 1] - toff[i];
        uint64_t lcp = 0;
        while (lcp < L && lcp < prev_len && cur[lcp] == prev[lcp]) ++lcp;
        path.resize(lcp + 1);
        uint32_t node = path[lcp];
        for (uint64_t l = lcp; l < L; ++l) {
            node = child_of(node, cur[l]);
            path.push_back(node);
        }
        prev = cur; prev_len = L;
        uint32_t fid = nodes_[node].term_fid;
        if (fid == kNone) {
            if (!free_fids_.empty()) { fid = free_fids_.back(); free_fids_.pop_back(); }
            else { fid = uint32_t(filters_.size()); filters_.emplace_back(); }
            filters_[fid].node = node;
            filters_[fid].subs.clear();
            delta_.fids.push_back(fid);
            if (deferred) nodes_[node].term_fid = fid;          // headers are written by materialize_edges()
            else {
                set_term_fid(node, fid);
                if (nodes_[node].token == kTokHash) set_hash_fid(nodes_[node].parent, fid);
            }
            n_filters_++;
        }
        sub_add(fid, sub_ids ? sub_ids[i] : i, qos ? qos[i] : 0, flags ? flags[i] : 0);
        if (fids_out) fids_out[i] = fid;
    }
    lap("insert");
    if (deferred) materialize_edges(lit_bits);
    lap("materialise");
    if (fids_out) for (uint64_t i = 0; i < n; ++i) if (!valid[i]) fids_out[i] = kNone;
    if (n_rejected) *n_rejected = rejected;
}

// Build the whole edge table from the node array (bulk restore into an empty table).
void HostTable::materialize_edges(const std::vector<uint64_t>& lit_bits) {
    // load <= 0.25 by default: short probe sequences for the walk kernel.  RGR_EDGE_SLOTS_PER_NODE (2..16) trades probe
    // length against footprint — a denser table may stay resident in the 256 MiB Infinity Cache (tools/walk_lab)
    uint64_t per_node = 4;
    if (const char* e = std::getenv("RGR_EDGE_SLOTS_PER_NODE")) per_node = std::min<uint64_t>(16, std::max<uint64_t>(2, std::strtoull(e, nullptr, 10)));
    uint64_t cap = 1024;
    while (cap < n_nodes_ * per_node) cap <<= 1;
    const bool prof = std::getenv("RGR_BULK_PROFILE") != nullptr;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = tnow();
    auto lap = [&](const char* what) { if (prof) { const double t = tnow(); std::fprintf(stderr, "[bulk]   mat.%-8s %.3f s\n", what, t - t_prev); t_prev = t; } };
    edges_.assign(cap, empty_edge());
    lap("assign");
    const uint32_t mask = uint32_t(cap - 1);
    const uint32_t total = uint32_t(nodes_.size());
    std::vector<uint8_t> dead(total, 0);
    for (uint32_t f : free_nodes_) dead[f] = 1;
    auto hdr_hash_fid = [&](uint32_t id) { const uint32_t hc = nodes_[id].hash_child; return hc == kNone ? kNone : nodes_[hc].term_fid; };
    // Cache-friendly placement: bucket the nodes by the high bits of their home slot (counting
    // sort), then fill bucket after bucket — every bucket spans 256 KiB of the table, so the random
    // probes of a bucket stay in cache / TLB instead of touching a different page per node.
    const uint32_t bucket_shift = cap > (1u << 13) ? uint32_t(__builtin_ctzll(cap)) - 13 : 0;   // 8192 slots per bucket
    const uint32_t nbuckets = uint32_t(cap >> bucket_shift);
    std::vector<uint32_t> bstart(size_t(nbuckets) + 1, 0), order(n_nodes_ ? n_nodes_ - 1 : 0);
    for (uint32_t id = 1; id < total; ++id)
        if (!dead[id]) bstart[((edge_hash(nodes_[id].parent, nodes_[id].token) & mask) >> bucket_shift) + 1]++;
    for (uint32_t b = 0; b < nbuckets; ++b) bstart[b + 1] += bstart[b];
    {
        std::vector<uint32_t> pos(bstart.begin(), bstart.end() - 1);
        for (uint32_t id = 1; id < total; ++id)
            if (!dead[id]) order[pos[(edge_hash(nodes_[id].parent, nodes_[id].token) & mask) >> bucket_shift]++] = id;
    }
    lap("bucket");
    for (uint32_t id : order) {
        Node& nd = nodes_[id];
        uint32_t i = edge_hash(nd.parent, nd.token) & mask;
        while (edges_[i].parent != kEdgeEmpty) i = (i + 1) & mask;
        edges_[i] = EdgeEntry{nd.parent, nd.token, id, kNone, hdr_hash_fid(id), nd.term_fid, uint32_t(lit_bits[id]), uint32_t(lit_bits[id] >> 32)};
        nd.slot = i;
    }
    lap("fill");
    for (uint32_t id = 1; id < total; ++id) {         // '+' edge slots are known only now
        if (dead[id] || nodes_[id].plus_child == kNone) continue;
        edges_[nodes_[id].slot].plus_slot = nodes_[nodes_[id].plus_child].slot;
    }
    root_hdr_ = NodeHeader{nodes_[0].plus_child == kNone ? kNone : nodes_[nodes_[0].plus_child].slot, hdr_hash_fid(0), nodes_[0].term_fid,
                           uint32_t(lit_bits[0]), uint32_t(lit_bits[0] >> 32)};
    edge_live_ = edge_used_ = n_nodes_ - 1;
    delta_.relocated = true;
}

void HostTable::flatten_filters(std::vector<FilterDesc>& filt, std::vector<SubEntry>& subs) const {
    filt.resize(filters_.size());
    subs.clear();
    subs.reserve(n_subs_);
    for (size_t i = 0; i < filters_.size(); ++i) {
        filt[i] = FilterDesc{uint32_t(subs.size()), uint32_t(filters_[i].subs.size())};
        subs.insert(subs.end(), filters_[i].subs.begin(), filters_[i].subs.end());
    }
}

uint64_t HostTable::max_filter_subs() const {
    uint64_t m = 0;
    for (auto& f : filters_) m = std::max<uint64_t>(m, f.subs.size());
    return m;
}

// ------------------------------------------------------------------------------ snapshot file
namespace {
constexpr char kSnapMagic[8] = {'R', 'G', 'R', 'S', 'N', 'A', 'P', '1'};
constexpr uint32_t kSnapVersion = 2;      // 2: the last 8 bytes of an edge record are the child-token bitmap (1: lit_cnt, lit_xor)
struct SnapWriter {
    std::FILE* f;
    uint64_t sum = 0xcbf29ce484222325ull;
    bool ok = true;
    void raw(const void* p, size_t n) {
        if (!n) return;
        const unsigned char* b = static_cast<const unsigned char*>(p);
        // checksum: FNV-1a over 8-byte words (tail bytewise) — a corruption check, not a MAC
        size_t i = 0;
        for (; i + 8 <= n; i += 8) { uint64_t w; std::memcpy(&w, b + i, 8); sum ^= w; sum *= 0x100000001b3ull; }
        for (; i < n; ++i) { sum ^= b[i]; sum *= 0x100000001b3ull; }
        if (ok && std::fwrite(p, 1, n, f) != n) ok = false;
    }
    template <class T> void pod(const T& v) { raw(&v, sizeof v); }
    template <class T> void vec(const std::vector<T>& v) { const uint64_t n = v.size(); pod(n); raw(v.data(), n * sizeof(T)); }
};
struct SnapReader {
    std::FILE* f;
    uint64_t sum = 0xcbf29ce484222325ull, left;
    bool raw(void* p, size_t n) {
        if (!n) return true;
        if (n > left || std::fread(p, 1, n, f) != n) return false;
        left -= n;
        const unsigned char* b = static_cast<const unsigned char*>(p);
        size_t i = 0;
        for (; i + 8 <= n; i += 8) { uint64_t w; std::memcpy(&w, b + i, 8); sum ^= w; sum *= 0x100000001b3ull; }
        for (; i < n; ++i) { sum ^= b[i]; sum *= 0x100000001b3ull; }
        return true;
    }
    template <class T> bool pod(T& v) { return raw(&v, sizeof v); }
    template <class T> bool vec(std::vector<T>& v) {
        uint64_t n = 0;
        if (!pod(n) || n > left / sizeof(T)) return false;      // a length that cannot fit in the file: corrupt
        v.resize(n);
        return raw(v.data(), n * sizeof(T));
    }
};
struct SnapHeader {
    char magic[8];
    uint32_t version, edge_bytes, node_bytes, sub_bytes;
    uint64_t n_filters, n_subs, n_nodes, n_v5, edge_used, edge_live;
};
}  // namespace

bool HostTable::save(const std::string& path, std::string* err) const {
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { if (err) *err = "cannot open " + path + " for writing"; return false; }
    SnapWriter w{f};
    SnapHeader h{};
    std::memcpy(h.magic, kSnapMagic, 8);
    // RGR_SNAPSHOT_WRITE_V1 (tests only): a version-1 file as round-2 builds wrote it — the last 8 bytes of every node header are
    // not a child-token bitmap (here: a pattern that would break the walk if a loader trusted it)
    const bool legacy = std::getenv("RGR_SNAPSHOT_WRITE_V1") != nullptr;
    h.version = legacy ? 1 : kSnapVersion; h.edge_bytes = sizeof(EdgeEntry); h.node_bytes = sizeof(Node); h.sub_bytes = sizeof(SubEntry);
    h.n_filters = n_filters_; h.n_subs = n_subs_; h.n_nodes = n_nodes_; h.n_v5 = n_v5_; h.edge_used = edge_used_; h.edge_live = edge_live_;
    w.pod(h);
    dict_.save(w);
    w.vec(nodes_); w.vec(free_nodes_);
    {   // the edge table is mostly empty slots (load <= 0.25): only occupied slots (tombstones included) are stored
        const uint64_t cap = edges_.size();
        std::vector<uint32_t> idx;
        std::vector<EdgeEntry> recs;
        idx.reserve(edge_used_); recs.reserve(edge_used_);
        for (uint64_t i = 0; i < cap; ++i)
            if (edges_[i].parent != kEdgeEmpty) { idx.push_back(uint32_t(i)); recs.push_back(edges_[i]); }
        if (legacy) for (EdgeEntry& e : recs) { e.lit_lo = 1; e.lit_hi = 0; }
        w.pod(cap); w.vec(idx); w.vec(recs);
    }
    NodeHeader root = root_hdr_;
    if (legacy) { root.lit_lo = 1; root.lit_hi = 0; }
    w.pod(root);
    std::vector<uint32_t> fnode(filters_.size());
    std::vector<FilterDesc> fdesc;
    std::vector<SubEntry> subs;
    for (size_t i = 0; i < filters_.size(); ++i) fnode[i] = filters_[i].node;
    flatten_filters(fdesc, subs);
    w.vec(fnode); w.vec(fdesc); w.vec(subs); w.vec(free_fids_);
    const uint8_t ha = has_attrs_ ? 1 : 0;
    w.pod(ha);
    w.vec(attrs_);
    const uint64_t sum = w.sum;
    if (w.ok && std::fwrite(&sum, 1, 8, f) != 8) w.ok = false;
    if (std::fclose(f) != 0) w.ok = false;
    if (!w.ok && err) *err = "write error on " + path;
    return w.ok;
}

bool HostTable::load(const std::string& path, std::string* err) {
    auto bad = [&](const char* what) { if (err) *err = path + ": " + what; return false; };
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return bad("cannot open");
    struct Closer { std::FILE* f; ~Closer() { std::fclose(f); } } closer{f};
    if (std::fseek(f, 0, SEEK_END) != 0) return bad("cannot seek");
    const long size = std::ftell(f);
    if (size < long(sizeof(SnapHeader) + 8)) return bad("not a snapshot (too short)");
    std::rewind(f);
    SnapReader r{f};
    r.left = uint64_t(size) - 8;
    SnapHeader h{};
    if (!r.pod(h) || std::memcmp(h.magic, kSnapMagic, 8) != 0) return bad("not a snapshot (bad magic)");
    // version 1 differs only in the last 8 bytes of an edge record ({lit_cnt, lit_xor} instead of the child-token bitmap), and those are
    // derived data: a v1 file loads like a v2 one and has its bitmaps rebuilt below (an upgrade does not discard persisted snapshots)
    if ((h.version != kSnapVersion && h.version != 1) || h.edge_bytes != sizeof(EdgeEntry) || h.node_bytes != sizeof(Node) || h.sub_bytes != sizeof(SubEntry))
        return bad("snapshot of an incompatible build");
    const bool prof = std::getenv("RGR_BULK_PROFILE") != nullptr;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = tnow();
    auto lap = [&](const char* what) { if (prof) { const double tn = tnow(); std::fprintf(stderr, "[snapshot] %-12s %.3f s\n", what, tn - t_prev); t_prev = tn; } };
    HostTable t;                                   // parsed on the side: *this is untouched unless everything checks out
    std::vector<uint32_t> fnode;
    std::vector<FilterDesc> fdesc;
    std::vector<SubEntry> subs;
    uint8_t ha = 0;
    uint64_t ecap = 0;
    std::vector<uint32_t> eidx;
    std::vector<EdgeEntry> erecs;
    if (!t.dict_.load(r) || !r.vec(t.nodes_) || !r.vec(t.free_nodes_) || !r.pod(ecap) || !r.vec(eidx) || !r.vec(erecs) || !r.pod(t.root_hdr_) || !r.vec(fnode) ||
        !r.vec(fdesc) || !r.vec(subs) || !r.vec(t.free_fids_) || !r.pod(ha) || !r.vec(t.attrs_))
        return bad("truncated or corrupt");
    lap("read");
    uint64_t sum = 0;
    if (r.left != 0 || std::fread(&sum, 1, 8, f) != 8 || sum != r.sum) return bad("checksum mismatch");
    if (ecap < 2 || ecap > (1ull << 32) || (ecap & (ecap - 1)) || eidx.size() != erecs.size() || eidx.size() > ecap) return bad("inconsistent edge table");
    t.edges_.assign(ecap, empty_edge());
    for (size_t i = 0; i < eidx.size(); ++i) {
        if (eidx[i] >= ecap || (i && eidx[i] <= eidx[i - 1])) return bad("edge slots out of order");
        t.edges_[eidx[i]] = erecs[i];
    }
    std::vector<uint32_t>().swap(eidx);
    std::vector<EdgeEntry>().swap(erecs);
    lap("edge table");
    // structural checks: everything the kernels index with must be in range
    if (t.edges_.size() < 2 || (t.edges_.size() & (t.edges_.size() - 1)) || t.nodes_.empty() || fdesc.size() != fnode.size())
        return bad("inconsistent arrays");
    const uint64_t nf = fnode.size();
    for (const EdgeEntry& e : t.edges_) {
        if (e.parent == kEdgeEmpty || e.parent == kEdgeTomb) continue;
        if (e.parent >= t.nodes_.size() || e.child >= t.nodes_.size() || (e.plus_slot != kNone && e.plus_slot >= t.edges_.size()) ||
            (e.hash_fid != kNone && e.hash_fid >= nf) || (e.term_fid != kNone && e.term_fid >= nf))
            return bad("edge record out of range");
    }
    if ((t.root_hdr_.plus_slot != kNone && t.root_hdr_.plus_slot >= t.edges_.size()) || (t.root_hdr_.hash_fid != kNone && t.root_hdr_.hash_fid >= nf) ||
        (t.root_hdr_.term_fid != kNone && t.root_hdr_.term_fid >= nf))
        return bad("root header out of range");
    for (const Node& nd : t.nodes_) {
        auto in = [](uint32_t v, uint64_t lim) { return v == kNone || v < lim; };
        if (!in(nd.parent, t.nodes_.size()) || !in(nd.slot, t.edges_.size()) || !in(nd.term_fid, nf) || !in(nd.plus_child, t.nodes_.size()) ||
            !in(nd.hash_child, t.nodes_.size()))
            return bad("trie node out of range");
    }
    for (uint32_t v : t.free_nodes_) if (v >= t.nodes_.size()) return bad("free list out of range");
    for (uint32_t v : t.free_fids_) if (v >= nf) return bad("free list out of range");
    uint64_t total = 0;
    for (uint64_t i = 0; i < nf; ++i) {
        if (fdesc[i].begin != total || uint64_t(fdesc[i].begin) + fdesc[i].count > subs.size()) return bad("subscriber runs out of range");
        if (fnode[i] != kNone && fnode[i] >= t.nodes_.size()) return bad("filter node out of range");
        total += fdesc[i].count;
    }
    if (total != subs.size() || total != h.n_subs) return bad("subscriber count mismatch");
    lap("checks");
    t.filters_.resize(nf);
    for (uint64_t i = 0; i < nf; ++i) {
        t.filters_[i].node = fnode[i];
        t.filters_[i].subs.assign(subs.begin() + fdesc[i].begin, subs.begin() + fdesc[i].begin + fdesc[i].count);
    }
    // the counters are recomputed from the arrays (the header's copies are only cross-checked): find_slot /
    // insert_edge / the device probe loops terminate only while the table keeps empty slots, and the rehash
    // thresholds read edge_used_ / edge_live_
    uint64_t used = 0, live = 0, n_v5 = 0, live_filters = 0;
    for (const EdgeEntry& e : t.edges_) { if (e.parent == kEdgeEmpty) continue; used++; if (e.parent != kEdgeTomb) live++; }
    for (uint64_t i = 0; i < nf; ++i) if (fnode[i] != kNone) live_filters++;
    uint32_t max_sub = 0, max_node = 0;
    for (const SubEntry& se : subs) { if ((se.qos_flags >> 8) & kSubV5) n_v5++; if (se.sub_id > max_sub) max_sub = se.sub_id; if ((se.qos_flags >> 16) > max_node) max_node = se.qos_flags >> 16; }
    const uint64_t n_nodes = t.nodes_.size() - t.free_nodes_.size();
    if (used * 2 > t.edges_.size()) return bad("edge table over its load limit");
    if (t.free_nodes_.size() >= t.nodes_.size() || live + 1 != n_nodes) return bad("trie node count mismatch");
    if (h.edge_used != used || h.edge_live != live || h.n_nodes != n_nodes || h.n_filters != live_filters || h.n_v5 != n_v5 ||
        t.free_fids_.size() + live_filters != nf)
        return bad("header counters do not match the arrays");
    t.n_filters_ = live_filters; t.n_subs_ = total; t.n_nodes_ = n_nodes; t.n_v5_ = n_v5; t.max_sub_id_ = max_sub; t.max_node_idx_ = max_node;
    t.edge_used_ = used; t.edge_live_ = live;
    if (h.version == 1) t.rebuild_child_bitmaps();
    t.has_attrs_ = ha != 0;
    t.dict_gen_ = dict_gen_ + 1;
    t.delta_ = Delta{};
    t.delta_.relocated = true;                     // every device image must be rebuilt
    t.attrs_all_dirty_ = true;
    lap("filters");
    *this = std::move(t);
    lap("swap");
    return true;
}

}  // namespace rgr
