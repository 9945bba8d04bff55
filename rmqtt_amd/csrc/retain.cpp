// RetainTree twin — host table + snapshot compiler (see retain.hpp).
#include "retain.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "rmqtt_gpu_router.h"
#include "topic.hpp"

namespace rgr {

static REdge empty_redge() { return REdge{kEdgeEmpty, 0, kNone, 0}; }

RetainTable::RetainTable() {
    nodes_.push_back(Node{kNone, 0, kNone, 0, kNone, false});
    edges_.assign(1024, empty_redge());
}

uint32_t RetainTable::find(uint32_t parent, uint32_t token) const {
    const uint32_t mask = uint32_t(edges_.size() - 1);
    for (uint32_t i = edge_hash(parent, token) & mask;; i = (i + 1) & mask) {
        const REdge& e = edges_[i];
        if (e.parent == kEdgeEmpty) return kNone;
        if (e.parent == parent && e.token == token) return i;
    }
}

void RetainTable::rehash(uint64_t cap) {
    std::vector<REdge> old;
    old.swap(edges_);
    edges_.assign(cap, empty_redge());
    const uint32_t mask = uint32_t(cap - 1);
    for (const REdge& e : old) {
        if (e.parent == kEdgeEmpty || e.parent == kEdgeTomb) continue;
        uint32_t i = edge_hash(e.parent, e.token) & mask;
        while (edges_[i].parent != kEdgeEmpty) i = (i + 1) & mask;
        edges_[i] = e;
        nodes_[e.child].slot = i;
    }
    edge_used_ = edge_live_;
}

uint32_t RetainTable::insert_edge(uint32_t parent, uint32_t token, uint32_t child) {
    if ((edge_used_ + 1) * 2 > edges_.size()) rehash(edge_live_ * 4 > edges_.size() ? edges_.size() * 2 : edges_.size());
    const uint32_t mask = uint32_t(edges_.size() - 1);
    uint32_t i = edge_hash(parent, token) & mask;
    while (edges_[i].parent != kEdgeEmpty && edges_[i].parent != kEdgeTomb) i = (i + 1) & mask;
    if (edges_[i].parent == kEdgeEmpty) edge_used_++;
    edges_[i] = REdge{parent, token, child, 0};
    edge_live_++;
    return i;
}

bool RetainTable::tokenize(std::string_view s, std::vector<uint32_t>& toks, bool intern, bool* first_meta) {
    toks.clear();
    if (for_each_level(s, [](int64_t, std::string_view, LevelKind) {}) < 0) return false;
    for_each_level(s, [&](int64_t idx, std::string_view seg, LevelKind k) {
        if (idx == 0 && first_meta) *first_meta = k == LevelKind::Metadata;
        if (k == LevelKind::Plus) toks.push_back(kTokPlus);
        else if (k == LevelKind::Hash) toks.push_back(kTokHash);
        else toks.push_back(intern ? dict_.intern(seg) : dict_.find(seg));
    });
    return true;
}

uint8_t RetainTable::tokenize_filter(std::string_view f, std::vector<uint32_t>& toks) const {
    const size_t mark = toks.size();
    int64_t n = for_each_level(f, [&](int64_t, std::string_view seg, LevelKind k) {
        if (k == LevelKind::Plus) toks.push_back(kTokPlus);
        else if (k == LevelKind::Hash) toks.push_back(kTokHash);
        else toks.push_back(dict_.find(seg));
    });
    if (n < 0) { toks.resize(mark); return kTopicInvalid; }
    return 0;
}

int32_t RetainTable::topic_add(std::string_view topic, uint32_t topic_id) {
    std::vector<uint32_t> toks;
    bool meta = false;
    if (!tokenize(topic, toks, true, &meta)) return RGR_EINVAL_TOPIC;
    uint32_t cur = 0;
    for (size_t l = 0; l < toks.size(); ++l) {
        const uint32_t s = find(cur, toks[l]);
        if (s != kNone) { cur = edges_[s].child; continue; }
        uint32_t id;
        if (!free_nodes_.empty()) { id = free_nodes_.back(); free_nodes_.pop_back(); }
        else { id = uint32_t(nodes_.size()); nodes_.push_back(Node{}); }
        nodes_[id] = Node{cur, toks[l], kNone, 0, kNone, l == 0 && meta};
        const uint32_t slot = insert_edge(cur, toks[l], id);
        nodes_[id].slot = slot;
        nodes_[cur].nchild++;
        n_nodes_++;
        cur = id;
    }
    if (nodes_[cur].value == kNone) n_values_++;
    if (nodes_[cur].value != topic_id) version_++;   // re-publishing a retained topic under the same id changes nothing on the device
    nodes_[cur].value = topic_id;                    // value.replace(), retain.rs:384
    if (topic_id > max_id_) max_id_ = topic_id;
    return RGR_OK;
}

int32_t RetainTable::topic_remove(std::string_view topic) {
    std::vector<uint32_t> toks;
    if (!tokenize(topic, toks, false, nullptr)) return RGR_EINVAL_TOPIC;
    uint32_t cur = 0;
    for (uint32_t t : toks) {
        if (t == kTokUnknown) return RGR_ENOENT;
        const uint32_t s = find(cur, t);
        if (s == kNone) return RGR_ENOENT;
        cur = edges_[s].child;
    }
    if (nodes_[cur].value == kNone) return RGR_ENOENT;
    nodes_[cur].value = kNone;
    n_values_--;
    version_++;
    while (cur != 0 && nodes_[cur].value == kNone && nodes_[cur].nchild == 0) {   // retain.rs:405-407
        const uint32_t p = nodes_[cur].parent;
        edges_[nodes_[cur].slot].parent = kEdgeTomb;
        edge_live_--;
        nodes_[p].nchild--;
        free_nodes_.push_back(cur);
        n_nodes_--;
        cur = p;
    }
    return RGR_OK;
}

uint32_t RetainTable::find_node(std::string_view topic) const {
    std::vector<uint32_t> toks;
    if (for_each_level(topic, [](int64_t, std::string_view, LevelKind) {}) < 0) return kNone;
    bool known = true;
    for_each_level(topic, [&](int64_t, std::string_view seg, LevelKind k) {
        const uint32_t t = k == LevelKind::Plus ? kTokPlus : k == LevelKind::Hash ? kTokHash : dict_.find(seg);
        if (t == kTokUnknown) known = false;
        toks.push_back(t);
    });
    if (!known) return kNone;
    uint32_t cur = 0;
    for (uint32_t t : toks) {
        const uint32_t s = find(cur, t);
        if (s == kNone) return kNone;
        cur = edges_[s].child;
    }
    return cur;
}

void RetainTable::compile(RetainImage& out, std::vector<uint32_t>* val_of_node) const {
    const bool prof = std::getenv("RGR_BULK_PROFILE") != nullptr;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = tnow();
    auto lap = [&](const char* what) { if (prof) { const double tn = tnow(); std::fprintf(stderr, "[retain.compile] %-12s %.3f s\n", what, tn - t_prev); t_prev = tn; } };
    const uint32_t total = uint32_t(nodes_.size());
    // ---- children lists of the mutable ids as packed (token << 32 | child), counting-sorted by
    // parent and ordered by token; the root's non-'$' children first (retain.rs:486-490, 505-509
    // skip '$' children at the root).  One pass over the edge table; everything after this works
    // on sequential per-preorder arrays instead of chasing nodes_[].
    RetainCompileScratch& sc = scratch_;
    auto& cnt = sc.cnt;
    cnt.assign(size_t(total) + 1, 0);
    auto& hash_kid = sc.hash_kid;                         // node -> its literal "#" child (rare: filled on first use)
    hash_kid.clear();
    for (const REdge& e : edges_) {
        if (e.parent == kEdgeEmpty || e.parent == kEdgeTomb) continue;
        cnt[e.parent + 1]++;
        if (e.token == kTokHash) {
            if (hash_kid.empty()) hash_kid.assign(total, kNone);
            hash_kid[e.parent] = e.child;
        }
    }
    for (uint32_t i = 0; i < total; ++i) cnt[i + 1] += cnt[i];
    auto& kt = sc.kt;
    kt.resize(cnt[total]);
    {
        auto& pos = sc.pos;
        pos.assign(cnt.begin(), cnt.end() - 1);
        for (const REdge& e : edges_)
            if (e.parent != kEdgeEmpty && e.parent != kEdgeTomb) kt[pos[e.parent]++] = (uint64_t(e.token) << 32) | e.child;
    }
    auto kid = [&](uint32_t k) { return uint32_t(kt[k]); };
    if (cnt[1] - cnt[0] >= 2)
        std::sort(kt.begin() + cnt[0], kt.begin() + cnt[1], [&](uint64_t x, uint64_t y) {
            const bool mx = nodes_[uint32_t(x)].meta, my = nodes_[uint32_t(y)].meta;
            return mx != my ? !mx : x < y;
        });
    for (uint32_t p = 1; p < total; ++p)
        if (cnt[p + 1] - cnt[p] >= 2) std::sort(kt.begin() + cnt[p], kt.begin() + cnt[p + 1]);   // token is the high half
    out.root_nonmeta = 0;
    for (uint32_t k = cnt[0]; k < cnt[1]; ++k) out.root_nonmeta += !nodes_[kid(k)].meta;
    lap("children");
    // ---- iterative DFS: preorder ids; per preorder id its mutable id, parent (preorder), token, subtree end
    const uint32_t N = uint32_t(n_nodes_);
    auto &order = sc.order, &ppre = sc.ppre, &ptok = sc.ptok, &sub_end = sc.sub_end;
    order.assign(N, 0); ppre.assign(N, kNone); ptok.assign(N, 0); sub_end.assign(N, 0);
    {
        struct Frame { uint32_t m, p, k; };
        std::vector<Frame> st;
        uint32_t next = 1;
        order[0] = 0;
        st.push_back(Frame{0u, 0u, cnt[0]});
        while (!st.empty()) {
            Frame& top = st.back();
            if (top.k < cnt[top.m + 1]) {
                const uint64_t w = kt[top.k++];
                const uint32_t c = uint32_t(w), p = next++;
                order[p] = c; ppre[p] = top.p; ptok[p] = uint32_t(w >> 32);
                st.push_back(Frame{c, p, cnt[c]});
            } else {
                sub_end[top.p] = next;
                st.pop_back();
            }
        }
    }
    // the root's first '$' child in preorder (N if none): children of p are p+1, sub_end[p+1], ...
    uint32_t first_meta_pre = N > 1 ? 1 : N;
    for (uint32_t i = 0; i < out.root_nonmeta; ++i) first_meta_pre = sub_end[first_meta_pre];
    if (cnt[1] - cnt[0] == out.root_nonmeta) first_meta_pre = N;
    lap("preorder");
    out.n_nodes = N;
    out.child_off.assign(size_t(N) + 1, 0);
    out.child_ids.clear();
    out.child_ids.reserve(N);
    out.vals.clear();
    for (uint32_t p = 0; p < N; ++p) {
        const uint32_t m = order[p];
        out.child_off[p] = uint32_t(out.child_ids.size());
        uint32_t c = p + 1;
        for (uint32_t k = cnt[m]; k < cnt[m + 1]; ++k) { out.child_ids.push_back(c); c = sub_end[c]; }
    }
    out.child_off[N] = uint32_t(out.child_ids.size());
    // ---- value layout.  desc[2p] = p's own value; desc[2p+1] = H(p) = what `p._matches(["#"])`
    // yields (retain.rs:502-524): every value below p — EXCEPT that a node y storing a literal "#"
    // level (child c) answers through the exact-first branch (retain.rs:472-483): below y only c's
    // own value is visible from y or from above.  Values are placed by a DFS that stops at such a
    // y after placing y's and c's values; the rest of subtree(y) goes to a deferred area of its
    // own (still contiguous for every start inside it).  With no literal "#" levels stored this is
    // plain preorder.
    auto &own_pos = sc.own_pos, &hp_b = sc.hp_b, &hp_e = sc.hp_e;               // by mutable id, literal-"#" path only
    auto hash_child = [&](uint32_t m) -> uint32_t { return hash_kid.empty() ? kNone : hash_kid[m]; };
    FilterDesc root_desc{0, 0};
    auto& has = sc.has;
    has.assign(N, 0);
    if (hash_kid.empty()) {
        // no literal "#" level anywhere (the normal case): plain preorder, no traversal needed —
        // H(p) is the value range of p's proper descendants
        auto& val_rank = sc.val_rank;
        val_rank.assign(size_t(N) + 1, 0);
        for (uint32_t p = 0; p < N; ++p) {
            const uint32_t v = nodes_[order[p]].value;
            val_rank[p] = uint32_t(out.vals.size());
            if (v != kNone) { out.vals.push_back(SubEntry{v, 0}); has[p] = 1; }
        }
        val_rank[N] = uint32_t(out.vals.size());
        out.desc.assign(2 * size_t(N) + 1, FilterDesc{0, 0});
        for (uint32_t p = 0; p < N; ++p) {
            out.desc[2 * size_t(p)] = FilterDesc{val_rank[p], has[p]};
            out.desc[2 * size_t(p) + 1] = FilterDesc{val_rank[p] + has[p], val_rank[sub_end[p]] - val_rank[p] - has[p]};
        }
        root_desc = FilterDesc{val_rank[0] + has[0], val_rank[first_meta_pre] - val_rank[0] - has[0]};
        if (val_of_node) {
            val_of_node->assign(total, kNone);
            for (uint32_t p = 0; p < N; ++p) if (has[p]) (*val_of_node)[order[p]] = val_rank[p];
        }
    } else {
        own_pos.assign(total, 0); hp_b.assign(total, 0); hp_e.assign(total, 0);
        auto place_value = [&](uint32_t m) { own_pos[m] = uint32_t(out.vals.size()); if (nodes_[m].value != kNone) out.vals.push_back(SubEntry{nodes_[m].value, 0}); };
        std::vector<uint32_t> deferred;                       // barrier nodes whose hidden part is still to be placed
        uint32_t first_meta_pos = kNone;
        // DFS below `m` (its own value is already placed): fills hp_b/hp_e of m and of everything visited
        auto place_below = [&](uint32_t m0) {
            std::vector<std::pair<uint32_t, uint32_t>> st;   // (node, next child index)
            auto enter = [&](uint32_t m) {
                hp_b[m] = uint32_t(out.vals.size());
                const uint32_t c = hash_child(m);
                if (c != kNone) {                             // barrier: only c's own value is visible below m
                    place_value(c);
                    hp_e[m] = uint32_t(out.vals.size());
                    deferred.push_back(m);
                    return;
                }
                st.emplace_back(m, cnt[m]);
            };
            enter(m0);
            while (!st.empty()) {
                auto& top = st.back();
                if (top.second < cnt[top.first + 1]) {
                    if (top.first == 0 && top.second == cnt[0] + out.root_nonmeta) first_meta_pos = uint32_t(out.vals.size());
                    const uint32_t k = kid(top.second++);
                    place_value(k);
                    enter(k);
                } else {
                    hp_e[top.first] = uint32_t(out.vals.size());
                    st.pop_back();
                }
            }
        };
        place_value(0);
        place_below(0);
        const bool root_barrier = hash_child(0) != kNone;
        if (first_meta_pos == kNone) first_meta_pos = hp_e[0];
        root_desc = root_barrier ? FilterDesc{hp_b[0], hp_e[0] - hp_b[0]} : FilterDesc{hp_b[0], first_meta_pos - hp_b[0]};
        for (size_t di = 0; di < deferred.size(); ++di) {     // (grows while we iterate)
            const uint32_t y = deferred[di], c = hash_child(y);
            for (uint32_t k = cnt[y]; k < cnt[y + 1]; ++k) {
                const uint32_t ch = kid(k);
                if (ch != c) place_value(ch);                 // c's own value sits next to y's
                place_below(ch);
            }
        }
        out.desc.assign(2 * size_t(N) + 1, FilterDesc{0, 0});
        for (uint32_t p = 0; p < N; ++p) {
            const uint32_t m = order[p];
            out.desc[2 * size_t(p)] = FilterDesc{own_pos[m], nodes_[m].value != kNone ? 1u : 0u};
            out.desc[2 * size_t(p) + 1] = FilterDesc{hp_b[m], hp_e[m] - hp_b[m]};
        }
        if (val_of_node) {
            val_of_node->assign(total, kNone);
            for (uint32_t p = 0; p < N; ++p) { const uint32_t m = order[p]; if (nodes_[m].value != kNone) (*val_of_node)[m] = own_pos[m]; }
        }
    }
    // root '#': everything outside the root's '$' subtrees (the non-meta children come first), or the
    // literal "#" topic alone when one is stored
    out.desc[2 * size_t(N)] = root_desc;
    lap("values");
    // ---- grandchild index: (grandparent g, literal token t) -> run of nodes x (preorder ascending)
    {
        using Tri = RetainCompileScratch::Tri;
        auto& tri = sc.tri;
        tri.clear();
        tri.reserve(N);
        for (uint32_t x = 1; x < N; ++x) {
            const uint32_t p = ppre[x];
            if (ptok[x] < kTokFirst || p == 0) continue;             // wildcard-token levels and depth-1 nodes are not indexed
            const uint32_t g = ppre[p];
            if (g == 0 && p >= first_meta_pre) continue;             // a '+' at the root skips '$' children (retain.rs:486-490)
            tri.push_back(Tri{g, ptok[x], x});
        }
        {   // order by (g, tok, x): counting sort on the grandparent, then each (small) group by (tok, x)
            auto& goff = sc.goff;
            goff.assign(size_t(N) + 1, 0);
            for (const Tri& t : tri) goff[t.g + 1]++;
            for (uint32_t i = 0; i < N; ++i) goff[i + 1] += goff[i];
            auto& sorted = sc.tri_sorted;
            sorted.resize(tri.size());
            {
                auto& pos = sc.pos;
                pos.assign(goff.begin(), goff.end() - 1);
                for (const Tri& t : tri) sorted[pos[t.g]++] = t;
            }
            tri.swap(sorted);
            for (uint32_t g = 0; g < N; ++g)
                if (goff[g + 1] - goff[g] > 1)
                    std::sort(tri.begin() + goff[g], tri.begin() + goff[g + 1], [](const Tri& a, const Tri& b) { return a.tok != b.tok ? a.tok < b.tok : a.x < b.x; });
        }
        out.gc_ids.resize(tri.size());
        size_t runs = 0;
        for (size_t i = 0; i < tri.size(); ++i) { out.gc_ids[i] = tri[i].x; runs += i == 0 || tri[i].g != tri[i - 1].g || tri[i].tok != tri[i - 1].tok; }
        uint64_t gcap = 1024;
        while (gcap < runs * 2) gcap <<= 1;
        out.gc_edges.assign(gcap, GcEdge{kEdgeEmpty, 0, 0, 0});
        const uint32_t gmask = uint32_t(gcap - 1);
        for (size_t i = 0; i < tri.size();) {
            size_t j = i;
            while (j < tri.size() && tri[j].g == tri[i].g && tri[j].tok == tri[i].tok) ++j;
            uint32_t s = edge_hash(tri[i].g, tri[i].tok) & gmask;
            while (out.gc_edges[s].gparent != kEdgeEmpty) s = (s + 1) & gmask;
            out.gc_edges[s] = GcEdge{tri[i].g, tri[i].tok, uint32_t(i), uint32_t(j - i)};
            i = j;
        }
    }
    lap("gc index");
    // ---- edge table over preorder ids.  Built in kParts fixed slot ranges (independent of the thread
    // count, so the image is deterministic): entries — in preorder — are bucketed by the range of
    // their home slot, every range is filled by one thread, and the few entries whose probe sequence
    // leaves its range are inserted afterwards, sequentially.
    uint64_t cap = 1024;
    while (cap < uint64_t(N) * 2) cap <<= 1;
    out.edges.assign(cap, empty_redge());
    lap("  edge.alloc");
    const uint32_t mask = uint32_t(cap - 1);
    {
        constexpr uint32_t kParts = 64;
        const uint64_t psz = cap / kParts;
        auto &home = sc.home, &pcount = sc.pcount;
        home.assign(N, 0); pcount.assign(kParts + 1, 0);
        for (uint32_t x = 1; x < N; ++x) { home[x] = edge_hash(ppre[x], ptok[x]) & mask; pcount[home[x] / psz + 1]++; }
        for (uint32_t k = 0; k < kParts; ++k) pcount[k + 1] += pcount[k];
        auto& sorted = sc.sorted;
        sorted.resize(N > 1 ? N - 1 : 0);
        {
            auto& pos = sc.pos;
            pos.assign(pcount.begin(), pcount.end() - 1);
            for (uint32_t x = 1; x < N; ++x) sorted[pos[home[x] / psz]++] = x;      // stable: preorder inside a range
        }
        lap("  edge.bucket");
        std::vector<std::vector<uint32_t>> spill(kParts);
        auto fill = [&](uint32_t part) {
            const uint64_t hi = uint64_t(part + 1) * psz;
            for (uint32_t k = pcount[part]; k < pcount[part + 1]; ++k) {
                const uint32_t x = sorted[k];
                uint64_t i = home[x];
                while (i < hi && out.edges[i].parent != kEdgeEmpty) ++i;
                if (i == hi) spill[part].push_back(x);
                else out.edges[i] = REdge{ppre[x], ptok[x], x, 0};
            }
        };
        const unsigned nt = std::max(1u, std::min<unsigned>({std::thread::hardware_concurrency(), kParts, unsigned(N / 65536 + 1)}));
        if (nt <= 1) { for (uint32_t part = 0; part < kParts; ++part) fill(part); }
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t)
                th.emplace_back([&, t] { for (uint32_t part = t; part < kParts; part += nt) fill(part); });
            for (auto& t : th) t.join();
        }
        lap("  edge.fill");
        for (uint32_t part = 0; part < kParts; ++part)
            for (const uint32_t x : spill[part]) {
                uint32_t i = home[x];
                while (out.edges[i].parent != kEdgeEmpty) i = (i + 1) & mask;
                out.edges[i] = REdge{ppre[x], ptok[x], x, 0};
            }
    }
    lap("edge table");
}

// ------------------------------------------------------------------ TieredRetain
namespace {
bool has_wildcard_level(std::string_view topic) {
    bool w = false;
    for_each_level(topic, [&](int64_t, std::string_view, LevelKind k) { if (k == LevelKind::Plus || k == LevelKind::Hash) w = true; });
    return w;
}
}  // namespace

bool TieredRetain::in_delta(std::string_view topic) const {
    const uint32_t n = delta_.find_node(topic);
    return n != kNone && delta_.node_value(n) != kNone;
}

void TieredRetain::mark_dead(uint32_t node, uint32_t topic_id) {
    const uint32_t idx = node < base_val_of_node_.size() ? base_val_of_node_[node] : kNone;
    if (idx == kNone) return;                  // (cannot happen: a valued topic outside the delta is a live base value)
    base_val_of_node_[node] = kNone;
    pending_dead_.push_back(Dead{idx, topic_id});
    n_dead_++;
}

int32_t TieredRetain::topic_add(std::string_view topic, uint32_t topic_id) {
    const uint32_t node = all_.find_node(topic);
    const uint32_t old = node != kNone ? all_.node_value(node) : kNone;
    const bool was_delta = merged_once_ && old != kNone && in_delta(topic);
    const int32_t rc = all_.topic_add(topic, topic_id);
    if (rc == RGR_OK && old == kNone && has_wildcard_level(topic)) { n_wild_++; force_merge_ = true; }
    if (rc != RGR_OK || !merged_once_ || old == topic_id) return rc;
    if (old != kNone && !was_delta) mark_dead(node, old);     // value of a base topic replaced: the new one lives in the delta
    return delta_.topic_add(topic, topic_id);
}

int32_t TieredRetain::topic_remove(std::string_view topic) {
    const uint32_t node = all_.find_node(topic);
    const uint32_t old = node != kNone ? all_.node_value(node) : kNone;
    const bool was_delta = merged_once_ && old != kNone && in_delta(topic);
    const int32_t rc = all_.topic_remove(topic);
    if (rc == RGR_OK && has_wildcard_level(topic)) { n_wild_--; force_merge_ = true; }
    if (rc != RGR_OK || !merged_once_) return rc;
    if (was_delta) return delta_.topic_remove(topic);
    mark_dead(node, old);
    return RGR_OK;
}

void TieredRetain::compile_base(RetainImage& out) {
    all_.compile(out, &base_val_of_node_);
    delta_ = RetainTable();
    delta_compiled_ = delta_.version();
    pending_dead_.clear();
    n_dead_ = 0;
    base_topics_ = all_.n_topics();
    merged_once_ = true;
    force_merge_ = false;
}

void TieredRetain::compile_delta(RetainImage& out) {
    delta_.compile(out);
    delta_compiled_ = delta_.version();
}

}  // namespace rgr
