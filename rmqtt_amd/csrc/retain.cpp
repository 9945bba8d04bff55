// RetainTree twin — host table + snapshot compiler (see retain.hpp).
#include "retain.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

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

void RetainTable::compile(RetainImage& out) const {
    const bool prof = std::getenv("RGR_BULK_PROFILE") != nullptr;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = tnow();
    auto lap = [&](const char* what) { if (prof) { const double tn = tnow(); std::fprintf(stderr, "[retain.compile] %-12s %.3f s\n", what, tn - t_prev); t_prev = tn; } };
    const uint32_t total = uint32_t(nodes_.size());
    // children lists of the mutable ids (counting sort by parent), ordered by token; the
    // root's non-'$' children first (retain.rs:486-490, 505-509 skip '$' children at the root)
    std::vector<uint32_t> cnt(size_t(total) + 1, 0), kids;
    for (const REdge& e : edges_)
        if (e.parent != kEdgeEmpty && e.parent != kEdgeTomb) cnt[e.parent + 1]++;
    for (uint32_t i = 0; i < total; ++i) cnt[i + 1] += cnt[i];
    kids.resize(cnt[total]);
    {
        std::vector<uint32_t> pos(cnt.begin(), cnt.end() - 1);
        for (const REdge& e : edges_)
            if (e.parent != kEdgeEmpty && e.parent != kEdgeTomb) kids[pos[e.parent]++] = e.child;
    }
    for (uint32_t p = 0; p < total; ++p) {
        auto b = kids.begin() + cnt[p], e = kids.begin() + cnt[p + 1];
        if (e - b < 2) continue;
        if (p == 0)
            std::sort(b, e, [&](uint32_t x, uint32_t y) {
                if (nodes_[x].meta != nodes_[y].meta) return !nodes_[x].meta;
                return nodes_[x].token < nodes_[y].token;
            });
        else
            std::sort(b, e, [&](uint32_t x, uint32_t y) { return nodes_[x].token < nodes_[y].token; });
    }
    lap("children");
    // iterative DFS: preorder ids + subtree ends
    const uint32_t N = uint32_t(n_nodes_);
    std::vector<uint32_t> pre(total, kNone), order;   // order[preorder id] = mutable id
    order.reserve(N);
    std::vector<uint32_t> sub_end(N, 0);
    {
        std::vector<std::pair<uint32_t, uint32_t>> st;   // (node, next child index)
        pre[0] = 0;
        order.push_back(0);
        st.emplace_back(0u, cnt[0]);
        while (!st.empty()) {
            auto& top = st.back();
            if (top.second < cnt[top.first + 1]) {
                const uint32_t c = kids[top.second++];
                pre[c] = uint32_t(order.size());
                order.push_back(c);
                st.emplace_back(c, cnt[c]);
            } else {
                sub_end[pre[top.first]] = uint32_t(order.size());
                st.pop_back();
            }
        }
    }
    lap("preorder");
    out.n_nodes = N;
    out.child_off.assign(size_t(N) + 1, 0);
    out.child_ids.clear();
    out.child_ids.reserve(N);
    out.vals.clear();
    for (uint32_t p = 0; p < N; ++p) {
        const uint32_t m = order[p];
        out.child_off[p] = uint32_t(out.child_ids.size());
        for (uint32_t k = cnt[m]; k < cnt[m + 1]; ++k) out.child_ids.push_back(pre[kids[k]]);
    }
    out.child_off[N] = uint32_t(out.child_ids.size());
    out.root_nonmeta = 0;
    for (uint32_t k = cnt[0]; k < cnt[1]; ++k) out.root_nonmeta += !nodes_[kids[k]].meta;
    // ---- value layout.  desc[2p] = p's own value; desc[2p+1] = H(p) = what `p._matches(["#"])`
    // yields (retain.rs:502-524): every value below p — EXCEPT that a node y storing a literal "#"
    // level (child c) answers through the exact-first branch (retain.rs:472-483): below y only c's
    // own value is visible from y or from above.  Values are placed by a DFS that stops at such a
    // y after placing y's and c's values; the rest of subtree(y) goes to a deferred area of its
    // own (still contiguous for every start inside it).  With no literal "#" levels stored this is
    // plain preorder.
    std::vector<uint32_t> own_pos(total, 0), hp_b(total, 0), hp_e(total, 0);
    std::vector<uint32_t> hash_kid;                       // node -> its literal "#" child (rare: allocated on first use)
    for (const REdge& e : edges_)
        if (e.parent != kEdgeEmpty && e.parent != kEdgeTomb && e.token == kTokHash) {
            if (hash_kid.empty()) hash_kid.assign(total, kNone);
            hash_kid[e.parent] = e.child;
        }
    auto hash_child = [&](uint32_t m) -> uint32_t { return hash_kid.empty() ? kNone : hash_kid[m]; };
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
                const uint32_t k = kids[top.second++];
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
    const FilterDesc root_desc = root_barrier ? FilterDesc{hp_b[0], hp_e[0] - hp_b[0]} : FilterDesc{hp_b[0], first_meta_pos - hp_b[0]};
    for (size_t di = 0; di < deferred.size(); ++di) {     // (grows while we iterate)
        const uint32_t y = deferred[di], c = hash_child(y);
        for (uint32_t k = cnt[y]; k < cnt[y + 1]; ++k) {
            const uint32_t ch = kids[k];
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
    // root '#': everything outside the root's '$' subtrees (the non-meta children come first), or the
    // literal "#" topic alone when one is stored
    out.desc[2 * size_t(N)] = root_desc;
    lap("values");
    // grandchild index: (grandparent g, literal token t) -> run of nodes x (preorder ascending)
    {
        struct Tri { uint32_t g, tok, x; };
        std::vector<Tri> tri;
        tri.reserve(N);
        for (uint32_t m = 1; m < total; ++m) {
            if (pre[m] == kNone) continue;                       // freed slot
            const Node& nx = nodes_[m];
            if (nx.token < kTokFirst || nx.parent == 0) continue;   // wildcard-token levels and depth-1 nodes are not indexed
            const uint32_t p = nx.parent, g = nodes_[p].parent;
            if (g == 0 && nodes_[p].meta) continue;              // a '+' at the root skips '$' children (retain.rs:486-490)
            tri.push_back(Tri{pre[g], nx.token, pre[m]});
        }
        {   // order by (g, tok, x): counting sort on the grandparent, then each (small) group by (tok, x)
            std::vector<uint32_t> goff(size_t(N) + 1, 0);
            for (const Tri& t : tri) goff[t.g + 1]++;
            for (uint32_t i = 0; i < N; ++i) goff[i + 1] += goff[i];
            std::vector<Tri> sorted(tri.size());
            {
                std::vector<uint32_t> pos(goff.begin(), goff.end() - 1);
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
    // edge table over preorder ids
    uint64_t cap = 1024;
    while (cap < uint64_t(N) * 2) cap <<= 1;
    out.edges.assign(cap, empty_redge());
    const uint32_t mask = uint32_t(cap - 1);
    for (const REdge& e : edges_) {
        if (e.parent == kEdgeEmpty || e.parent == kEdgeTomb) continue;
        const uint32_t pp = pre[e.parent], pc = pre[e.child];
        uint32_t i = edge_hash(pp, e.token) & mask;
        while (out.edges[i].parent != kEdgeEmpty) i = (i + 1) & mask;
        out.edges[i] = REdge{pp, e.token, pc, 0};
    }
    lap("edge table");
}

}  // namespace rgr
