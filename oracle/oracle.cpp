// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.hpp).  Non-template parts of the
// restatement plus the ctypes-facing C API used by tests/ and bench.py's
// cpu_baseline leg.
#include "oracle.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

namespace orc {

// rmqtt/src/topic.rs:357-377
bool parse_level(std::string_view s, Level& out) {
    if (s == "+") { out = Level{Kind::SingleWildcard, "+"}; return true; }
    if (s == "#") { out = Level{Kind::MultiWildcard, "#"}; return true; }
    if (s.empty()) { out = Level{Kind::Blank, ""}; return true; }
    if (s.find_first_of("+#") != std::string_view::npos) return false;          // InvalidLevel
    if (s[0] == '$') { out = Level{Kind::Metadata, std::string(s)}; return true; }   // topic.rs:66-68
    out = Level{Kind::Normal, std::string(s)};
    return true;
}

// rmqtt/src/topic.rs:379-394 (split on '/') + 231-243 (is_valid)
bool parse_topic(std::string_view s, Topic& out) {
    out.clear();
    size_t start = 0;
    for (;;) {
        size_t pos = s.find('/', start);
        std::string_view seg = s.substr(start, pos == std::string_view::npos ? std::string_view::npos : pos - start);
        Level l;
        if (!parse_level(seg, l)) return false;
        out.push_back(std::move(l));
        if (pos == std::string_view::npos) break;
        start = pos + 1;
    }
    // Level::is_valid holds by construction of parse_level; positional rules:
    for (size_t i = 0; i < out.size(); ++i) {
        if (out[i].kind == Kind::MultiWildcard && i != out.size() - 1) return false;
        if (out[i].kind == Kind::Metadata && i != 0) return false;
    }
    return true;
}

std::string join_levels(const std::vector<const Level*>& levels) {
    std::string r;
    for (size_t i = 0; i < levels.size(); ++i) { if (i) r.push_back('/'); r += levels[i]->s; }
    return r;
}

std::string topic_to_string(const Topic& t) {
    std::string r;
    for (size_t i = 0; i < t.size(); ++i) { if (i) r.push_back('/'); r += t[i].s; }
    return r;
}

// ------------------------------------------------------------ DefaultRouter
bool DefaultRouter::add(std::string_view topic_filter, const Id& id, const SubscriptionOptions& opts, uint32_t rel_id) {
    Topic topic;
    if (!parse_topic(topic_filter, topic)) return false;       // router.rs:436 (`?`)
    ++mutations_;
    topics_.insert(topic, Unit{});                             // router.rs:438
    auto it = relations_.find(std::string(topic_filter));
    if (it == relations_.end()) {                              // router.rs:443-446
        topics_count_++;
        it = relations_.emplace(std::string(topic_filter), FilterEntry{next_filter_id_++, {}, {}}).first;
    }
    auto& rels = it->second.rels;
    auto old = rels.find(id.client_id);
    if (old == rels.end()) { relations_count_++; rels.emplace(id.client_id, Rel{id, opts, rel_id}); }   // router.rs:447-450
    else old->second = Rel{id, opts, rel_id};
    return true;
}

int DefaultRouter::remove(std::string_view topic_filter, const Id& id) {
    auto it = relations_.find(std::string(topic_filter));
    if (it == relations_.end()) return 1;
    auto& rels = it->second.rels;
    auto r = rels.find(id.client_id);
    if (r == rels.end() || !(r->second.id == id)) return 1;    // router.rs:460-467
    rels.erase(r);
    ++mutations_;
    relations_count_--;
    if (rels.empty()) {                                        // router.rs:484-490
        relations_.erase(it);
        topics_count_--;
        Topic topic;
        if (!parse_topic(topic_filter, topic)) return -1;
        topics_.remove(topic, Unit{});
    }
    return 0;
}

uint32_t DefaultRouter::filter_id_of(const std::string& f) const {
    auto it = relations_.find(f);
    return it == relations_.end() ? 0xFFFFFFFFu : it->second.filter_id;
}

namespace {
// types.rs:503-541
struct Collector {
    std::vector<SubRelation> v3_rels;
    std::vector<std::string> v5_order;                       // deterministic stand-in for HashMap order
    std::unordered_map<std::string, SubRelation> v5_rels;
    void add(const std::string& filter, const std::string& client, const SubscriptionOptions& opts, uint32_t rel_id,
             std::optional<SharedGroupInfo> group = std::nullopt) {
        if (opts.is_v3()) {
            v3_rels.push_back(SubRelation{filter, client, opts, std::nullopt, rel_id, std::move(group)});
            return;
        }
        auto it = v5_rels.find(client);
        if (it != v5_rels.end()) {                            // types.rs:526-534
            if (opts.sub_ident) {
                if (it->second.sub_ids) it->second.sub_ids->push_back(opts.sub_ident);
                else it->second.sub_ids = std::vector<uint32_t>{opts.sub_ident};
            }
        } else {                                              // types.rs:535-538
            SubRelation r{filter, client, opts, std::nullopt, rel_id, std::move(group)};
            if (opts.sub_ident) r.sub_ids = std::vector<uint32_t>{opts.sub_ident};
            v5_rels.emplace(client, std::move(r));
            v5_order.push_back(client);
        }
    }
};
}  // namespace

bool DefaultRouter::matches(const Id& this_id, std::string_view topic_name, SubRelationsMap& out, WalkStats* st) const {
    out.clear();
    Topic topic;
    if (!parse_topic(topic_name, topic)) { if (st) st->invalid++; return false; }   // router.rs:177
    if (st) st->levels += topic.size();
    std::map<NodeId, Collector> collector_map;
    for (auto& item : topics_.matches(topic, st)) {             // router.rs:178
        const std::string filter = join_levels(item.first);    // router.rs:179 (to_topic_filter)
        auto rit = relations_.find(filter);                    // router.rs:194
        if (rit == relations_.end()) continue;
        // Reference iteration order over `rels` is unspecified (ahash RandomState);
        // canonicalise by rel_id.
        std::vector<const std::pair<const std::string, Rel>*> ordered;
        for (auto& kv : rit->second.rels) ordered.push_back(&kv);
        std::sort(ordered.begin(), ordered.end(), [](auto* a, auto* b) { return a->second.rel_id < b->second.rel_id; });
        std::map<std::string, std::vector<SharedCandidate>> groups;                       // router.rs:183-192
        for (auto* kv : ordered) {
            const Rel& rel = kv->second;
            auto nl = rel.opts.opt_no_local();
            if (nl && *nl && this_id == rel.id) continue;      // router.rs:196-201
            if (!rel.opts.shared_group.empty()) {              // router.rs:204-213: deferred to the group choice
                groups[rel.opts.shared_group].push_back(SharedCandidate{rel.id.node_id, kv->first, rel.opts, true, rel.rel_id});
                continue;
            }
            collector_map[rel.id.node_id].add(filter, kv->first, rel.opts, rel.rel_id);   // router.rs:214-229
            if (st) st->hits++;
        }
        for (auto& gk : groups) {                              // router.rs:236-255
            auto& subs = gk.second;
            std::vector<std::string> cids;
            for (auto& c : subs) cids.push_back(c.client_id);
            const auto idx = shared_choice_ ? shared_choice_(gk.first, this_id, topic_name, subs) : std::nullopt;
            if (!idx) continue;
            const SharedCandidate& c = subs[*idx];
            collector_map[c.node_id].add(filter, c.client_id, c.opts, c.rel_id, SharedGroupInfo{gk.first, c.is_online, cids});
            if (st) st->hits++;
        }
    }
    for (auto& kv : collector_map) {                            // router.rs:258-261 + types.rs:488-497
        auto& dst = out[kv.first];
        dst = std::move(kv.second.v3_rels);
        for (auto& c : kv.second.v5_order) dst.push_back(std::move(kv.second.v5_rels[c]));
    }
    return true;
}

bool DefaultRouter::match_flat(uint32_t topic_idx, std::string_view topic_name, std::vector<FlatHit>& out, WalkStats* st) const {
    Topic topic;
    if (!parse_topic(topic_name, topic)) { if (st) st->invalid++; return false; }
    if (st) st->levels += topic.size();
    std::vector<FlatHit> tmp;
    for (auto& item : topics_.matches(topic, st)) {
        auto rit = relations_.find(join_levels(item.first));
        if (rit == relations_.end()) continue;
        tmp.clear();
        for (auto& kv : rit->second.rels) {
            const Rel& r = kv.second;
            uint8_t flags = uint8_t((r.opts.v5 ? 1 : 0) | (r.opts.no_local ? 2 : 0));
            tmp.push_back(FlatHit{topic_idx, rit->second.filter_id, r.rel_id, r.opts.qos, flags});
        }
        std::sort(tmp.begin(), tmp.end(), [](const FlatHit& a, const FlatHit& b) { return a.sub_id < b.sub_id; });
        out.insert(out.end(), tmp.begin(), tmp.end());
        if (st) st->hits += tmp.size();
    }
    return true;
}

// ---- O(matched filters) digests (checker only) ------------------------------------------------------
// match_flat lists, per matched filter in TopicTree::matches order, the filter's relations ascending by rel_id; the digest
// of orc_router_match_digest over that list is a sum over hits, so it splits by filter: with `off` hits before filter f,
//   count += n_f;  sum v += S1_f;  sum (k+1) v += off * S1_f + P_f;  sum v^2 += S2_f
// where P_f = sum over f's own list of (j+1) * v_j.  The walk itself (parse, TopicTree::matches, join, relations lookup)
// is the same code match_flat runs.
void DefaultRouter::prepare_digests(int threads) {
    if (digests_at_ == mutations_) return;
    std::vector<FilterEntry*> all;
    all.reserve(relations_.size());
    for (auto& kv : relations_) all.push_back(&kv.second);
    if (threads < 1) threads = 1;
    std::atomic<size_t> next{0};
    auto work = [&] {
        std::vector<uint64_t> v;
        for (;;) {
            const size_t lo = next.fetch_add(4096), hi = std::min(all.size(), lo + 4096);
            if (lo >= all.size()) break;
            for (size_t i = lo; i < hi; ++i) {
                v.clear();
                for (auto& kv : all[i]->rels) v.push_back((uint64_t(kv.second.rel_id) << 8) | kv.second.opts.qos);
                std::sort(v.begin(), v.end());                                  // ascending rel_id (rel ids are unique)
                FilterDigest d;
                for (size_t j = 0; j < v.size(); ++j) {
                    const uint64_t x = (v[j] >> 8) * 4 + (v[j] & 0xFF);
                    d.n++; d.s1 += x; d.p += uint64_t(j + 1) * x; d.s2 += x * x;
                }
                all[i]->dig = d;
            }
        }
    };
    std::vector<std::thread> th;
    for (int k = 0; k < threads; ++k) th.emplace_back(work);
    for (auto& t : th) t.join();
    digests_at_ = mutations_;
}

bool DefaultRouter::match_digest_fast(std::string_view topic_name, uint64_t out[4]) const {
    out[0] = out[1] = out[2] = out[3] = 0;
    Topic topic;
    if (!parse_topic(topic_name, topic)) return false;
    for (auto& item : topics_.matches(topic, nullptr)) {
        auto rit = relations_.find(join_levels(item.first));
        if (rit == relations_.end()) continue;
        const FilterDigest& d = rit->second.dig;
        out[2] += out[0] * d.s1 + d.p;
        out[0] += d.n; out[1] += d.s1; out[3] += d.s2;
    }
    return true;
}

bool DefaultRouter::deliver_digest(const Id& this_id, std::string_view topic_name, uint8_t pub_qos, bool pub_retain, uint64_t out[4]) const {
    out[0] = out[1] = out[2] = out[3] = 0;
    SubRelationsMap m;
    if (!matches(this_id, topic_name, m)) return false;
    std::vector<uint32_t> delivered;
    for (auto& kv : m) for (auto& row : kv.second) delivered.push_back(row.rel_id);
    std::sort(delivered.begin(), delivered.end());
    Topic topic;
    parse_topic(topic_name, topic);
    std::vector<const Rel*> ordered;
    uint64_t k = 0;
    for (auto& item : topics_.matches(topic, nullptr)) {
        auto rit = relations_.find(join_levels(item.first));
        if (rit == relations_.end()) continue;
        ordered.clear();
        for (auto& kv : rit->second.rels) ordered.push_back(&kv.second);
        std::sort(ordered.begin(), ordered.end(), [](const Rel* a, const Rel* b) { return a->rel_id < b->rel_id; });
        for (const Rel* rel : ordered) {
            uint64_t w = std::min<uint8_t>(pub_qos, rel->opts.qos);
            if (rel->opts.v5 && rel->opts.retain_as_published && pub_retain) w |= 4;
            if (!std::binary_search(delivered.begin(), delivered.end(), rel->rel_id))
                w |= (rel->opts.v5 && rel->opts.no_local && this_id == rel->id) ? 8 : 16;
            const uint64_t x = uint64_t(rel->rel_id) * 32 + w;
            ++k;
            out[0]++; out[1] += x; out[2] += k * x; out[3] += x * x;
        }
    }
    return true;
}

// _matches with the collector + forwards_to's per-recipient transform, reference-shaped (container order, ref-counted clones)
uint64_t DefaultRouter::forwards_shaped(const Id& this_id, std::string_view topic_name, uint8_t pub_qos, bool pub_retain, WalkStats* st, ShapedScratch* scratch) const {
    using Out = ShapedScratch::OutFwd;
    Topic topic;
    if (!parse_topic(topic_name, topic)) { if (st) st->invalid++; return 0; }
    if (st) st->levels += topic.size();
    ShapedScratch local;
    auto& cm = (scratch ? scratch : &local)->fwd;
    for (auto& kv : cm) { kv.second.rows.clear(); kv.second.v5_index.clear(); }
    uint64_t hits = 0;
    for (auto& item : topics_.matches(topic, st)) {
        const auto filter = std::make_shared<const std::string>(join_levels(item.first));            // router.rs:179
        auto rit = shaped_.find(*filter);                                                            // router.rs:194
        if (rit == shaped_.end()) continue;
        for (auto& e : rit->second.rels) {
            const Rel& rel = *e.rel;
            ++hits;
            auto nl = rel.opts.opt_no_local();
            if (nl && *nl && this_id == rel.id) continue;                                            // router.rs:196-201
            auto& node = cm[rel.id.node_id];
            if (rel.opts.is_v3()) { node.rows.push_back(Out{filter, e.client, rel.opts, {}, 0, false}); continue; }      // types.rs:519-521
            auto it = node.v5_index.find(std::string_view(*e.client));                                // types.rs:524-539
            if (it != node.v5_index.end()) { if (rel.opts.sub_ident) node.rows[it->second].sub_ids.push_back(rel.opts.sub_ident); continue; }
            node.v5_index.emplace(std::string_view(*e.client), node.rows.size());
            node.rows.push_back(Out{filter, e.client, rel.opts, rel.opts.sub_ident ? std::vector<uint32_t>{rel.opts.sub_ident} : std::vector<uint32_t>{}, 0, false});
        }
    }
    uint64_t rows = 0;
    for (auto& kv : cm)                                                                              // shared.rs:886-908, per recipient
        for (auto& r : kv.second.rows) {
            r.retain = r.opts.v5 ? (r.opts.retain_as_published && pub_retain) : false;
            r.qos = std::min<uint8_t>(pub_qos, r.opts.qos);
            ++rows;
        }
    if (st) st->hits += hits;
    for (auto& kv : cm) { kv.second.rows.clear(); kv.second.v5_index.clear(); }                     // the result is dropped here
    return rows;
}

// Reference-shaped publish match for the cpu_baseline leg (see orc_router_matches_timed): the work
// of router.rs:174-265 per hit, results dropped at the end of the call like the caller's map.
void DefaultRouter::prepare_shaped() {
    if (shaped_ready_ == relations_count_ && !shaped_.empty()) return;
    shaped_.clear();
    for (auto& kv : relations_) {
        auto& dst = shaped_[kv.first];
        for (auto& r : kv.second.rels) dst.rels.push_back(ShapedEntry{std::make_shared<const std::string>(r.first), &r.second});
    }
    shaped_ready_ = relations_count_;
}

uint64_t DefaultRouter::matches_shaped(const Id& this_id, std::string_view topic_name, WalkStats* st, bool refcounted, ShapedScratch* scratch) const {
    if (!refcounted) return matches_shaped_plain(this_id, topic_name, st, scratch);
    using Out = ShapedScratch::OutRc;
    Topic topic;
    if (!parse_topic(topic_name, topic)) { if (st) st->invalid++; return 0; }            // router.rs:177
    if (st) st->levels += topic.size();
    // router.rs:176.  The reference builds a fresh map per call under jemalloc (rmqtt-bin/src/server.rs:27-28), whose
    // thread caches hand the same blocks back without a syscall; under glibc a fresh 0.8 MB vector per publish means
    // mmap/munmap per call and 256 threads queueing on the process's mmap lock.  The per-thread scratch keeps the
    // vectors' capacity between calls (elements are still constructed and destroyed per call).
    ShapedScratch local;
    auto& collector_map = (scratch ? scratch : &local)->rc;
    for (auto& kv : collector_map) kv.second.clear();
    uint64_t hits = 0;
    for (auto& item : topics_.matches(topic, st)) {                                      // router.rs:178
        // router.rs:179: `to_topic_filter()` builds a FRESH ByteString per matched filter per publish, so the
        // per-hit `topic_filter.clone()` (types.rs:520) bumps a refcount private to this call; only the ClientId
        // clone (router.rs:224) touches a counter shared with other threads.
        const auto filter = std::make_shared<const std::string>(join_levels(item.first));
        auto rit = shaped_.find(*filter);                                                // router.rs:194
        if (rit == shaped_.end()) continue;
        for (auto& e : rit->second.rels) {                                               // container order
            const Rel& rel = *e.rel;
            auto nl = rel.opts.opt_no_local();
            if (nl && *nl && this_id == rel.id) continue;                                // router.rs:196-201
            collector_map[rel.id.node_id].push_back(Out{filter, e.client, rel.opts});    // router.rs:222-229
            ++hits;
        }
    }
    if (st) st->hits += hits;
    for (auto& kv : collector_map) kv.second.clear();                                    // the result is dropped here: the clones are released
    return hits;
}

// The same pass with plain pointers instead of ref-counted clones: what the work costs WITHOUT the contended
// atomic increments on hot ClientIds (an upper bound on what a reference build with interned ids could do).
uint64_t DefaultRouter::matches_shaped_plain(const Id& this_id, std::string_view topic_name, WalkStats* st, ShapedScratch* scratch) const {
    using Out = ShapedScratch::OutPlain;
    Topic topic;
    if (!parse_topic(topic_name, topic)) { if (st) st->invalid++; return 0; }
    if (st) st->levels += topic.size();
    ShapedScratch local;
    auto& collector_map = (scratch ? scratch : &local)->plain;
    for (auto& kv : collector_map) kv.second.clear();
    uint64_t hits = 0;
    for (auto& item : topics_.matches(topic, st)) {
        const std::string filter = join_levels(item.first);
        auto rit = shaped_.find(filter);
        if (rit == shaped_.end()) continue;
        for (auto& e : rit->second.rels) {
            const Rel& rel = *e.rel;
            auto nl = rel.opts.opt_no_local();
            if (nl && *nl && this_id == rel.id) continue;
            collector_map[rel.id.node_id].push_back(Out{&rit->first, e.client.get(), rel.opts});
            ++hits;
        }
    }
    if (st) st->hits += hits;
    return hits;
}

bool DefaultRouter::has_matches(std::string_view t) const {
    Topic topic;
    if (!parse_topic(t, topic)) return false;
    return topics_.is_match(topic);
}

std::vector<std::string> DefaultRouter::get_routes(std::string_view t, bool* ok) const {
    std::vector<std::string> r;
    Topic topic;
    *ok = parse_topic(t, topic);
    if (!*ok) return r;
    for (auto& item : topics_.matches(topic)) {                 // .unique(), router.rs:166
        std::string f = join_levels(item.first);
        if (std::find(r.begin(), r.end(), f) == r.end()) r.push_back(std::move(f));
    }
    return r;
}

}  // namespace orc

// =====================================================================  C API
using namespace orc;

namespace {
char* dup_str(const std::string& s) {
    char* p = static_cast<char*>(std::malloc(s.size() + 1));
    std::memcpy(p, s.data(), s.size());
    p[s.size()] = 0;
    return p;
}
template <class T> T* dup_vec(const std::vector<T>& v) {
    T* p = static_cast<T*>(std::malloc(std::max<size_t>(1, v.size()) * sizeof(T)));
    if (!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}
std::string esc(const std::string& s) {   // strings in canonical dumps are newline/tab free in practice
    std::string r;
    for (char c : s) { if (c == '\n') r += "\\n"; else if (c == '\t') r += "\\t"; else r.push_back(c); }
    return r;
}
}  // namespace

extern "C" {

void orc_free(void* p) { std::free(p); }

// ---- parser -------------------------------------------------------------
// Returns number of levels, or -1 on Err.  kinds (if non-null) receives one byte per level.
int orc_parse_topic(const char* s, uint64_t len, uint8_t* kinds, int cap) {
    Topic t;
    if (!parse_topic(std::string_view(s, len), t)) return -1;
    for (size_t i = 0; i < t.size() && int(i) < cap; ++i) kinds[i] = uint8_t(t[i].kind);
    return int(t.size());
}

// ---- TopicTree<u64> (golden vectors) --------------------------------------
void* orc_tree_new() { return new TopicTree<uint64_t>(); }
void orc_tree_free(void* t) { delete static_cast<TopicTree<uint64_t>*>(t); }
int orc_tree_insert(void* t, const char* f, uint64_t len, uint64_t v) {
    Topic tp;
    if (!parse_topic(std::string_view(f, len), tp)) return -1;
    return static_cast<TopicTree<uint64_t>*>(t)->insert(tp, v) ? 1 : 0;
}
int orc_tree_remove(void* t, const char* f, uint64_t len, uint64_t v) {
    Topic tp;
    if (!parse_topic(std::string_view(f, len), tp)) return -1;
    return static_cast<TopicTree<uint64_t>*>(t)->remove(tp, v) ? 1 : 0;
}
uint64_t orc_tree_values_size(void* t) { return static_cast<TopicTree<uint64_t>*>(t)->values_size(); }
uint64_t orc_tree_nodes_size(void* t) { return static_cast<TopicTree<uint64_t>*>(t)->nodes_size(); }
// One line per yielded item, in iterator order: "<filter>\t<v1>,<v2>,...\n".  NULL on parse Err.
char* orc_tree_matches(void* t, const char* topic, uint64_t len) {
    Topic tp;
    if (!parse_topic(std::string_view(topic, len), tp)) return nullptr;
    std::string r;
    for (auto& item : static_cast<TopicTree<uint64_t>*>(t)->matches(tp)) {
        r += esc(join_levels(item.first));
        r.push_back('\t');
        for (size_t i = 0; i < item.second.size(); ++i) { if (i) r.push_back(','); r += std::to_string(*item.second[i]); }
        r.push_back('\n');
    }
    return dup_str(r);
}
int orc_tree_is_match(void* t, const char* topic, uint64_t len) {
    Topic tp;
    if (!parse_topic(std::string_view(topic, len), tp)) return -1;
    return static_cast<TopicTree<uint64_t>*>(t)->is_match(tp) ? 1 : 0;
}

// ---- RetainTree<i64> ------------------------------------------------------
void* orc_retain_new() { return new RetainTree<int64_t>(); }
void orc_retain_digest_table_drop(void* t);
void orc_retain_free(void* t) { orc_retain_digest_table_drop(t); delete static_cast<RetainTree<int64_t>*>(t); }
int orc_retain_insert(void* t, const char* s, uint64_t len, int64_t v) {
    Topic tp;
    if (!parse_topic(std::string_view(s, len), tp)) return -1;
    static_cast<RetainTree<int64_t>*>(t)->insert(tp, v);
    return 0;
}
// 1 = removed (value in *v), 0 = nothing stored, -1 = parse Err
int orc_retain_remove(void* t, const char* s, uint64_t len, int64_t* v) {
    Topic tp;
    if (!parse_topic(std::string_view(s, len), tp)) return -1;
    auto r = static_cast<RetainTree<int64_t>*>(t)->remove(tp);
    if (r && v) *v = *r;
    return r ? 1 : 0;
}
// Removes every value < keep_from (test hook for retain(); mirrors retain.rs:420-447).
uint64_t orc_retain_retain_ge(void* t, uint64_t max_limit, int64_t keep_from) {
    return static_cast<RetainTree<int64_t>*>(t)->retain(size_t(max_limit), [&](int64_t& v) { return v >= keep_from; });
}
uint64_t orc_retain_values_size(void* t) { return static_cast<RetainTree<int64_t>*>(t)->values_size(); }
uint64_t orc_retain_nodes_size(void* t) { return static_cast<RetainTree<int64_t>*>(t)->nodes_size(); }
// One line per hit "<topic>\t<value>\n", sorted by topic string (canonical form, App. A.5).
char* orc_retain_matches(void* t, const char* filter, uint64_t len) {
    Topic tp;
    if (!parse_topic(std::string_view(filter, len), tp)) return nullptr;
    std::vector<std::pair<std::string, int64_t>> rows;
    for (auto& kv : static_cast<RetainTree<int64_t>*>(t)->matches(tp)) rows.emplace_back(topic_to_string(kv.first), kv.second);
    std::sort(rows.begin(), rows.end());
    std::string r;
    for (auto& row : rows) { r += esc(row.first); r.push_back('\t'); r += std::to_string(row.second); r.push_back('\n'); }
    return dup_str(r);
}
// Batch form for the GPU parity tests: values per filter, each filter's list sorted ascending.
// status[i] = 0 ok / -1 parse Err.  Returns total hits; *offsets (n+1) and *values are malloc'ed.
uint64_t orc_retain_match_batch(void* t, const char* blob, const uint64_t* offs, uint64_t n, int32_t* status,
                                uint64_t** offsets, int64_t** values, uint64_t* visited) {
    auto* tree = static_cast<RetainTree<int64_t>*>(t);
    std::vector<uint64_t> off(n + 1, 0);
    std::vector<int64_t> vals;
    WalkStats st;
    for (uint64_t i = 0; i < n; ++i) {
        Topic tp;
        if (!parse_topic(std::string_view(blob + offs[i], offs[i + 1] - offs[i]), tp)) { status[i] = -1; off[i + 1] = vals.size(); continue; }
        status[i] = 0;
        size_t b = vals.size();
        for (auto& kv : tree->matches(tp, &st)) vals.push_back(kv.second);
        std::sort(vals.begin() + b, vals.end());
        off[i + 1] = vals.size();
    }
    *offsets = dup_vec(off); *values = dup_vec(vals);
    if (visited) *visited = st.visited;
    return vals.size();
}

// Bulk insert: topic i -> value ids ? ids[i] : i.  Returns the number of rejected names.
uint64_t orc_retain_insert_bulk(void* t, const char* blob, const uint64_t* offs, uint64_t n, const uint32_t* ids) {
    auto* tree = static_cast<RetainTree<int64_t>*>(t);
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n; ++i) {
        Topic tp;
        if (!parse_topic(std::string_view(blob + offs[i], offs[i + 1] - offs[i]), tp)) { ++bad; continue; }
        tree->insert(tp, ids ? int64_t(ids[i]) : int64_t(i));
    }
    return bad;
}
// Timed multi-thread RetainTree::matches over a batch of filters (cpu_baseline, config 5):
// filters statically partitioned over threads sharing the read-only tree; results discarded.
double orc_retain_match_timed(void* t, const char* blob, const uint64_t* offs, uint64_t n, int threads, uint64_t* hits, uint64_t* visited) {
    auto* tree = static_cast<RetainTree<int64_t>*>(t);
    if (threads < 1) threads = 1;
    std::vector<WalkStats> sts(threads);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int k = 0; k < threads; ++k) {
        th.emplace_back([&, k] {
            uint64_t lo = n * k / threads, hi = n * (k + 1) / threads;
            for (uint64_t i = lo; i < hi; ++i) {
                Topic tp;
                if (!parse_topic(std::string_view(blob + offs[i], offs[i + 1] - offs[i]), tp)) continue;
                auto v = tree->matches(tp, &sts[k]);
                (void)v;
            }
        });
    }
    for (auto& x : th) x.join();
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    WalkStats tot;
    for (auto& s : sts) tot.add(s);
    if (hits) *hits = tot.hits;
    if (visited) *visited = tot.visited;
    return sec;
}

// ---- DefaultRouter ----------------------------------------------------------
struct orc_id {
    uint64_t node_id;
    const char* client_id; uint32_t client_len;
    int64_t create_time;
    uint16_t lid;
};
static Id mk_id(const orc_id* i) {
    Id id;
    id.node_id = i->node_id; id.lid = i->lid; id.create_time = i->create_time;
    id.client_id.assign(i->client_id, i->client_len);
    return id;
}
struct orc_opts { uint8_t v5, qos, no_local, retain_as_published, retain_handling; uint32_t sub_ident; const char* shared_group; uint32_t shared_group_len; };
static SubscriptionOptions mk_opts(const orc_opts* o) {
    SubscriptionOptions s;
    s.v5 = o->v5; s.qos = o->qos; s.no_local = o->no_local; s.retain_as_published = o->retain_as_published;
    s.retain_handling = o->retain_handling; s.sub_ident = o->sub_ident;
    if (o->shared_group && o->shared_group_len) s.shared_group.assign(o->shared_group, o->shared_group_len);
    return s;
}

// "\t$<group>:<online>:<sorted member client ids,>" for the member a shared group selected, "" otherwise
static std::string group_text(const SubRelation& s) {
    if (!s.group) return "";
    auto c = s.group->group_cids;
    std::sort(c.begin(), c.end());
    std::string r = "\t$" + esc(s.group->group) + ":" + std::to_string(int(s.group->is_online)) + ":";
    for (size_t i = 0; i < c.size(); ++i) { if (i) r.push_back(','); r += esc(c[i]); }
    return r;
}

void* orc_router_new() { return new DefaultRouter(); }
// Test policies for SharedSubscription::choice (order-independent): 0 = the reference's default (nobody),
// 1 = the member with the smallest client id, 2 = the member with the largest rel_id.
void orc_router_set_shared_policy(void* r, int policy) {
    auto* rt = static_cast<DefaultRouter*>(r);
    if (policy == 0) { rt->set_shared_choice(nullptr); return; }
    rt->set_shared_choice([policy](const std::string&, const Id&, std::string_view, const std::vector<SharedCandidate>& c) -> std::optional<size_t> {
        if (c.empty()) return std::nullopt;
        size_t best = 0;
        for (size_t i = 1; i < c.size(); ++i)
            if (policy == 1 ? c[i].client_id < c[best].client_id : c[i].rel_id > c[best].rel_id) best = i;
        return best;
    });
}
void orc_router_free(void* r) { delete static_cast<DefaultRouter*>(r); }
int orc_router_add(void* r, const char* f, uint64_t len, const orc_id* id, const orc_opts* o, uint32_t rel_id) {
    return static_cast<DefaultRouter*>(r)->add(std::string_view(f, len), mk_id(id), mk_opts(o), rel_id) ? 0 : -1;
}
int orc_router_remove(void* r, const char* f, uint64_t len, const orc_id* id) {
    return static_cast<DefaultRouter*>(r)->remove(std::string_view(f, len), mk_id(id));
}
int64_t orc_router_topics(void* r) { return static_cast<DefaultRouter*>(r)->topics_count(); }
int64_t orc_router_routes(void* r) { return static_cast<DefaultRouter*>(r)->relations_count(); }
uint64_t orc_router_topics_tree(void* r) { return static_cast<DefaultRouter*>(r)->topics_tree(); }
uint32_t orc_router_filter_id(void* r, const char* f, uint64_t len) {
    return static_cast<DefaultRouter*>(r)->filter_id_of(std::string(f, len));
}

// Bulk add used by the large parity tests / the bench: subscription i = (filter i,
// client "c<client[i]>", node_id 1, v3 opts with qos[i]), rel_id = i.
int orc_router_add_bulk(void* r, const char* blob, const uint64_t* offs, uint64_t n, const uint32_t* client, const uint8_t* qos) {
    auto* rt = static_cast<DefaultRouter*>(r);
    int bad = 0;
    for (uint64_t i = 0; i < n; ++i) {
        Id id; id.node_id = 1; id.client_id = "c" + std::to_string(client[i]);
        SubscriptionOptions o; o.qos = qos[i];
        if (!rt->add(std::string_view(blob + offs[i], offs[i + 1] - offs[i]), id, o, uint32_t(i))) ++bad;
    }
    return bad;
}

struct orc_stats { uint64_t levels, visited, matched, hits, invalid; };

// Bulk add with per-subscription v5 flags (the bench's delivery-stage workload): flags bit 0 = v5, bit 1 = No Local, bit 3 = Retain As
// Published (the RGR_SUB_* bits); Id = { node 1, client "c<client[i]>" }, rel_id = i.
int orc_router_add_bulk_ex(void* r, const char* blob, const uint64_t* offs, uint64_t n, const uint32_t* client, const uint8_t* qos, const uint8_t* flags) {
    auto* rt = static_cast<DefaultRouter*>(r);
    int bad = 0;
    for (uint64_t i = 0; i < n; ++i) {
        Id id; id.node_id = 1; id.client_id = "c" + std::to_string(client[i]);
        SubscriptionOptions o; o.qos = qos[i];
        if (flags) { o.v5 = flags[i] & 1; o.no_local = o.v5 && (flags[i] & 2); o.retain_as_published = o.v5 && (flags[i] & 8); }
        if (!rt->add(std::string_view(blob + offs[i], offs[i + 1] - offs[i]), id, o, uint32_t(i))) ++bad;
    }
    return bad;
}

// Delivery-stage digests (DefaultRouter::deliver_digest) of a batch: publisher of topic i = client pub_client[i] (0xFFFFFFFF: nobody
// the table knows), pub_qr[i] = publish qos | retain << 2.  out: [4 n].
void orc_router_deliver_digest(void* r, const char* blob, const uint64_t* offs, uint64_t n, const uint32_t* pub_client, const uint8_t* pub_qr, int threads,
                               int32_t* status, uint64_t* out) {
    auto* rt = static_cast<DefaultRouter*>(r);
    if (threads < 1) threads = 1;
    std::atomic<uint64_t> next{0};
    auto work = [&] {
        for (;;) {
            const uint64_t lo = next.fetch_add(4), hi = std::min<uint64_t>(n, lo + 4);
            if (lo >= n) break;
            for (uint64_t i = lo; i < hi; ++i) {
                Id id; id.node_id = pub_client[i] == 0xFFFFFFFFu ? 0 : 1; id.client_id = "c" + std::to_string(pub_client[i]);
                status[i] = rt->deliver_digest(id, std::string_view(blob + offs[i], offs[i + 1] - offs[i]), pub_qr[i] & 3, (pub_qr[i] & 4) != 0, out + 4 * i) ? 0 : -1;
            }
        }
    };
    std::vector<std::thread> th;
    for (int k = 0; k < threads; ++k) th.emplace_back(work);
    for (auto& t : th) t.join();
}

// cpu_baseline of the delivery stage: DefaultRouter::forwards_shaped per publish on `threads` threads, chunks of 16 from an atomic cursor.
// stats->hits = relations visited, *rows = rows delivered (after No Local and the v5 collector).
double orc_router_forwards_timed(void* r, const char* blob, const uint64_t* offs, uint64_t n, const uint32_t* pub_client, const uint8_t* pub_qr, int threads,
                                 orc_stats* stats, uint64_t* rows_out) {
    auto* rt = static_cast<DefaultRouter*>(r);
    if (threads < 1) threads = 1;
    rt->prepare_shaped();
    std::vector<WalkStats> sts(threads);
    std::vector<uint64_t> rows(threads, 0);
    std::atomic<uint64_t> next{0};
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int k = 0; k < threads; ++k) {
        th.emplace_back([&, k] {
            DefaultRouter::ShapedScratch scratch;
            for (;;) {
                const uint64_t lo = next.fetch_add(16), hi = std::min<uint64_t>(n, lo + 16);
                if (lo >= n) break;
                for (uint64_t i = lo; i < hi; ++i) {
                    Id id; id.node_id = pub_client[i] == 0xFFFFFFFFu ? 0 : 1; id.client_id = "c" + std::to_string(pub_client[i]);
                    rows[k] += rt->forwards_shaped(id, std::string_view(blob + offs[i], offs[i + 1] - offs[i]), pub_qr[i] & 3, (pub_qr[i] & 4) != 0, &sts[k], &scratch);
                }
            }
        });
    }
    for (auto& t : th) t.join();
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    WalkStats tot;
    for (auto& s : sts) tot.add(s);
    if (stats) *stats = orc_stats{tot.levels, tot.visited, tot.matched, tot.hits, tot.invalid};
    if (rows_out) { *rows_out = 0; for (auto x : rows) *rows_out += x; }
    return sec;
}

// Canonical dump of DefaultRouter::matches (App. A.5): per node id ascending,
//   "N <node>\n" then "3 <filter>\t<client>\t<qos>\t<rel_id>\n" rows sorted, then
//   "5 <client>\t<first filter>\t<qos>\t<nl>\t<sorted sub ids,>\n" rows sorted by client.
// NULL => Err.
char* orc_router_matches(void* r, const orc_id* this_id, const char* topic, uint64_t len) {
    SubRelationsMap m;
    if (!static_cast<DefaultRouter*>(r)->matches(mk_id(this_id), std::string_view(topic, len), m)) return nullptr;
    std::string out;
    for (auto& kv : m) {
        out += "N " + std::to_string(kv.first) + "\n";
        std::vector<std::string> v3, v5;
        for (auto& s : kv.second) {
            if (s.opts.is_v3()) {
                v3.push_back("3 " + esc(s.topic_filter) + "\t" + esc(s.client_id) + "\t" + std::to_string(s.opts.qos) + "\t" + std::to_string(s.rel_id) + group_text(s) + "\n");
            } else {
                std::string ids;
                if (s.sub_ids) {
                    auto v = *s.sub_ids; std::sort(v.begin(), v.end());
                    for (size_t i = 0; i < v.size(); ++i) { if (i) ids.push_back(','); ids += std::to_string(v[i]); }
                } else ids = "-";
                v5.push_back("5 " + esc(s.client_id) + "\t" + esc(s.topic_filter) + "\t" + std::to_string(s.opts.qos) + "\t" + std::to_string(int(s.opts.no_local)) + "\t" + ids + group_text(s) + "\n");
            }
        }
        std::sort(v3.begin(), v3.end()); std::sort(v5.begin(), v5.end());
        for (auto& s : v3) out += s;
        for (auto& s : v5) out += s;
    }
    return dup_str(out);
}

// forwards_to (rmqtt/src/shared.rs:876-903) applied to the SubRelationsMap of one publish: what
// each recipient is sent.  Per relation: retain = opts.retain_as_published() (Some only for v5,
// types.rs) ? (rap && publish.retain) : false (shared.rs:886-897); qos = publish.qos.less_value(
// opts.qos()) (shared.rs:902); subscription_ids = the collector's list (shared.rs:904-908).
//   "N <node>\n" then "<client>\t<filter>\t<qos'>\t<retain'>\t<sub ids in collected order,|->\n" rows sorted.
// NULL => Err (callers use an empty map, shared.rs:774-777).
char* orc_router_forwards(void* r, const orc_id* this_id, const char* topic, uint64_t len, uint8_t pub_qos, uint8_t pub_retain) {
    SubRelationsMap m;
    if (!static_cast<DefaultRouter*>(r)->matches(mk_id(this_id), std::string_view(topic, len), m)) return nullptr;
    std::string out;
    for (auto& kv : m) {
        out += "N " + std::to_string(kv.first) + "\n";
        std::vector<std::string> rows;
        for (auto& s : kv.second) {
            const bool retain = s.opts.v5 ? (s.opts.retain_as_published && pub_retain) : false;
            const uint8_t qos = std::min<uint8_t>(pub_qos, s.opts.qos);
            std::string ids;
            if (s.sub_ids) { for (size_t i = 0; i < s.sub_ids->size(); ++i) { if (i) ids.push_back(','); ids += std::to_string((*s.sub_ids)[i]); } }
            else ids = "-";
            rows.push_back(esc(s.client_id) + "\t" + esc(s.topic_filter) + "\t" + std::to_string(qos) + "\t" + std::to_string(int(retain)) + "\t" + ids + "\n");
        }
        std::sort(rows.begin(), rows.end());
        for (auto& x : rows) out += x;
    }
    return dup_str(out);
}

// Flat id-level match of a batch (single thread): status[n] (0 / -1), hit_offsets[n+1],
// and per hit filter_id / sub_id / qos / flags.  Arrays are malloc'ed (orc_free).
uint64_t orc_router_match_flat(void* r, const char* blob, const uint64_t* offs, uint64_t n, int32_t* status,
                               uint64_t** hit_offsets, uint32_t** filter_ids, uint32_t** sub_ids, uint8_t** qos,
                               uint8_t** flags, orc_stats* stats) {
    auto* rt = static_cast<DefaultRouter*>(r);
    std::vector<FlatHit> hits;
    std::vector<uint64_t> off(n + 1, 0);
    WalkStats st;
    for (uint64_t i = 0; i < n; ++i) {
        bool ok = rt->match_flat(uint32_t(i), std::string_view(blob + offs[i], offs[i + 1] - offs[i]), hits, &st);
        status[i] = ok ? 0 : -1;
        off[i + 1] = hits.size();
    }
    std::vector<uint32_t> f(hits.size()), s(hits.size());
    std::vector<uint8_t> q(hits.size()), fl(hits.size());
    for (size_t i = 0; i < hits.size(); ++i) { f[i] = hits[i].filter_id; s[i] = hits[i].sub_id; q[i] = hits[i].qos; fl[i] = hits[i].flags; }
    *hit_offsets = dup_vec(off); *filter_ids = dup_vec(f); *sub_ids = dup_vec(s); *qos = dup_vec(q); *flags = dup_vec(fl);
    if (stats) *stats = orc_stats{st.levels, st.visited, st.matched, st.hits, st.invalid};
    return hits.size();
}

// Timed multi-thread match (the cpu_baseline leg): topics statically partitioned over
// `threads` threads sharing the read-only table (mirrors tokio tasks under the trie's
// RwLock read guard, router.rs:178).  Includes per-topic parsing, as the reference does
// (router.rs:177).  Hits are produced into per-thread vectors and discarded.
double orc_router_match_timed(void* r, const char* blob, const uint64_t* offs, uint64_t n, int threads, orc_stats* stats) {
    auto* rt = static_cast<DefaultRouter*>(r);
    if (threads < 1) threads = 1;
    std::vector<WalkStats> sts(threads);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int k = 0; k < threads; ++k) {
        th.emplace_back([&, k] {
            std::vector<FlatHit> hits;
            uint64_t lo = n * k / threads, hi = n * (k + 1) / threads;
            for (uint64_t i = lo; i < hi; ++i) {
                hits.clear();
                rt->match_flat(uint32_t(i), std::string_view(blob + offs[i], offs[i + 1] - offs[i]), hits, &sts[k]);
            }
        });
    }
    for (auto& t : th) t.join();
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    WalkStats tot;
    for (auto& s : sts) tot.add(s);
    if (stats) *stats = orc_stats{tot.levels, tot.visited, tot.matched, tot.hits, tot.invalid};
    return sec;
}


// ---- per-topic digests (bench.py `parity_sample`, full-size GPU tests) ---------------------------
// The flat result of 10^5 topics at config-3 fan-out is ~10^9 hits: too big to hand over, so the
// checker compares digests instead.  Per topic four u64 (arithmetic mod 2^64):
//   [0] hit count
//   [1] sum of v          where v = sub_id * 4 + qos            (order-independent)
//   [2] sum of (k+1) * v  with k = position of the hit inside the topic's list, in the canonical
//                         order of SURVEY.md App. A.5 (filters in TopicTree::matches order, sub_id
//                         ascending inside a filter)             (order-DEPENDENT)
//   [3] sum of v * v                                             (order-independent)
// Invalid topics give zeros and status -1.  Threads share the read-only table, dynamic chunks.
void orc_router_match_digest(void* r, const char* blob, const uint64_t* offs, uint64_t n, int threads, int32_t* status, uint64_t* out) {
    auto* rt = static_cast<DefaultRouter*>(r);
    if (threads < 1) threads = 1;
    std::atomic<uint64_t> next{0};
    auto work = [&] {
        std::vector<FlatHit> hits;
        for (;;) {
            const uint64_t lo = next.fetch_add(16), hi = std::min<uint64_t>(n, lo + 16);
            if (lo >= n) break;
            for (uint64_t i = lo; i < hi; ++i) {
                hits.clear();
                const bool ok = rt->match_flat(uint32_t(i), std::string_view(blob + offs[i], offs[i + 1] - offs[i]), hits, nullptr);
                status[i] = ok ? 0 : -1;
                uint64_t s1 = 0, s2 = 0, s3 = 0;
                for (size_t k = 0; k < hits.size(); ++k) {
                    const uint64_t v = uint64_t(hits[k].sub_id) * 4 + hits[k].qos;
                    s1 += v; s2 += uint64_t(k + 1) * v; s3 += v * v;
                }
                out[4 * i] = hits.size(); out[4 * i + 1] = s1; out[4 * i + 2] = s2; out[4 * i + 3] = s3;
            }
        }
    };
    std::vector<std::thread> th;
    for (int k = 0; k < threads; ++k) th.emplace_back(work);
    for (auto& t : th) t.join();
}

// The same digests in O(matched filters) per topic (DefaultRouter::match_digest_fast): what lets bench.py compare EVERY topic
// of a 10 M-publish batch.  tests/test_oracle_digest.py holds it equal to orc_router_match_digest.
void orc_router_match_digest_fast(void* r, const char* blob, const uint64_t* offs, uint64_t n, int threads, int32_t* status, uint64_t* out) {
    auto* rt = static_cast<DefaultRouter*>(r);
    if (threads < 1) threads = 1;
    rt->prepare_digests(threads);
    std::atomic<uint64_t> next{0};
    auto work = [&] {
        for (;;) {
            const uint64_t lo = next.fetch_add(256), hi = std::min<uint64_t>(n, lo + 256);
            if (lo >= n) break;
            for (uint64_t i = lo; i < hi; ++i)
                status[i] = rt->match_digest_fast(std::string_view(blob + offs[i], offs[i + 1] - offs[i]), out + 4 * i) ? 0 : -1;
        }
    };
    std::vector<std::thread> th;
    for (int k = 0; k < threads; ++k) th.emplace_back(work);
    for (auto& t : th) t.join();
}

// ---- RetainTree::matches digests in O(visited nodes that are not under a '#') ------------------------------------------
// A filter that ends in '#' returns a whole subtree (retain.rs:502-524), which costs the walk O(subtree).  The digest is a sum
// over the returned values, so the '#' step can use a bottom-up aggregate instead.  For a node n with branches, let
//   H(n) = digest of what RetainTree::walk(n, path = [.., '#'], i = last) emits when it is entered from its parent
// following retain.rs literally:
//   * n has a child stored under the literal key '#' (only possible for retained topic names that contain a '#' level; the
//     exact-child branch, retain.rs:472-482, comes first): the walk descends THAT child with the path exhausted, which
//     emits the child's own value and nothing else                              => H(n) = value('#'-child)
//   * otherwise (retain.rs:502-524) every child c contributes its value, and H(c) when c has branches
//                                                                                => H(n) = sum_c value(c) + [c has branches] H(c)
// (The '$'-children are skipped only at the ROOT, retain.rs:505-509: the root is handled at query time, H(root) is never used.)
// Everything that is not a '#' step (exact children, '+', parent matches, path end) is walked exactly as RetainTree::walk does.
namespace {
struct RAgg { uint64_t c = 0, s1 = 0, s2 = 0;
    void add(const RAgg& o) { c += o.c; s1 += o.s1; s2 += o.s2; }
    void addv(int64_t v) { const uint64_t x = uint64_t(v); c++; s1 += x; s2 += x * x; } };
using RNode = RetainTree<int64_t>;
struct RetainAggTable {
    std::unordered_map<const RNode*, RAgg> h;
    uint64_t nodes_at = ~0ull, values_at = ~0ull;
    RAgg build(const RNode* n) {
        RAgg a;
        if (n->branches.empty()) return a;
        static const Level multi{Kind::MultiWildcard, "#"};
        auto lit = n->branches.find(multi);
        for (auto& kv : n->branches) {
            const RAgg sub = build(kv.second.get());                              // (children are always built: they are query entry points too)
            if (lit != n->branches.end()) continue;
            if (kv.second->value) a.addv(*kv.second->value);
            if (!kv.second->branches.empty()) a.add(sub);
        }
        if (lit != n->branches.end() && lit->second->value) a.addv(*lit->second->value);
        h.emplace(n, a);
        return a;
    }
};
std::mutex g_ragg_mu;
std::unordered_map<const RNode*, std::unique_ptr<RetainAggTable>> g_ragg;       // per tree root (test infrastructure: trees are few)

void retain_walk_digest(const RetainAggTable& T, const RNode* n, const Topic& path, size_t i, bool at_root, RAgg& out) {
    const size_t rem = path.size() - i;
    if (n->branches.empty() || rem == 0) {                                        // retain.rs:464-470
        if (rem == 0 && n->value) out.addv(*n->value);
        return;
    }
    const bool next_multi = rem > 1 && path[i + 1].kind == Kind::MultiWildcard;
    auto e = n->branches.find(path[i]);
    if (e != n->branches.end()) {                                                 // retain.rs:472-482
        if (next_multi && e->second->value) out.addv(*e->second->value);
        retain_walk_digest(T, e->second.get(), path, i + 1, false, out);
    } else if (path[i].kind == Kind::SingleWildcard) {                            // retain.rs:483-501
        for (auto& kv : n->branches) {
            if (at_root && kv.first.kind != Kind::Blank && kv.first.is_metadata()) continue;
            if (next_multi && kv.second->value) out.addv(*kv.second->value);
            retain_walk_digest(T, kv.second.get(), path, i + 1, false, out);
        }
    } else if (path[i].kind == Kind::MultiWildcard) {                             // retain.rs:502-524, through the aggregates
        if (!at_root) { out.add(T.h.at(n)); return; }
        for (auto& kv : n->branches) {
            if (kv.first.kind != Kind::Blank && kv.first.is_metadata()) continue;
            if (kv.second->value) out.addv(*kv.second->value);
            if (!kv.second->branches.empty()) out.add(T.h.at(kv.second.get()));
        }
    }
}
}  // namespace

// Invalidate / rebuild: the table is rebuilt when the tree's node or value count changed since it was built (the checker
// builds its tree once and then only queries it; a same-count mutation between two digest calls is not supported).
void orc_retain_match_digest_fast(void* t, const char* blob, const uint64_t* offs, uint64_t n, int threads, int32_t* status, uint64_t* out) {
    auto* tree = static_cast<RNode*>(t);
    if (threads < 1) threads = 1;
    RetainAggTable* T;
    {
        std::lock_guard<std::mutex> lk(g_ragg_mu);
        auto& slot = g_ragg[tree];
        if (!slot) slot = std::make_unique<RetainAggTable>();
        T = slot.get();
        const uint64_t nn = tree->nodes_size(), nv = tree->values_size();
        if (T->nodes_at != nn || T->values_at != nv) { T->h.clear(); T->h.reserve(nn / 2 + 16); T->build(tree); T->nodes_at = nn; T->values_at = nv; }
    }
    std::atomic<uint64_t> next{0};
    auto work = [&] {
        for (;;) {
            const uint64_t lo = next.fetch_add(64), hi = std::min<uint64_t>(n, lo + 64);
            if (lo >= n) break;
            for (uint64_t i = lo; i < hi; ++i) {
                Topic tp;
                out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = 0;
                if (!parse_topic(std::string_view(blob + offs[i], offs[i + 1] - offs[i]), tp)) { status[i] = -1; continue; }
                status[i] = 0;
                RAgg a;
                retain_walk_digest(*T, tree, tp, 0, true, a);
                out[3 * i] = a.c; out[3 * i + 1] = a.s1; out[3 * i + 2] = a.s2;
            }
        }
    };
    std::vector<std::thread> th;
    for (int k = 0; k < threads; ++k) th.emplace_back(work);
    for (auto& x : th) x.join();
}
void orc_retain_digest_table_drop(void* t) { std::lock_guard<std::mutex> lk(g_ragg_mu); g_ragg.erase(static_cast<RNode*>(t)); }

// RetainTree::matches digests, per filter three u64: [0] hits, [1] sum of ids, [2] sum of id * id.
// Order-independent only: the reference's own order is hash-map order (retain.rs:485, 504).
void orc_retain_match_digest(void* t, const char* blob, const uint64_t* offs, uint64_t n, int threads, int32_t* status, uint64_t* out) {
    auto* tree = static_cast<RetainTree<int64_t>*>(t);
    if (threads < 1) threads = 1;
    std::atomic<uint64_t> next{0};
    auto work = [&] {
        for (;;) {
            const uint64_t lo = next.fetch_add(4), hi = std::min<uint64_t>(n, lo + 4);
            if (lo >= n) break;
            for (uint64_t i = lo; i < hi; ++i) {
                Topic tp;
                out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = 0;
                if (!parse_topic(std::string_view(blob + offs[i], offs[i + 1] - offs[i]), tp)) { status[i] = -1; continue; }
                status[i] = 0;
                uint64_t c = 0, s1 = 0, s2 = 0;
                for (auto& kv : tree->matches(tp)) { const uint64_t v = uint64_t(kv.second); c++; s1 += v; s2 += v * v; }
                out[3 * i] = c; out[3 * i + 1] = s1; out[3 * i + 2] = s2;
            }
        }
    };
    std::vector<std::thread> th;
    for (int k = 0; k < threads; ++k) th.emplace_back(work);
    for (auto& x : th) x.join();
}

// ---- cpu_baseline, reference-shaped -----------------------------------------------------------------
// Times what DefaultRouter::_matches does per publish (router.rs:174-265) and nothing else: parse the
// topic (:177), walk the trie (:178), re-join each matched filter (:179), look it up in `relations`
// (:194), iterate its relation map in CONTAINER order (no canonicalising sort — that is test
// infrastructure), No-Local check (:196-201), and per hit the collector's push of (filter, client,
// opts) (:222-229, types.rs:510-521).  The reference's per-hit clones of TopicFilter / ClientId are
// ByteString clones = atomic refcount bumps; the stand-in is a std::shared_ptr copy, which costs the
// same atomic increment (and the matching decrement when the result map is dropped).  Threads model
// tokio workers under the trie's RwLock read guard; topics are handed out in chunks of 16 from an
// atomic cursor, so the Zipf hit distribution cannot strand a thread with the heavy topics.
double orc_router_matches_timed(void* r, const char* blob, const uint64_t* offs, uint64_t n, int threads, int refcounted, orc_stats* stats) {
    auto* rt = static_cast<DefaultRouter*>(r);
    if (threads < 1) threads = 1;
    rt->prepare_shaped();
    std::vector<WalkStats> sts(threads);
    std::atomic<uint64_t> next{0};
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int k = 0; k < threads; ++k) {
        th.emplace_back([&, k] {
            Id nobody; nobody.node_id = 0;
            DefaultRouter::ShapedScratch scratch;
            for (;;) {
                const uint64_t lo = next.fetch_add(16), hi = std::min<uint64_t>(n, lo + 16);
                if (lo >= n) break;
                for (uint64_t i = lo; i < hi; ++i)
                    rt->matches_shaped(nobody, std::string_view(blob + offs[i], offs[i + 1] - offs[i]), &sts[k], refcounted != 0, &scratch);
            }
        });
    }
    for (auto& t : th) t.join();
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    WalkStats tot;
    for (auto& s : sts) tot.add(s);
    if (stats) *stats = orc_stats{tot.levels, tot.visited, tot.matched, tot.hits, tot.invalid};
    return sec;
}

// Dynamic-chunk variant of orc_retain_match_timed (same work per filter).
double orc_retain_match_timed_dyn(void* t, const char* blob, const uint64_t* offs, uint64_t n, int threads, uint64_t* hits, uint64_t* visited) {
    auto* tree = static_cast<RetainTree<int64_t>*>(t);
    if (threads < 1) threads = 1;
    std::vector<WalkStats> sts(threads);
    std::atomic<uint64_t> next{0};
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int k = 0; k < threads; ++k) {
        th.emplace_back([&, k] {
            for (;;) {
                const uint64_t i = next.fetch_add(1);
                if (i >= n) break;
                Topic tp;
                if (!parse_topic(std::string_view(blob + offs[i], offs[i + 1] - offs[i]), tp)) continue;
                auto v = tree->matches(tp, &sts[k]);
                (void)v;
            }
        });
    }
    for (auto& x : th) x.join();
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    WalkStats tot;
    for (auto& s : sts) tot.add(s);
    if (hits) *hits = tot.hits;
    if (visited) *visited = tot.visited;
    return sec;
}

}  // extern "C"
