// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product.
//
// CPU restatement (C++17) of the rmqtt publish-time topic matching path, written
// from the behaviour of the reference sources (no code copied; the reference is
// Rust).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// may include, link or call anything under oracle/; the product library
// (rmqtt_amd/csrc) never does.
//
// Parity pinning: the reference cannot be compiled here (no rustc/cargo), so this
// restatement is pinned against every known-answer vector the reference's own unit
// tests hold for the path (rmqtt/src/trie.rs:443-541, rmqtt/src/retain.rs:608-641,
// rmqtt/src/topic.rs:460-617) — see tests/test_oracle_golden.py — and against an
// independent brute-force pairwise matcher (oracle/brute.py).
//
// Restated items (reference file:line):
//   Level / Level::from_str          rmqtt/src/topic.rs:97-103, 357-377
//   Topic::from_str / is_valid       rmqtt/src/topic.rs:379-394, 231-243
//   TopicTree insert/remove/matches  rmqtt/src/trie.rs:113-149, 301-409
//   DefaultRouter add/remove/_matches rmqtt/src/router.rs:434-496, 174-265
//   SubscriptioRelationsCollector    rmqtt/src/types.rs:488-541
//   RetainTree insert/remove/retain/matches  rmqtt/src/retain.rs:373-447, 450-526
//
// Container shapes deliberately mirror the reference (hash map per trie node keyed
// by the level, ordered value set, string-keyed relations map) so that this is
// also an honest single-/multi-thread CPU baseline ("port" in bench.py).
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <set>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

namespace orc {

// ---------------------------------------------------------------- topic.rs
enum class Kind : uint8_t { Normal, Metadata, Blank, SingleWildcard, MultiWildcard };

// rmqtt/src/topic.rs:97-103.  Equality/hash are derived on (variant, string); the
// variant is a pure function of the string for every level the parser can produce,
// so the string alone is a faithful key.
struct Level {
    Kind kind;
    std::string s;   // textual form ("+", "#", "" for the three non-string kinds)
    bool operator==(const Level& o) const { return kind == o.kind && s == o.s; }
    bool is_metadata() const { return kind == Kind::Metadata; }
};

struct LevelHash {
    size_t operator()(const Level& l) const { return std::hash<std::string>()(l.s) ^ (size_t(l.kind) << 1); }
};

using Topic = std::vector<Level>;

// Level::from_str, rmqtt/src/topic.rs:357-377.  Returns false on InvalidLevel.
bool parse_level(std::string_view s, Level& out);
// Topic::from_str + is_valid, rmqtt/src/topic.rs:379-394, 231-243.  false => Err.
bool parse_topic(std::string_view s, Topic& out);
std::string join_levels(const std::vector<const Level*>& levels);   // trie.rs:251-256
std::string topic_to_string(const Topic& t);                          // topic.rs:407-423

// Walk statistics that feed SURVEY.md §8(d)'s algorithmic-bytes formula.
struct WalkStats {
    uint64_t levels = 0;     // Σ L_t over valid topics
    uint64_t visited = 0;    // nV: MatchedIter instantiations (trie.rs:235,361,369)
    uint64_t matched = 0;    // nM: yielded items
    uint64_t hits = 0;       // nH: emitted subscriber tuples
    uint64_t invalid = 0;    // topics rejected by the parser
    void add(const WalkStats& o) {
        levels += o.levels; visited += o.visited; matched += o.matched; hits += o.hits; invalid += o.invalid;
    }
};

// ---------------------------------------------------------------- trie.rs
// The router instantiates TopicTree<()>: BTreeSet<()> holds at most one element.
struct Unit {
    bool operator<(const Unit&) const { return false; }
};

template <class V> struct TopicTree {   // = Node<V>, rmqtt/src/trie.rs:84-87
    std::set<V> values;
    std::unordered_map<Level, std::unique_ptr<TopicTree>, LevelHash> branches;

    // trie.rs:113-126 — true iff the value was newly inserted.
    bool insert(const Topic& filter, const V& v) {
        TopicTree* n = this;
        for (const Level& l : filter) {
            auto& slot = n->branches[l];
            if (!slot) slot = std::make_unique<TopicTree>();
            n = slot.get();
        }
        return n->values.insert(v).second;
    }
    // trie.rs:129-149 — remove + prune children left without values and branches.
    bool remove(const Topic& filter, const V& v) { return remove_at(filter, 0, v); }

    using Item = std::pair<std::vector<const Level*>, std::vector<const V*>>;
    // trie.rs:157-159 + MatchedIter (trie.rs:301-409), evaluated eagerly in the
    // exact order the lazy iterator yields.
    std::vector<Item> matches(const Topic& topic, WalkStats* st = nullptr) const {
        std::vector<Item> out;
        std::vector<const Level*> sub_path;
        walk(topic, 0, sub_path, out, st);
        return out;
    }
    bool is_match(const Topic& topic) const { return !matches(topic).empty(); }   // trie.rs:152-154
    size_t values_size() const {                                                    // trie.rs:162-165
        size_t n = values.size();
        for (auto& kv : branches) n += kv.second->values_size();
        return n;
    }
    size_t nodes_size() const {                                                     // trie.rs:168-171
        size_t n = branches.size();
        for (auto& kv : branches) n += kv.second->nodes_size();
        return n;
    }

   private:
    bool remove_at(const Topic& f, size_t i, const V& v) {
        if (i == f.size()) return values.erase(v) > 0;
        auto it = branches.find(f[i]);
        if (it == branches.end()) return false;
        bool res = it->second->remove_at(f, i + 1, v);
        if (it->second->values.empty() && it->second->branches.empty()) branches.erase(it);
        return res;
    }
    static const Level& multi() { static const Level l{Kind::MultiWildcard, "#"}; return l; }
    static const Level& single() { static const Level l{Kind::SingleWildcard, "+"}; return l; }
    static void push_item(std::vector<Item>& out, std::vector<const Level*> levels, const std::set<V>& vs, WalkStats* st) {
        if (vs.empty()) return;                        // add_to_items, trie.rs:306-310
        std::vector<const V*> vals;
        for (const V& v : vs) vals.push_back(&v);
        out.emplace_back(std::move(levels), std::move(vals));
        if (st) st->matched++;
    }
    void walk(const Topic& path, size_t i, std::vector<const Level*>& sub_path, std::vector<Item>& out, WalkStats* st) const {
        if (st) st->visited++;
        if (i == path.size()) {
            // trie.rs:328-338: '#'-child pushed first, own values second, popped LIFO
            // (trie.rs:313-316) => own values are yielded first.
            push_item(out, sub_path, values, st);
            auto h = branches.find(multi());
            if (h != branches.end()) {
                auto sp = sub_path; sp.push_back(&h->first);
                push_item(out, std::move(sp), h->second->values, st);
            }
            return;
        }
        const bool at_root = sub_path.empty();
        // trie.rs:342-346: '$'-topics are isolated from wildcards at the root only.
        const bool skip_wild = at_root && path[i].kind != Kind::Blank && path[i].is_metadata() &&
                               (branches.count(multi()) || branches.count(single()));
        const TopicTree* plus = nullptr; const Level* plus_key = nullptr;
        if (!skip_wild) {
            auto h = branches.find(multi());                 // trie.rs:349-355
            if (h != branches.end()) {
                auto sp = sub_path; sp.push_back(&h->first);
                push_item(out, std::move(sp), h->second->values, st);
            }
            auto p = branches.find(single());                // trie.rs:358-362
            if (p != branches.end()) { plus = p->second.get(); plus_key = &p->first; }
        }
        if (plus) {
            sub_path.push_back(plus_key);
            plus->walk(path, i + 1, sub_path, out, st);
            sub_path.pop_back();
        }
        auto e = branches.find(path[i]);                     // trie.rs:366-370
        if (e != branches.end()) {
            sub_path.push_back(&e->first);
            e->second->walk(path, i + 1, sub_path, out, st);
            sub_path.pop_back();
        }
    }
};

// ---------------------------------------------------------------- types.rs
using NodeId = uint64_t;

// types.rs:1899-1911; equality over every field (types.rs:1841-1851).
struct Id {
    NodeId node_id = 0;
    uint16_t lid = 0;
    std::string local_addr, remote_addr;   // Option<SocketAddr> as text ("" = None)
    std::string client_id;
    std::string username;                  // "" = None
    int64_t create_time = 0;
    bool operator==(const Id& o) const {
        return node_id == o.node_id && lid == o.lid && client_id == o.client_id && local_addr == o.local_addr &&
               remote_addr == o.remote_addr && username == o.username && create_time == o.create_time;
    }
};

// types.rs:607-827, restricted to what the matching path reads.
struct SubscriptionOptions {
    bool v5 = false;
    uint8_t qos = 0;
    bool no_local = false;                 // v5 only
    bool retain_as_published = false;      // v5 only (carried, not interpreted here)
    uint8_t retain_handling = 0;           // v5 only (carried)
    uint32_t sub_ident = 0;                // v5 subscription identifier, 0 = None
    std::string shared_group;              // "" = None (types.rs:776, 815): the <group> of $share/<group>/<filter>
    std::optional<bool> opt_no_local() const { return v5 ? std::optional<bool>(no_local) : std::nullopt; }
    bool is_v3() const { return !v5; }
};

// SharedGroupType (types.rs:474): (group, is_online of the chosen member, client ids of the whole group)
struct SharedGroupInfo { std::string group; bool is_online = true; std::vector<std::string> group_cids; };
// SubRelation (types.rs:478-484)
struct SubRelation {
    std::string topic_filter;
    std::string client_id;
    SubscriptionOptions opts;
    std::optional<std::vector<uint32_t>> sub_ids;
    uint32_t rel_id = 0;    // test-only: dense relation id registered with add()
    std::optional<SharedGroupInfo> group;   // Some for the member SharedSubscription::choice picked
};
// One candidate handed to SharedSubscription::choice (subscribe.rs:80-95): (node, client, opts, is_online)
struct SharedCandidate { NodeId node_id; std::string client_id; SubscriptionOptions opts; bool is_online; uint32_t rel_id; };
// choice(group, publisher, topic, candidates) -> index or nullopt.  The reference's default
// (DefaultSharedSubscription, subscribe.rs:107) returns None: without the shared-subscription plugin no
// member receives the publish.  Test policies must not depend on the candidates' order (it is hash-map order).
using SharedChoice = std::function<std::optional<size_t>(const std::string&, const Id&, std::string_view, const std::vector<SharedCandidate>&)>;
using SubRelationsMap = std::map<NodeId, std::vector<SubRelation>>;

// ---------------------------------------------------------------- router.rs
struct FlatHit { uint32_t topic_idx, filter_id, sub_id; uint8_t qos, flags; };

class DefaultRouter {   // router.rs:121-127
   public:
    // router.rs:434-453.  rel_id/filter_id are the dense ids the host glue would
    // hand to the C ABI for this relation / filter (test bookkeeping only).
    bool add(std::string_view topic_filter, const Id& id, const SubscriptionOptions& opts, uint32_t rel_id = 0);
    // router.rs:456-496.  0 = removed, 1 = not removed, -1 = Err (parse error).
    int remove(std::string_view topic_filter, const Id& id);
    // router.rs:174-265.  false => Err (invalid topic name).
    bool matches(const Id& this_id, std::string_view topic_name, SubRelationsMap& out, WalkStats* st = nullptr) const;
    // SharedSubscription::choice used by matches() for $share members (router.rs:236-255); unset = the
    // reference's default, which selects nobody.
    void set_shared_choice(SharedChoice c) { shared_choice_ = std::move(c); }
    // Id-level view of the same walk for the C-ABI parity tests: per matched filter in
    // App. A.2 order, its relations sorted by rel_id (App. A.5 canonical form).
    // No-Local and the v5 collector are host glue and are not applied here.
    bool match_flat(uint32_t topic_idx, std::string_view topic_name, std::vector<FlatHit>& out, WalkStats* st = nullptr) const;
    bool has_matches(std::string_view topic) const;                          // router.rs:151-154
    std::vector<std::string> get_routes(std::string_view topic, bool* ok) const;   // router.rs:157-170 (unique())
    size_t topics_tree() const { return topics_.values_size(); }            // router.rs:571-574
    int64_t topics_count() const { return topics_count_; }
    int64_t relations_count() const { return relations_count_; }
    uint32_t filter_id_of(const std::string& f) const;

    // Checker only (oracle.cpp: orc_router_match_digest_fast): the per-topic digest of match_flat's hit list in
    // O(matched filters) instead of O(hits).  prepare_digests() pre-reduces every filter's relation list (sorted by
    // rel_id, App. A.5) to FilterDigest; a topic's digest is then composed from the digests of its matched filters in
    // TopicTree::matches order (the order-dependent term shifts by the number of hits that precede the filter).
    struct FilterDigest { uint64_t n = 0, s1 = 0, p = 0, s2 = 0; };   // count, sum v, sum (j+1) v_j, sum v^2;  v = rel_id * 4 + qos
    void prepare_digests(int threads);
    bool match_digest_fast(std::string_view topic_name, uint64_t out[4]) const;

    // Checker for the delivery stage at sizes where the text dump of forwards() is too big: per publish the digest of
    // the per-hit delivery verdicts in canonical hit order (filters in TopicTree::matches order, rel_id ascending inside a filter).
    // WHICH hits are delivered comes from matches() itself — the restated _matches + collector (router.rs:174-265, types.rs:510-540):
    // a hit is delivered iff its rel_id is a row of the returned SubRelationsMap (a v5 row carries the rel_id of the client's FIRST
    // hit, types.rs:535-538).  Per hit x = rel_id * 32 + w, w = min(publish qos, subscription qos) (shared.rs:902) | 4 if v5 &&
    // retain_as_published && publish.retain (shared.rs:886-897) | 8 if dropped by No Local (router.rs:196-201) | 16 if a later v5 hit
    // of a client already collected (types.rs:526-534).  out = { hits, sum x, sum (k+1) x, sum x^2 } mod 2^64.
    bool deliver_digest(const Id& this_id, std::string_view topic_name, uint8_t pub_qos, bool pub_retain, uint64_t out[4]) const;

    // cpu_baseline only (oracle.cpp: orc_router_matches_timed): the reference's per-publish work without the
    // checker's canonicalisation; prepare_shaped() snapshots the relation maps with ref-counted strings.
    void prepare_shaped();
    struct ShapedScratch {     // per-thread result storage of the timed pass (capacity kept between calls: see oracle.cpp)
        struct OutRc { std::shared_ptr<const std::string> filter, client; SubscriptionOptions opts; };
        struct OutPlain { const std::string* filter; const std::string* client; SubscriptionOptions opts; };
        std::unordered_map<NodeId, std::vector<OutRc>> rc;
        std::unordered_map<NodeId, std::vector<OutPlain>> plain;
        struct OutFwd { std::shared_ptr<const std::string> filter, client; SubscriptionOptions opts; std::vector<uint32_t> sub_ids; uint8_t qos; bool retain; };
        struct NodeFwd { std::vector<OutFwd> rows; std::unordered_map<std::string_view, size_t> v5_index; };
        std::unordered_map<NodeId, NodeFwd> fwd;
    };
    // cpu_baseline of the delivery stage: _matches with the v3 / v5 collector (no canonicalising sort) + forwards_to's per-recipient
    // transform (shared.rs:886-908); returns the rows delivered.
    uint64_t forwards_shaped(const Id& this_id, std::string_view topic_name, uint8_t pub_qos, bool pub_retain, WalkStats* st, ShapedScratch* scratch) const;
    uint64_t matches_shaped(const Id& this_id, std::string_view topic_name, WalkStats* st, bool refcounted = true, ShapedScratch* scratch = nullptr) const;
    uint64_t matches_shaped_plain(const Id& this_id, std::string_view topic_name, WalkStats* st, ShapedScratch* scratch = nullptr) const;

   private:
    struct Rel { Id id; SubscriptionOptions opts; uint32_t rel_id; };
    struct FilterEntry { uint32_t filter_id; std::unordered_map<std::string, Rel> rels; FilterDigest dig; };
    struct ShapedEntry { std::shared_ptr<const std::string> client; const Rel* rel; };
    struct ShapedFilter { std::vector<ShapedEntry> rels; };
    SharedChoice shared_choice_;
    std::unordered_map<std::string, ShapedFilter> shaped_;
    int64_t shaped_ready_ = -1;
    TopicTree<Unit> topics_;
    std::unordered_map<std::string, FilterEntry> relations_;   // AllRelationsMap, types.rs:476
    int64_t topics_count_ = 0, relations_count_ = 0;
    uint32_t next_filter_id_ = 0;
    uint64_t mutations_ = 0, digests_at_ = ~0ull;   // prepare_digests() is valid for the mutation count it ran at
};

// ---------------------------------------------------------------- retain.rs
template <class V> struct RetainTree {   // = Node<V>, rmqtt/src/retain.rs:355-358
    std::optional<V> value;
    std::unordered_map<Level, std::unique_ptr<RetainTree>, LevelHash> branches;

    void insert(const Topic& topic, const V& v) {             // retain.rs:373-386
        RetainTree* n = this;
        for (const Level& l : topic) {
            auto& slot = n->branches[l];
            if (!slot) slot = std::make_unique<RetainTree>();
            n = slot.get();
        }
        n->value = v;
    }
    std::optional<V> remove(const Topic& topic) { return remove_at(topic, 0); }   // retain.rs:393-413
    // retain.rs:420-447 — drop values failing f, prune empty nodes, honour max_limit.
    template <class F> size_t retain(size_t max_limit, F f) {
        size_t removed = 0;
        retain_rec(f, removed, max_limit);
        return removed;
    }
    // retain.rs:450-526.
    std::vector<std::pair<Topic, V>> matches(const Topic& filter, WalkStats* st = nullptr) const {
        std::vector<std::pair<Topic, V>> out;
        Topic sub_path;
        walk(filter, 0, sub_path, out, st);
        return out;
    }
    size_t values_size() const {
        size_t n = value ? 1 : 0;
        for (auto& kv : branches) n += kv.second->values_size();
        return n;
    }
    size_t nodes_size() const {
        size_t n = branches.size();
        for (auto& kv : branches) n += kv.second->nodes_size();
        return n;
    }

   private:
    std::optional<V> remove_at(const Topic& t, size_t i) {
        if (i == t.size()) { auto v = value; value.reset(); return v; }
        auto it = branches.find(t[i]);
        if (it == branches.end()) return std::nullopt;
        auto res = it->second->remove_at(t, i + 1);
        if (!it->second->value && it->second->branches.empty()) branches.erase(it);
        return res;
    }
    template <class F> void retain_rec(F& f, size_t& removed, size_t max_limit) {
        if (removed >= max_limit) return;
        for (auto it = branches.begin(); it != branches.end();) {
            RetainTree& c = *it->second;
            c.retain_rec(f, removed, max_limit);
            if (c.value && !f(*c.value)) { c.value.reset(); ++removed; }
            if (!c.value && c.branches.empty()) it = branches.erase(it); else ++it;
        }
    }
    static bool is_multi(const Level& l) { return l.kind == Kind::MultiWildcard; }
    void walk(const Topic& path, size_t i, Topic& sub_path, std::vector<std::pair<Topic, V>>& out, WalkStats* st) const {
        if (st) st->visited++;
        const size_t rem = path.size() - i;
        if (branches.empty() || rem == 0) {                   // retain.rs:464-470
            if (rem == 0 && value) { out.emplace_back(sub_path, *value); if (st) st->hits++; }
            return;
        }
        auto e = branches.find(path[i]);
        if (e != branches.end()) {                            // retain.rs:472-482
            sub_path.push_back(path[i]);
            if (rem > 1 && is_multi(path[i + 1]) && e->second->value) {
                out.emplace_back(sub_path, *e->second->value); if (st) st->hits++;
            }
            e->second->walk(path, i + 1, sub_path, out, st);
            sub_path.pop_back();
        } else if (path[i].kind == Kind::SingleWildcard) {     // retain.rs:483-501
            for (auto& kv : branches) {
                if (sub_path.empty() && kv.first.kind != Kind::Blank && kv.first.is_metadata()) continue;
                sub_path.push_back(kv.first);
                if (rem > 1 && is_multi(path[i + 1]) && kv.second->value) {
                    out.emplace_back(sub_path, *kv.second->value); if (st) st->hits++;
                }
                kv.second->walk(path, i + 1, sub_path, out, st);
                sub_path.pop_back();
            }
        } else if (is_multi(path[i])) {                        // retain.rs:502-524
            for (auto& kv : branches) {
                if (sub_path.empty() && kv.first.kind != Kind::Blank && kv.first.is_metadata()) continue;
                sub_path.push_back(kv.first);
                if (kv.second->branches.empty()) {
                    if (kv.second->value) { out.emplace_back(sub_path, *kv.second->value); if (st) st->hits++; }
                } else {
                    if (kv.second->value) { out.emplace_back(sub_path, *kv.second->value); if (st) st->hits++; }
                    kv.second->walk(path, i, sub_path, out, st);
                }
                sub_path.pop_back();
            }
        }
    }
};

}  // namespace orc
