// RetainTree twin — host table + snapshot compiler.
//
// Mutable image of what the reference keeps in `RetainTree<V>` (rmqtt/src/retain.rs:355-358):
// a trie of concrete retained topic names, one optional value per node (here: the caller's
// u32 topic_id; the message itself stays with the host / KV store, as in
// rmqtt-plugins/rmqtt-retainer/src/storage.rs:256,604-611).  insert = retain.rs:373-386
// (value replaced), remove = retain.rs:393-413 (prune nodes left without value and branches).
// compile() renumbers the nodes in DFS preorder and emits the device layout of
// kernels.hpp::RetainView.
#pragma once
#include <cstdint>
#include <string_view>
#include <vector>

#include "kernels.hpp"
#include "table.hpp"

namespace rgr {

struct RetainImage {   // host copy of one snapshot, preorder-numbered
    FlatArray<REdge> edges;                // open-addressed, filled in parallel (retain.cpp)
    std::vector<GcEdge> gc_edges;          // grandchild index (kernels.hpp)
    std::vector<uint32_t> gc_ids;
    std::vector<uint32_t> child_off, child_ids;
    std::vector<FilterDesc> desc;
    std::vector<SubEntry> vals;
    uint32_t root_nonmeta = 0, n_nodes = 0;
};

// compile()'s temporaries, kept between calls: on this class of hosts first-touching a few hundred
// MB of fresh pages costs more than the work done in them.
struct RetainCompileScratch {
    struct Tri { uint32_t g, tok, x; };
    std::vector<uint32_t> cnt, hash_kid, order, ppre, ptok, sub_end, own_pos, hp_b, hp_e, val_rank, home, sorted, pcount, goff, pos;
    std::vector<uint64_t> kt;
    std::vector<uint8_t> has;
    std::vector<Tri> tri, tri_sorted;
};

class RetainTable {
   public:
    RetainTable();
    // RGR_OK / RGR_EINVAL_TOPIC.  The topic may be any string Topic::from_str accepts.
    int32_t topic_add(std::string_view topic, uint32_t topic_id);
    int32_t topic_remove(std::string_view topic);     // RGR_OK / RGR_ENOENT / RGR_EINVAL_TOPIC
    // Tokenise a filter against the dictionary (read-only): flags as HostTable::tokenize_topic.
    uint8_t tokenize_filter(std::string_view f, std::vector<uint32_t>& toks) const;
    // Not re-entrant (scratch buffers are reused): callers serialise compiles.  val_of_node
    // (optional, indexed by the table's own node ids): where each valued node landed in vals[].
    void compile(RetainImage& out, std::vector<uint32_t>* val_of_node = nullptr) const;
    // Node a topic name ends at (kNone: absent or not a valid name) and its value (kNone: none).
    uint32_t find_node(std::string_view topic) const;
    uint32_t node_value(uint32_t node) const { return nodes_[node].value; }
    const StringDict& dict() const { return dict_; }
    uint64_t n_topics() const { return n_values_; }
    uint64_t n_nodes() const { return n_nodes_; }
    // bumped by every mutation that changes what compile() would emit
    uint64_t version() const { return version_; }
    uint32_t max_id() const { return max_id_; }        // upper bound of the topic ids ever stored

   private:
    struct Node { uint32_t parent, token, slot, nchild, value; bool meta; };
    StringDict dict_;
    std::vector<Node> nodes_;
    std::vector<uint32_t> free_nodes_;
    std::vector<REdge> edges_;     // (parent,token)->child over the mutable node ids
    uint64_t edge_used_ = 0, edge_live_ = 0, n_values_ = 0, n_nodes_ = 1, version_ = 1;
    uint32_t max_id_ = 0;
    uint32_t find(uint32_t parent, uint32_t token) const;
    uint32_t insert_edge(uint32_t parent, uint32_t token, uint32_t child);
    void rehash(uint64_t cap);
    bool tokenize(std::string_view s, std::vector<uint32_t>& toks, bool intern, bool* first_meta);
    mutable RetainCompileScratch scratch_;
};

// ---- two tiers: the retained set as an immutable compiled BASE plus a small DELTA
// of the topics added since the base was compiled.  A structural change no longer forces a full
// recompile: additions recompile the (small) delta only, removals of base topics set a dead bit in
// the base's vals[] entry (the hit is still produced, flagged, and dropped by the consumer), and
// the two tiers are merged — one full compile — when the delta or the dead fraction grows past a
// threshold.  Base and delta hold disjoint topic sets, so a query's answer is the base hits that
// are not dead followed by the delta hits.  No kernel is involved in any of this.
constexpr uint32_t kRetainDead = 1u;      // SubEntry::qos_flags bit of a base value removed since the last merge

class TieredRetain {
   public:
    int32_t topic_add(std::string_view topic, uint32_t topic_id);   // as RetainTable::topic_add
    int32_t topic_remove(std::string_view topic);                    // as RetainTable::topic_remove
    uint64_t n_topics() const { return all_.n_topics(); }
    uint64_t n_delta() const { return delta_.n_topics(); }
    uint64_t n_dead() const { return n_dead_; }
    // Merge now?  Never compiled yet, delta larger than delta_max topics, > 25 % of the base dead — or
    // the set holds (or just lost) a topic NAME with a literal '+' / '#' level: the reference's
    // exact-first rule (retain.rs:472) makes such a name hide its siblings, which only one tree can
    // express, so tiering is suspended while any exists.  (MQTT forbids wildcards in PUBLISH topic
    // names; a broker never stores one.)
    bool wants_merge(uint64_t delta_max) const {
        return !merged_once_ || force_merge_ || n_wild_ > 0 || delta_.n_topics() > delta_max || n_dead_ * 4 > base_topics_ + 4096;
    }
    void compile_base(RetainImage& out);       // whole set -> base image; the delta tier becomes empty
    bool delta_dirty() const { return delta_.version() != delta_compiled_; }
    void compile_delta(RetainImage& out);      // topics added since the last merge
    struct Dead { uint32_t val_index, topic_id; };
    std::vector<Dead> take_dead() { std::vector<Dead> d; d.swap(pending_dead_); return d; }   // base entries to flag
    // the same list left in place until the caller has applied it (a failed device patch must not lose it)
    const std::vector<Dead>& pending_dead() const { return pending_dead_; }
    void clear_pending_dead() { pending_dead_.clear(); }
    const RetainTable& base_table() const { return all_; }     // tokenises queries of the base tier
    const RetainTable& delta_table() const { return delta_; }

   private:
    RetainTable all_;                          // the logical set (what RetainTree holds in the reference)
    RetainTable delta_;
    std::vector<uint32_t> base_val_of_node_;   // all_ node id -> vals[] index in the base image (kNone: not a live base value)
    std::vector<Dead> pending_dead_;
    uint64_t n_dead_ = 0, base_topics_ = 0, delta_compiled_ = 0, n_wild_ = 0;
    bool merged_once_ = false, force_merge_ = false;
    bool in_delta(std::string_view topic) const;
    void mark_dead(uint32_t node, uint32_t topic_id);
};

// Answer of a tiered query from the two tiers' answers: per filter the base hits without the dead
// ones, then the delta hits (the tiers are disjoint).  `base_flags` = vals[].qos_flags of every
// base hit; delta arrays may be null (no delta tier).
inline void merge_tier_hits(uint32_t n, const uint64_t* base_off, const uint32_t* base_ids, const uint32_t* base_flags, const uint64_t* delta_off,
                            const uint32_t* delta_ids, std::vector<uint64_t>& out_off, std::vector<uint32_t>& out_ids) {
    out_off.assign(size_t(n) + 1, 0);
    out_ids.clear();
    for (uint32_t f = 0; f < n; ++f) {
        for (uint64_t k = base_off[f]; k < base_off[f + 1]; ++k)
            if (!(base_flags[k] & kRetainDead)) out_ids.push_back(base_ids[k]);
        if (delta_off)
            for (uint64_t k = delta_off[f]; k < delta_off[f + 1]; ++k) out_ids.push_back(delta_ids[k]);
        out_off[f + 1] = out_ids.size();
    }
}

}  // namespace rgr
