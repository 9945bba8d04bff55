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
    // Not re-entrant (scratch buffers are reused): callers serialise compiles.
    void compile(RetainImage& out) const;
    const StringDict& dict() const { return dict_; }
    uint64_t n_topics() const { return n_values_; }
    uint64_t n_nodes() const { return n_nodes_; }
    // bumped by every mutation that changes what compile() would emit
    uint64_t version() const { return version_; }

   private:
    struct Node { uint32_t parent, token, slot, nchild, value; bool meta; };
    StringDict dict_;
    std::vector<Node> nodes_;
    std::vector<uint32_t> free_nodes_;
    std::vector<REdge> edges_;     // (parent,token)->child over the mutable node ids
    uint64_t edge_used_ = 0, edge_live_ = 0, n_values_ = 0, n_nodes_ = 1, version_ = 1;
    uint32_t find(uint32_t parent, uint32_t token) const;
    uint32_t insert_edge(uint32_t parent, uint32_t token, uint32_t child);
    void rehash(uint64_t cap);
    bool tokenize(std::string_view s, std::vector<uint32_t>& toks, bool intern, bool* first_meta);
    mutable RetainCompileScratch scratch_;
};

}  // namespace rgr
