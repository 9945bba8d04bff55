// Host-side subscription table compiler.
//
// Owns the mutable image of what the reference keeps in `TopicTree<()>` +
// `AllRelationsMap` (rmqtt/src/router.rs:121-127, rmqtt/src/trie.rs:84-87), already in
// the layout the kernels read: a token dictionary, an open-addressed edge table of
// 32-byte records (kernels.hpp) and per-filter subscriber runs.  Mutations follow
// trie.rs:113-149 (insert; remove + prune nodes left with no value and no branches).
// rgr_commit() snapshots this image into HBM as an immutable epoch.
#pragma once
#include <cstdint>
#include <thread>
#include <cstdlib>
#include <algorithm>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include "kernels.hpp"

namespace rgr {

// Level-string -> token id dictionary (open addressing over an append-only byte arena).
class StringDict {
   public:
    StringDict();
    uint32_t find(std::string_view s) const;      // kTokUnknown if absent
    uint32_t intern(std::string_view s);          // existing or new token id
    uint32_t size() const { return uint32_t(entries_.size()); }
    static uint64_t hash(std::string_view s);
    std::string_view str(uint32_t tok) const;
    // raw image for the device tokeniser (kernels.hpp DictView)
    const std::vector<char>& arena() const { return arena_; }
    const std::vector<DictEntry>& entries() const { return entries_; }
    const std::vector<uint32_t>& slots() const { return slots_; }

    // snapshot (table.cpp: SnapWriter / SnapReader)
    template <class W> void save(W& w) const { w.vec(arena_); w.vec(entries_); w.vec(slots_); }
    template <class R> bool load(R& r) {
        if (!r.vec(arena_) || !r.vec(entries_) || !r.vec(slots_)) return false;
        if (slots_.size() < 2 || (slots_.size() & (slots_.size() - 1))) return false;
        for (uint32_t v : slots_) if (v > entries_.size()) return false;
        for (const Entry& e : entries_) if (e.off > arena_.size() || e.len > arena_.size() - e.off) return false;
        mask_ = slots_.size() - 1;
        return true;
    }

   private:
    using Entry = DictEntry;
    std::vector<char> arena_;
    std::vector<Entry> entries_;       // index = token - kTokFirst
    std::vector<uint32_t> slots_;      // token+1, 0 = empty
    uint64_t mask_;
    void grow();
};

// A flat array that is (re)filled in parallel.  The open-addressed edge tables are mostly empty
// slots (8 GiB at 10 M subscriptions) and a single-threaded first touch of those pages dominated
// the bulk build, the snapshot load and the retained-topic compile.
void* flat_alloc(size_t bytes);                       // 2 MiB-aligned + MADV_HUGEPAGE for big blocks; throws std::bad_alloc
unsigned flat_fill_threads(uint64_t n);
template <class T>
class FlatArray {
   public:
    FlatArray() = default;
    FlatArray(const FlatArray&) = delete;
    FlatArray& operator=(const FlatArray&) = delete;
    FlatArray(FlatArray&& o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_) { o.p_ = nullptr; o.n_ = o.cap_ = 0; }
    FlatArray& operator=(FlatArray&& o) noexcept {
        if (this != &o) { release(); p_ = o.p_; n_ = o.n_; cap_ = o.cap_; o.p_ = nullptr; o.n_ = o.cap_ = 0; }
        return *this;
    }
    ~FlatArray() { release(); }
    // discard contents, n copies of v.  The allocation is kept when it is big enough (and not more
    // than 4x too big): refilling mapped pages costs a fraction of first-touching fresh ones.
    void assign(uint64_t n, const T& v) {
        if (n > cap_ || n * 4 < cap_) {
            release();
            if (!n) return;
            p_ = static_cast<T*>(flat_alloc(size_t(n) * sizeof(T)));
            cap_ = n;
        }
        n_ = n;
        if (!n) return;
        const unsigned nt = flat_fill_threads(n);
        if (nt <= 1) { std::fill(p_, p_ + n, v); return; }
        std::vector<std::thread> th;
        T* p = p_;
        for (unsigned k = 0; k < nt; ++k) th.emplace_back([=] { std::fill(p + n * k / nt, p + n * (k + 1) / nt, v); });
        for (auto& t : th) t.join();
    }
    void swap(FlatArray& o) noexcept { std::swap(p_, o.p_); std::swap(n_, o.n_); std::swap(cap_, o.cap_); }
    uint64_t size() const { return n_; }
    const T* data() const { return p_; }
    T* data() { return p_; }
    const T& operator[](uint64_t i) const { return p_[i]; }
    T& operator[](uint64_t i) { return p_[i]; }
    const T* begin() const { return p_; }
    const T* end() const { return p_ + n_; }
    T* begin() { return p_; }
    T* end() { return p_ + n_; }

   private:
    T* p_ = nullptr;
    uint64_t n_ = 0, cap_ = 0;
    void release() { std::free(p_); p_ = nullptr; n_ = cap_ = 0; }
};
using EdgeArray = FlatArray<EdgeEntry>;

class HostTable {
   public:
    HostTable();
    // Parse a filter into tokens (interning new level strings).  false => invalid filter.
    bool tokenize_filter(std::string_view f, std::vector<uint32_t>& toks);
    // Tokenise a publish topic against the dictionary without mutating it.  Returns the
    // topic flags (kTopicInvalid / kTopicMeta); appends tokens only for valid topics.
    uint8_t tokenize_topic(std::string_view t, std::vector<uint32_t>& toks) const;

    int32_t filter_add(std::string_view f, uint32_t* fid);
    int32_t filter_find(std::string_view f, uint32_t* fid) const;
    int32_t filter_remove(uint32_t fid);
    int32_t sub_add(uint32_t fid, uint32_t sub_id, uint8_t qos, uint8_t flags, uint16_t node_idx = 0);
    // Delivery-stage attributes of a subscription (kernels.hpp SubAttr), indexed by sub_id.  They
    // ride in a run parallel to the subscriber run, so setting them marks nothing dirty by itself:
    // call it right before/after the sub_add that (re)writes the run, or use attrs_all_dirty().
    void sub_set_attr(uint32_t sub_id, uint32_t owner_id, uint32_t client_idx);
    bool has_attrs() const { return has_attrs_; }
    SubAttr sub_attr(uint32_t sub_id) const { return sub_id < attrs_.size() ? attrs_[sub_id] : SubAttr{kNone, kNone}; }
    // set by the bulk attribute call: every run must be rebuilt at the next commit
    bool take_attrs_all_dirty() { const bool d = attrs_all_dirty_; attrs_all_dirty_ = false; return d; }
    void mark_attrs_all_dirty() { attrs_all_dirty_ = true; }
    uint64_t n_v5_subs() const { return n_v5_; }
    uint32_t max_sub_id() const { return max_sub_id_; }   // upper bound of the sub ids ever added
    uint32_t max_node_idx() const { return max_node_idx_; }   // upper bound of the node indices ever added (rgr_sub_add_ex)
    int32_t sub_remove(uint32_t fid, uint32_t sub_id);
    // Restore / bulk path (rmqtt-cluster-raft/src/router.rs:557-566 re-inserts every filter after a
    // snapshot): filter_add + sub_add for n subscriptions.  Tokenises on `threads` threads, sorts the
    // filters by token sequence and inserts them in that order so that each filter only creates the
    // trie levels it does not share with its predecessor.  Result is identical to n single calls
    // (filter ids are assigned in sorted order instead of arrival order).
    void subscribe_bulk(const uint8_t* blob, const uint64_t* offs, uint64_t n, const uint32_t* sub_ids, const uint8_t* qos,
                        const uint8_t* flags, uint32_t* fids_out, uint64_t* n_rejected, unsigned threads);

    // Flattened image for the device.
    const EdgeArray& edges() const { return edges_; }
    NodeHeader root_header() const { return root_hdr_; }
    void flatten_filters(std::vector<FilterDesc>& filt, std::vector<SubEntry>& subs) const;
    // Incremental commits: what changed since the last take_delta().
    struct Delta {
        std::vector<uint32_t> slots;     // edge records written (may repeat)
        std::vector<uint32_t> fids;      // filters whose subscriber run / existence changed (may repeat)
        bool relocated = false;          // the edge table was rehashed: every slot moved
    };
    void take_delta(Delta& d);
    uint64_t filter_capacity() const { return filters_.size(); }
    const std::vector<SubEntry>* filter_subs(uint32_t fid) const {
        return fid < filters_.size() && filters_[fid].node != kNone ? &filters_[fid].subs : nullptr;
    }

    // Snapshot file (the format adjacent to the table).  The whole compiled host
    // image — dictionary, trie nodes, edge table, filters, subscriber runs, delivery attributes — as
    // flat arrays, so a cold start is a read + one full device upload instead of re-inserting every
    // subscription (rmqtt-cluster-raft/src/router.rs:557-566 re-inserts one by one).  load() replaces
    // the table; false + *err on I/O errors, a foreign / truncated / corrupt file (checksummed).
    bool save(const std::string& path, std::string* err) const;
    bool load(const std::string& path, std::string* err);
    // distinguishes dictionaries of equal size across a load(): batches tokenised against an older
    // dictionary are re-tokenised
    uint64_t dict_stamp() const { return uint64_t(dict_.size()) | (dict_gen_ << 40); }

    uint64_t n_filters() const { return n_filters_; }
    uint64_t n_subs() const { return n_subs_; }
    uint64_t n_nodes() const { return n_nodes_; }
    uint64_t n_tokens() const { return dict_.size(); }
    uint64_t max_filter_subs() const;
    const StringDict& dict() const { return dict_; }
    void reserve(uint64_t n_filters_hint, uint64_t levels_hint);

   private:
    struct Node {
        uint32_t parent, token;
        uint32_t slot;         // slot of this node's edge record (kNone for the root)
        uint32_t nchild;
        uint32_t term_fid;
        uint32_t plus_child, hash_child;   // node ids
    };
    struct Filter {
        uint32_t node = kNone;             // terminal node; kNone = free id
        std::vector<SubEntry> subs;        // ascending sub_id
    };
    StringDict dict_;
    std::vector<Node> nodes_;
    std::vector<uint32_t> free_nodes_;
    EdgeArray edges_;
    uint64_t edge_used_ = 0;               // live + tombstones
    uint64_t edge_live_ = 0;
    NodeHeader root_hdr_{kNone, kNone, kNone, 0, 0};
    std::vector<Filter> filters_;
    std::vector<uint32_t> free_fids_;
    uint64_t n_filters_ = 0, n_subs_ = 0, n_nodes_ = 1, n_v5_ = 0;
    uint32_t max_sub_id_ = 0;
    uint32_t max_node_idx_ = 0;
    std::vector<SubAttr> attrs_;
    bool has_attrs_ = false, attrs_all_dirty_ = false;
    uint64_t dict_gen_ = 0;
    Delta delta_;
    void touch(uint32_t slot) { delta_.slots.push_back(slot); }

    uint32_t find_slot(uint32_t parent, uint32_t token) const;
    uint32_t insert_edge(uint32_t parent, uint32_t token, uint32_t child);
    void rehash(uint64_t new_cap);
    void rebuild_child_bitmaps();
    uint32_t new_node(uint32_t parent, uint32_t token);
    void set_plus_slot(uint32_t node, uint32_t slot);
    void set_hash_fid(uint32_t node, uint32_t fid);
    void set_term_fid(uint32_t node, uint32_t fid);
    void literal_edge(uint32_t node, uint32_t token, int delta);
    void materialize_edges(const std::vector<uint64_t>& lit_bits);
    uint32_t walk_existing(const std::vector<uint32_t>& toks) const;
};

}  // namespace rgr
