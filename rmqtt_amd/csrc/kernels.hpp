// Device-visible table layout + kernel launch prototypes (HIP, gfx950 only).
//
// HBM layout of one epoch of the subscription table:
//   edges[cap]   open-addressed CSR edge table, 32-byte records keyed by (parent node,
//                level token); a record carries the child's id *and* the child's header
//                (its '+' edge slot, the filter id of "child/#", the filter id ending at
//                child, a 64-bit bitmap over its literal out-edges = the probe miss filter), so one
//                32 B read per visited trie node serves both the edge lookup and the header.
//   filt[nf]     per filter: [begin,count) of its subscriber run in subs[]
//   subs[ns]     packed (sub_id, qos|flags<<8), grouped per filter, ascending sub_id
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define RGR_HD __host__ __device__
#else
#define RGR_HD
#endif

namespace rgr {

struct U4 { uint32_t x, y, z, w; };   // one 16-byte half of an edge record

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kTokPlus = 0;          // reserved token ids: "+" and "#"
constexpr uint32_t kTokHash = 1;
constexpr uint32_t kTokFirst = 2;
constexpr uint32_t kTokUnknown = 0xFFFFFFFFu;   // level string absent from the dictionary
constexpr uint32_t kEdgeEmpty = 0xFFFFFFFFu;
constexpr uint32_t kEdgeTomb = 0xFFFFFFFEu;
constexpr uint32_t kBigPairs = 64;           // per-topic pair lists longer than this get a whole block

struct alignas(16) EdgeEntry {
    uint32_t parent;      // parent node id | kEdgeEmpty | kEdgeTomb
    uint32_t token;       // level token on this edge
    uint32_t child;       // child node id
    uint32_t plus_slot;   // slot of the child's '+' edge record, kNone if none
    uint32_t hash_fid;    // filter id of "<child>/#", kNone if none
    uint32_t term_fid;    // filter id ending at <child>, kNone if none
    uint32_t lit_lo;      // 64-bit bitmap over the literal (non-wildcard) edges leaving <child>: bit lit_bit(token) is set for
    uint32_t lit_hi;      // every such edge.  A clear bit proves the edge absent (no probe); a set bit means "maybe"
};
// Which bit of the child-token bitmap a level token maps to.  (r3: replaces {lit_cnt, lit_xor}, which was exact only for nodes
// with at most one literal child: 32 % of the walk's record reads were probes for children that do not exist.)
RGR_HD inline uint32_t lit_bit(uint32_t token) { return (token * 0x9E3779B1u) >> 26; }
static_assert(sizeof(EdgeEntry) == 32, "edge record is one 32-byte sector");

struct NodeHeader { uint32_t plus_slot, hash_fid, term_fid, lit_lo, lit_hi; };

struct FilterDesc { uint32_t begin, count; };
struct SubEntry { uint32_t sub_id, qos_flags; };
struct Tuple { uint32_t topic_idx, sub_id, qos_flags; };   // == rgr_tuple
// Delivery stage: per-subscription attributes parallel to subs[] (same index),
// per-publish attributes parallel to the batch's topics.
struct SubAttr { uint32_t owner_id, client_idx; };
struct PublishAttr { uint32_t from_id, qos_retain; };       // == rgr_publish_attr
// v5 hit that may be a per-client duplicate: window-relative position + client.  (r3: 16 bytes with the topic and the delivery word; the
// topic is implied by the position — a topic's hits are consecutive — and duplicates are rare, so flagging one is a read-modify-write
// of its tuple word instead of 8 more bytes written and read per candidate: r4.)
struct alignas(8) Cand { uint32_t pos, client_idx; };
// Entries between two tiles' slices of a window's candidate list (expand_tuple.inc writes them, dedup.inc reads them): one tile's worth, so a
// tile whose every hit is a candidate still fits.  A smaller stride is a DIAGNOSTIC (r5u: is the sparse 16 KB-strided layout what a delivery
// window's size is sensitive to? — no) and silently corrupts the lists of tiles with more candidates than the stride.
#ifdef RGR_DIAG_CAND_STRIDE
#ifndef RGR_DIAG_BUILD
#error "RGR_DIAG_CAND_STRIDE gives wrong results by construction: add -DRGR_DIAG_BUILD to say the build is a timing diagnostic"
#endif
#else
#define RGR_DIAG_CAND_STRIDE kTile
#endif
// The other timing diagnostics (expand_tuple.inc RGR_DIAG_LEAN_*: the lean delivery expansion without its stores / entry reads / v5 path / dependent
// gathers, with the tile record or the pair list synthesised; dedup.inc RGR_DIAG_NO_DEDUP_STAT) also give wrong results by construction.
#if (defined(RGR_DIAG_LEAN_NO_STORE) || defined(RGR_DIAG_LEAN_NO_LOAD) || defined(RGR_DIAG_LEAN_NO_V5) || defined(RGR_DIAG_LEAN_NO_ATTR) || \
     defined(RGR_DIAG_LEAN_WARM) || defined(RGR_DIAG_LEAN_FAKE_REC) || defined(RGR_DIAG_LEAN_FAKE_PAIRS)) && !defined(RGR_DIAG_BUILD)
#error "RGR_DIAG_LEAN_* builds give wrong results by construction: add -DRGR_DIAG_BUILD to say the build is a timing diagnostic"
#endif
constexpr uint32_t kSubV5 = 1u << 0, kSubNoLocal = 1u << 1, kSubShared = 1u << 2, kSubRap = 1u << 3;   // RGR_SUB_*
constexpr uint32_t kHitRetain = 1u << 2, kHitNoLocal = 1u << 3, kHitV5Dup = 1u << 4;                    // RGR_HIT_*
RGR_HD inline uint32_t mix32(uint32_t x) {           // bijective
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
struct DeliverArgs {
    const PublishAttr* pub;      // [n_batch]
    const SubAttr* attrs;        // parallel to TrieView::subs, may be null (no ids registered)
    Cand* cand;                  // dedup candidates of this window: tile i owns cand[i*tile_hits ..], null = none wanted
    uint32_t* tile_ncand;        // [tiles] candidates each tile wrote
    uint32_t* tile_trange;       // [2 * tiles] window-relative first / last topic with hits in the tile (written with the tile's flag: bounds the
                                 // tile-local dedup's search for a candidate's topic)
    uint32_t topic_lo;           // first topic of the window (batch-global index)
};

// Level-string dictionary image (host: table.cpp StringDict; device: one copy per epoch).
struct DictEntry { uint64_t hash; uint64_t off; uint32_t len; uint32_t pad; };
struct DictView {
    const uint32_t* slots;      // open addressing: token - kTokFirst + 1, 0 = empty
    uint64_t mask;
    const DictEntry* entries;   // index = token - kTokFirst
    const char* arena;          // level strings, back to back
};

// PUBLISH-packet scan (the step right before Topic::from_str): what the codec extracts from one
// framed PUBLISH packet (rmqtt-codec/src/v3/codec.rs:63-97, v3/decode.rs:110-128, v5/packet/publish.rs:31-52).
struct PubInfo {                 // == rgr_publish_info
    uint64_t topic_off;          // offset of the topic-name bytes inside the packets blob
    uint32_t topic_len;
    uint32_t payload_off;        // offset of the payload inside the packet
    uint16_t packet_id;          // 0 = none (qos 0)
    uint8_t qos, retain, dup;
    uint8_t error;               // 0 ok, else kPubErr*
    uint8_t pad[2];
};
static_assert(sizeof(PubInfo) == 24, "PubInfo layout");
constexpr uint8_t kPubErrNotPublish = 1, kPubErrLength = 2, kPubErrMalformed = 3, kPubErrUtf8 = 4;

// topic flag bits produced by the tokeniser
constexpr uint8_t kTopicInvalid = 1;   // parser rejected it: zero matches
constexpr uint8_t kTopicMeta = 2;      // first level starts with '$' (trie.rs:342-346)

RGR_HD inline uint32_t edge_hash(uint32_t parent, uint32_t token) {
    uint64_t x = (uint64_t(parent) << 32) | token;
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return uint32_t(x);
}

struct TrieView {
    const EdgeEntry* edges;
    uint32_t mask;            // capacity - 1 (capacity is a power of two)
    NodeHeader root;
    const FilterDesc* filt;
    const SubEntry* subs;
    const SubAttr* attrs = nullptr;   // parallel to subs (delivery stage), null when no ids were registered
    // parallel to subs: sub_id | (qos & 3) << 30, 4 bytes per entry (null when some id needs 31+ bits or the array was not built).
    // What RGR_FORMAT_PACKED writes per hit, and RGR_FORMAT_IDS24 after masking: those expansions read 4 bytes per hit instead of 8.
    const uint32_t* subs_packed = nullptr;
};

// ---- RetainTree twin (rmqtt/src/retain.rs): trie of concrete retained topics, nodes numbered
// in DFS preorder so that "every valued node in the subtree of p" is ONE contiguous run of
// vals[]; '+' enumerates a node's children through a CSR child list.
struct alignas(16) REdge { uint32_t parent, token, child, pad; };   // open-addressed (parent,token)->child
// grandchild index: (node g, literal token t) -> the run gc_ids[begin, begin+count) of all nodes x
// with token t whose grandparent is g (under the root: only via non-'$' children).  Lets a '+'
// level followed by a literal level jump two levels with ONE probe instead of enumerating every
// child of g and probing each.
struct alignas(16) GcEdge { uint32_t gparent, token, begin, count; };
struct RetainView {
    const REdge* edges;
    uint32_t mask;
    const GcEdge* gc_edges;
    uint32_t gc_mask;
    const uint32_t* gc_ids;
    const uint32_t* child_off;   // [N+1] CSR into child_ids (children in preorder order)
    const uint32_t* child_ids;
    uint32_t root_nonmeta;       // root's children that are not '$'-metadata come first
    uint32_t n_nodes;
    // run descriptors (FilterDesc) the downstream kernels expand: index 2p = p's own value,
    // 2p+1 = every value in subtree(p) (p included), 2N = every value outside '$' subtrees
    const FilterDesc* desc;
    const SubEntry* vals;        // {topic_id, 0} of valued nodes in preorder
};

struct WalkArgs {
    const uint32_t* tokens;      // CSR token ids of the batch
    const uint64_t* tok_off;     // [n_batch+1]
    const uint8_t* tflags;       // [n_batch]
    uint32_t topic_base;         // first topic of this chunk
    uint32_t n;                  // topics in this chunk
    uint32_t slot_cap;           // C
    uint32_t* slots;             // [C][n]  matched filter ids, j-major
    uint32_t* pair_cnt;          // [n] matched filters per topic (true count, may exceed C)
    uint32_t* path_scratch;      // [total tokens] spill of the DFS stack beyond the LDS window
    unsigned long long* visited; // optional: sum of visited trie nodes
    // overflow (count > C): main pass registers the topic, overflow pass re-walks it into the arena
    uint32_t* ovf_list;          // [n] chunk-local topic indices
    uint32_t* ovf_count;         // device scalar
    uint64_t* ovf_base;          // [n] arena base per overflow topic
    unsigned long long* ovf_cursor;
    uint32_t* ovf_arena;
    uint64_t ovf_arena_cap;
    // (r7m) main pass of a SMALL chunk: only every 2^lane_shift-th lane of a block walks a topic (set by launch_walk, see there)
    uint32_t lane_shift = 0;
};

struct ChunkArrays {
    uint32_t n;                  // topics in chunk
    uint32_t slot_cap;
    const uint32_t* slots;
    const uint32_t* pair_cnt;
    uint32_t* hit_cnt;           // [n]
    uint32_t* pair_live;         // [n] matched filters with at least one subscriber
    uint64_t* hit_off;           // [n+1] exclusive scan of hit_cnt (chunk-local)
    uint64_t* pair_base;         // [n+1] exclusive scan of pair_live
    const uint64_t* ovf_base;
    const uint32_t* ovf_arena;
    uint64_t ovf_arena_cap;
    uint32_t* error_flag;
    // topics with more than kBigPairs matched filters are counted / compacted by a whole block
    uint32_t* big_list;          // [n]
    uint32_t* big_count;         // device scalar (zeroed before the count kernel)
    // dense (topic, subscriber-run) pairs
    uint32_t* pair_src;          // [P] subs[] index of the run
    uint32_t* pair_topic;        // [P] batch-global topic index
    uint64_t* pair_off;          // [P+1] chunk-local output offset of the run
    // delivery stage only (null otherwise): publish qos|retain<<2 of the pair's topic, copied next to
    // the pair at compaction so that the expansion's prologue has no dependent load for it
    const PublishAttr* pub;      // [n_batch]
    uint8_t* pair_qr;            // [P]
    // rgr_batch_set_topic_ids: value written into rgr_tuple.topic_idx for batch topic i (null: i itself)
    const uint32_t* topic_ids = nullptr;
};

// incremental epoch update: patch `n` edge records / filter descriptors of a device image
void launch_scatter_edges(EdgeEntry* dst, const uint32_t* slots, const EdgeEntry* recs, uint32_t n, void* stream);
void launch_scatter_desc(FilterDesc* dst, const uint32_t* fids, const FilterDesc* recs, uint32_t n, void* stream);
// device tokeniser: blob/offsets -> per-topic level counts + flags, then token ids
// force_invalid (optional, [n]): nonzero entries are forced to kTopicInvalid (malformed PUBLISH packets)
void launch_tok_count(const uint8_t* blob, const uint64_t* offs, uint32_t n, uint32_t* level_cnt, uint8_t* tflags, void* stream,
                      const uint8_t* force_invalid = nullptr);
// PUBLISH packets -> PubInfo per packet (+ topic_len[] for the scan, bad[] for the tokeniser, optional publish attrs)
void launch_publish_scan(const uint8_t* pkts, const uint64_t* pkt_offs, uint32_t n, int version, PubInfo* info, uint32_t* topic_len, uint8_t* bad,
                         const uint32_t* from_ids, PublishAttr* attrs, void* stream);
// gather the topic-name fields into a dense blob (topic_offs = exclusive scan of topic_len)
void launch_publish_topics(const uint8_t* pkts, const PubInfo* info, uint32_t n, const uint64_t* topic_offs, uint8_t* blob, void* stream);
void launch_scan_u32(const uint32_t* in, uint64_t* out, uint32_t n, uint64_t* block_tmp, void* stream);   // out[n] = total
void launch_tok_fill(const DictView& d, const uint8_t* blob, const uint64_t* offs, uint32_t n, const uint64_t* tok_off,
                     const uint8_t* tflags, uint32_t* tokens, void* stream);
void launch_walk(const TrieView& t, const WalkArgs& a, bool overflow_pass, void* stream);
// RetainTree twin: level-synchronous frontier expansion
struct RetainRound {
    const uint32_t* tokens; const uint64_t* tok_off; const uint8_t* tflags;
    uint32_t topic_base;        // first filter of the chunk
    const uint32_t* fdepth;     // [n] level each filter's items stand at this round
    uint32_t m;                 // frontier size
    const uint32_t* f_filter;   // [m] chunk-local filter index (null in round 0: item i = filter i at the root)
    const uint32_t* f_node;     // [m]
    uint32_t* cnt; uint32_t* payload;       // [m] contribution to the next frontier
    uint32_t* ecnt; uint32_t* e0; uint32_t* e1;   // [m] emitted descriptors
};
void launch_retain_step(const RetainView& t, const RetainRound& r, void* stream);
// after a round: fdepth[f] += 1, or 2 where the level was a '+' followed by a literal level
void launch_retain_advance(const RetainRound& r, uint32_t n, uint32_t* fdepth, void* stream);
void launch_retain_next(const RetainView& t, const RetainRound& r, const uint64_t* out_off, uint32_t* nf_filter, uint32_t* nf_node,
                        uint32_t* big_list, uint32_t* big_count, void* stream);
void launch_retain_emit(const RetainRound& r, const uint64_t* epos, uint64_t g_base, uint32_t* arena, uint64_t* ovf_base,
                        uint64_t* ovf_end, void* stream);
void launch_retain_finish(uint32_t n, const uint64_t* ovf_base, const uint64_t* ovf_end, uint32_t* pair_cnt, void* stream);

// matched filter ids of every topic of the chunk, densely, in iteration order: out[off[t] + j] = j-th matched
// filter of topic t (off = exclusive scan of pair_cnt) — the result of rgr_match_filters
// reps: the first subscriber's sub id of each matched filter instead of its filter id (kNone: no subscriber)
void launch_pairs_dense(const TrieView& t, const ChunkArrays& c, const uint64_t* off, uint32_t* out, bool reps, void* stream);
void launch_count(const TrieView& t, const ChunkArrays& c, void* stream);
void launch_scan(const ChunkArrays& c, uint64_t* block_tmp, void* stream);
void launch_compact(const TrieView& t, const ChunkArrays& c, uint32_t topic_base, void* stream);
// Per output tile: the first pair that intersects it, plus that pair's view at the tile's first position (where in
// subs[] the tile starts reading, which topic it belongs to) — so a tile that lies inside ONE run, the common case at
// high fan-out, starts its subscriber loads after a single 16-byte read instead of tile_first -> pair arrays -> LDS.
struct alignas(16) TileRec { uint32_t first, src, topic, qr; };
void launch_tiles(const ChunkArrays& c, uint64_t pair_lo, uint64_t pair_hi, uint64_t hit_lo, TileRec* tile_first, void* stream);
// hits8: `out` receives 8-byte hits {sub_id, delivery word} (delivery passes in RGR_FORMAT_DELIVER8; lean expansion only)
// id_source (plain tuples): 0 = the 8-byte entries, 1 = t.subs_packed (entries without flags: the retained path's single-tier epochs),
// 2 = the entry's index in subs[] (retained-path passes that answer with positions, rgr_batch_set_retain_positions)
void launch_expand(const TrieView& t, const ChunkArrays& c, uint64_t pair_lo, uint64_t pair_hi, uint64_t hit_lo, uint64_t hit_hi,
                   const TileRec* tile_first, Tuple* out, void* stream, const DeliverArgs* deliver = nullptr, bool hits8 = false, int id_source = 0);
// compact result formats (rgr_batch_set_format): sub ids (+ a qos byte array) without the topic column
constexpr int kFmtTuple = 0, kFmtSoa = 1, kFmtPacked = 2, kFmtRuns = 3, kFmtIds24 = 4, kFmtDeliver8 = 5;     // == RGR_FORMAT_*
// packed[i] = subs[i].sub_id | (subs[i].qos_flags & 3) << 30 for i in [0, n)
constexpr uint32_t kPackedPad = 16;      // entries allocated past the end of a packed side array (never read for their value)
void launch_pack_subs(const SubEntry* subs, uint64_t n, uint32_t* packed, void* stream);
// RGR_TILES_FUSED: the pair range of the NEXT window of the chunk and where its tile records go (written by extra blocks of the expansion)
struct NextTiles { uint64_t pair_lo, pair_hi, hit_lo; TileRec* out; };
// -> true when `next` was given AND its records were written by this launch (only the lane-held IDS24 expansion does that)
bool launch_expand_compact(const TrieView& t, const ChunkArrays& c, uint64_t pair_lo, uint64_t pair_hi, uint64_t hit_lo, uint64_t hit_hi,
                           const TileRec* tile_first, int format, uint32_t* out_ids, uint8_t* out_qos, void* stream, const NextTiles* next = nullptr);
// v5 per-client dedup over a window's candidates (match_core.hpp: LDS tile tables + LDS topic tables): first position per
// (topic, client) wins, every other candidate gets kHitV5Dup.  hit_off points at the window's first topic (chunk-local
// offsets, hit_lo = the window's first position); nt = topics of the window; items must hold
// nt + n_hits / dedup_topic_cap() + 1 entries; stat[dedup_stat_slots()] accumulates the candidate count (block i of the tile pass adds to stat[i]).  Everything is stream-ordered: no host sync.
// work item of the topic pass: part `part` of `parts` of window topic `topic` (nc candidates in total)
// (r6: the item carries the topic's window-relative hit range itself — the topic pass used to fetch it through the topic index, one more dependent
// round trip per item of a pass that is bound by exactly those)
struct DedupItem { uint32_t h0, h1, part, parts_slots; };      // hits [h0, h1) of the window; parts_slots = parts | log2(table slots of a part) << 24
// where the delivery word of window position p lives: the third word of a 12-byte tuple, or the second of an 8-byte hit (RGR_FORMAT_DELIVER8)
struct HitWords {
    uint32_t* first; uint32_t stride;        // in 32-bit words
    RGR_HD uint32_t& at(uint32_t pos) const { return first[uint64_t(pos) * stride]; }
};
inline HitWords tuple_words(Tuple* t) { return HitWords{t ? &t->qos_flags : nullptr, 3}; }
inline HitWords hit8_words(void* p) { return HitWords{p ? static_cast<uint32_t*>(p) + 1 : nullptr, 2}; }
void launch_dedup(const Cand* cand, const uint32_t* tile_ncand, const uint32_t* tile_trange, uint32_t ntiles, HitWords words, uint32_t nt,
                  const uint64_t* hit_off, uint64_t hit_lo, DedupItem* items, uint32_t* item_counts /* two words, zero when the pass begins */,
                  uint32_t parity /* dedup launches so far in the pass & 1 */, unsigned long long* stat, void* stream);
uint32_t dedup_topic_cap();
uint32_t dedup_stat_slots();         // u64 slots of launch_dedup's `stat` array (one per block of the tile pass)
// Delivery results grouped by node (SubRelationsMap's shape, types.rs:486-497): stable partition of every topic's tuples by
// one byte (`shift` = 16 or 24) of the delivery word's node index; then the directory of the node groups — called twice: with
// group_node == null it writes group_cnt[t], with the scanned offsets it writes (node, first position + begin_bias) per group.
void launch_node_partition(const Tuple* in, Tuple* out, const uint64_t* hit_off, uint64_t hit_lo, uint32_t nt, uint32_t shift, void* stream);
void launch_node_groups(const Tuple* tuples, const uint64_t* hit_off, uint64_t hit_lo, uint32_t nt, uint32_t* group_cnt, const uint64_t* group_off,
                        uint32_t* group_node, uint64_t* group_begin, uint64_t begin_bias, void* stream);
// run descriptors of a window packed for the exchange step (== rgr_run)
struct RunDesc { uint32_t shard, src, len, topic; };
void launch_pack_runs(const uint32_t* src, const uint32_t* topic, const uint64_t* off, uint64_t n, uint32_t shard, RunDesc* out, void* stream);
// walk order of a publish batch (order.hip): sort by the first two level tokens, gather the token arrays into that order
size_t order_sort_temp_bytes(uint32_t n);
int launch_order_sort(const uint32_t* tokens, const uint64_t* tok_off, const uint8_t* tflags, uint32_t n, unsigned long long* keys, unsigned long long* keys_tmp,
                      uint32_t* idx_tmp, uint32_t* perm, void* temp, size_t temp_bytes, void* stream);
void launch_order_len(const uint32_t* perm, const uint64_t* tok_off, uint32_t n, uint32_t* len, void* stream);
void launch_order_gather(const uint32_t* perm, uint32_t n, const uint64_t* off_old, const uint32_t* tok_old, const uint8_t* fl_old, const uint64_t* off_new, uint32_t* tok_new,
                         uint8_t* fl_new, void* stream);
void launch_order_compose(const uint32_t* perm, const uint32_t* ids, uint32_t n, uint32_t* out, void* stream);
void launch_order_gather_attrs(const uint32_t* perm, const PublishAttr* in, uint32_t n, PublishAttr* out, void* stream);
uint32_t expand_tile_hits();
const char* expand_ids24_kernel_name();      // ... 3-byte-id windows by default (RGR_COMPACT_LP overrides per launch)
const char* expand_tuple_kernel_name();      // which kernel expands plain 12-byte tuple windows (profilers see this name)
uint32_t scan_block_topics();

}  // namespace rgr
