// C ABI of include/rmqtt_gpu_router.h: handle / epoch / batch orchestration.
//
// Pipeline per chunk of topics (all on the batch's own HIP stream):
//   walk (+ overflow re-walk) -> count -> scan -> compact      [one host sync: sizes]
//   then per window of topics: tiles -> expand                  [async]
// Epochs are immutable device snapshots of the host table; a pass binds the epoch that
// is current at rgr_batch_begin().
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include "device.hpp"
#include "kernels.hpp"
#include "retain.hpp"
#include "rmqtt_gpu_router.h"
#include "table.hpp"

using namespace rgr;

namespace {

thread_local std::string g_last_error;

int32_t fail(int32_t code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

template <class Fn> int32_t guarded(Fn&& fn) {
    try {
        return fn();
    } catch (const HipError& e) {
        return fail(RGR_EDEVICE, e.what());
    } catch (const std::bad_alloc&) {
        return fail(RGR_ENOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(RGR_EINVAL, e.what());
    }
}

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct DictImage {   // device copy of a StringDict for the device tokeniser
    DevBuf slots, entries, arena;
    DictView view{};
    uint64_t n_tokens = 0;
    void upload(const StringDict& sd, uint64_t stamp = ~0ull) {
        slots.ensure(sd.slots().size() * 4);
        entries.ensure(std::max<size_t>(1, sd.entries().size()) * sizeof(DictEntry));
        arena.ensure(std::max<size_t>(1, sd.arena().size()));
        RGR_HIP(hipMemcpy(slots.p, sd.slots().data(), sd.slots().size() * 4, hipMemcpyHostToDevice));
        if (!sd.entries().empty()) RGR_HIP(hipMemcpy(entries.p, sd.entries().data(), sd.entries().size() * sizeof(DictEntry), hipMemcpyHostToDevice));
        if (!sd.arena().empty()) RGR_HIP(hipMemcpy(arena.p, sd.arena().data(), sd.arena().size(), hipMemcpyHostToDevice));
        view.slots = slots.as<uint32_t>();
        view.mask = sd.slots().size() - 1;
        view.entries = entries.as<DictEntry>();
        view.arena = arena.as<char>();
        n_tokens = stamp == ~0ull ? sd.size() : stamp;     // identifies the dictionary a batch was tokenised against
    }
};

// Device images an epoch is made of.  Edge table and filter descriptors are ping-ponged between
// two images patched in place by the scatter kernels; subscriber runs live in an append-only pool
// (a run is never overwritten, so older epochs stay valid).  shared_ptr keeps whatever an
// in-flight pass still reads alive.
struct EdgeImage { DevBuf buf; uint64_t cap = 0; std::vector<uint32_t> pending; bool need_full = true; };
struct FiltImage { DevBuf buf; uint64_t cap = 0; std::vector<uint32_t> pending; bool need_full = true; };
// attr_buf: SubAttr parallel to buf; packed_buf: sub_id | qos << 30 parallel to buf (TrieView::subs_packed), filled on the device
// for every range of entries a commit uploads
struct SubsPool { DevBuf buf, attr_buf, packed_buf; uint64_t used = 0, cap = 0; bool has_attrs = false; };

struct Epoch {
    std::shared_ptr<DictImage> dict;
    std::shared_ptr<EdgeImage> edges;
    std::shared_ptr<FiltImage> filt;
    std::shared_ptr<SubsPool> subs;
    TrieView view{};
    uint64_t id = 0, n_filters = 0, n_subs = 0, n_nodes = 0, edge_slots = 0, bytes = 0, n_v5 = 0;
    uint32_t max_sub_id = 0;          // upper bound of the sub ids ever added (RGR_FORMAT_PACKED needs < 2^30)
    uint32_t max_node_idx = 0;        // upper bound of the node indices in the delivery words (node partition: how many radix passes)
};

struct RetainEpoch {
    DictImage dict;
    DevBuf edges, child_off, child_ids, desc, vals, gc_edges, gc_ids;
    RetainView view{};
    TrieView tv{};       // filt = run descriptors, subs = values: what count/compact/expand read
    uint64_t id = 0, n_topics = 0, n_nodes = 0, bytes = 0, table_version = 0;
    uint32_t max_id = 0;              // upper bound of the topic ids ever added
    // host mirror of vals[] (rgr_retain_match_ranges hands out pointers into it): immutable once published; the two-tier commit
    // replaces it copy-on-write when it marks entries dead.  Read / replaced under rgr_handle::epoch_mu.
    std::shared_ptr<const std::vector<SubEntry>> h_vals;
    DevBuf vals_packed;                // TrieView::subs_packed of the retained values (single-tier epochs only)
    const uint32_t* tv_packed = nullptr;
    uint32_t max_id_hint = 0xFFFFFFFFu;
};

enum SpanKind { kSpanWalk = 0, kSpanScan = 1, kSpanExpand = 2, kSpanDedup = 3 };

}  // namespace

struct rgr_handle {
    rgr_config cfg{};
    bool cfg_window_explicit = false;   // the caller chose window_hits (then every pass uses it as given)
    std::shared_mutex table_mu;     // HostTable: shared for tokenising, exclusive for mutation
    HostTable table;
    std::mutex epoch_mu;
    std::shared_ptr<Epoch> epoch;
    uint64_t epoch_counter = 0;
    // incremental-commit state (guarded by commit_mu)
    std::mutex commit_mu;
    std::shared_ptr<EdgeImage> edge_img[2];
    std::shared_ptr<FiltImage> filt_img[2];
    std::shared_ptr<SubsPool> sub_pool;
    std::vector<FilterDesc> host_desc;     // per filter id: its run in the pool
    uint64_t pool_garbage = 0;
    int cur_img = 1;
    uint64_t commits_full = 0, commits_delta = 0;
    std::mutex stats_mu;
    rgr_stats stats{};
    // recycled batch workspaces (stream + grow-only device buffers) for the one-shot entry points
    std::mutex pool_mu;
    std::vector<rgr_batch*> pool;
    std::shared_ptr<PinnedPool> pinned = std::make_shared<PinnedPool>();   // result blocks of the host-buffer entry points
    // RetainTree twin
    std::shared_mutex retain_mu;
    RetainTable retain_table;
    std::shared_ptr<RetainEpoch> retain_epoch;
    std::mutex retain_commit_mu;       // serialises rgr_retain_commit (the compile scratch and image are reused)
    RetainImage retain_img;
    // two-tier mode (cfg.retain_delta_max > 0): `tiers` replaces retain_table, retain_epoch is the base tier
    TieredRetain tiers;
    std::shared_ptr<RetainEpoch> retain_delta_epoch;   // null while the delta tier is empty
    std::shared_mutex retain_pair_mu;                  // shared: a query reads both tiers; exclusive: commit swaps the (base, delta) pair
    RetainImage retain_delta_img;
    uint64_t retain_merges = 0;
    bool retain_tiered() const { return cfg.retain_delta_max > 0; }
    // table that tokenises the filters of a retain batch of the given tier
    const RetainTable& retain_tokenizer(uint32_t tier) const {
        return !retain_tiered() ? retain_table : tier ? tiers.delta_table() : tiers.base_table();
    }
};

// Everything one chunk of topics needs between its walk and the expansion of its last window.
struct ChunkSlot {
    DevBuf slots, pair_cnt, hit_cnt, pair_live, hit_off, pair_base, ovf_list, ovf_base, scalars, arena;
    DevBuf pair_src, pair_topic, pair_off, pair_qr, r_big, scan_tmp;
    PinnedBuf h_hit_off, h_pair_base, h_scalars;
    uint64_t arena_cap = 0;
    uint32_t begin = 0, n = 0;
    uint64_t total_hits = 0, total_pairs = 0;
    bool host_arrays = false;    // h_hit_off / h_pair_base hold the chunk's per-topic offsets (skipped for a one-window chunk until someone asks)
    bool ready = false;          // walked, counted, scanned, compacted
    bool inflight = false;       // prepared asynchronously on prep_stream: `done` tells when
    hipEvent_t done = nullptr;
    ~ChunkSlot() { if (done) (void)hipEventDestroy(done); }
};

struct rgr_batch {
    rgr_handle* h = nullptr;
    uint32_t n = 0;
    std::vector<int32_t> status;
    uint64_t total_tokens = 0, valid_levels = 0, valid_topics = 0;
    DevBuf d_tokens, d_tok_off, d_tflags, d_path;
    uint64_t blob_bytes = 0;          // bytes of the topic strings in d_blob (device tokeniser: bounds the token count of a micro-batch)
    // rgr_batch_set_order(RGR_ORDER_WALK): the token arrays once more, gathered into walk order (sorted by the leading tokens, order.hip); d_order[k] = batch
    // index of the topic walked k-th.  The tokeniser keeps writing the caller-order arrays; apply_order() follows every (re)tokenisation.
    bool ordered = false;
    DevBuf s_tokens, s_tok_off, s_tflags, d_order, d_order_ids, order_tmp[4];
    std::vector<uint32_t> h_order;
    DevBuf d_blob, d_offs, d_level_cnt;   // raw topics (device tokeniser)
    // rgr_batch_create_from_publish: raw PUBLISH packets, what the scan extracted, which packets the codec rejects
    bool from_publish = false;
    DevBuf d_pkts, d_pkt_offs, d_pubinfo, d_force;
    std::vector<PubInfo> pub_info;
    std::vector<uint8_t> h_blob;          // raw topics kept on the host (host tokeniser only)
    std::vector<uint64_t> h_offs;
    uint64_t dict_tokens = ~0ull;         // stamp of the dictionary the batch was tokenised against
    bool host_tok = false;                // tokenised by the host threads (rgr_config.host_tokenize; never for two-tier retain batches)
    hipStream_t stream = nullptr;
    // chunk work buffers: two slots, so that the next chunk's walk / count / scan / compact can run on `prep_stream`
    // while the current chunk's windows are being expanded
    ChunkSlot cs[2];
    ChunkSlot* c = &cs[0];
    hipStream_t prep_stream = nullptr;
    hipEvent_t ev_windows_done = nullptr;   // recorded on `stream` after the last window launch of a chunk
    DevBuf tile_first, out, scan_tmp;
    // RGR_TILES_FUSED=1 (A/B switch, not measured yet): the tile records of window k + 1 are written by extra blocks at the end of the
    // grid that expands window k (two record buffers, window k uses buffer k & 1), so that consecutive expansions of a chunk follow each
    // other without a tiles_kernel launch — and its gap — in between (~20 us of a 0.65 ms ids24 window).  No second stream, no events:
    // the records are complete when the kernel that wrote them is.
    DevBuf tile_buf[2];
    struct TilePlan { bool valid = false; const void* chunk = nullptr; uint32_t lc = 0, le = 0; uint64_t hit_lo = 0, pair_lo = 0, pair_hi = 0; int slot = 0; } tile_plan;
    uint32_t win_seq = 0;                // windows expanded so far in this pass
    uint32_t dedup_seq = 0;              // dedup launches so far in this pass: its parity selects the item counter (launch_dedup zeroes the other one)
    DevBuf out2;                         // second window buffer (rgr_batch_run_to_host double-buffers)
    PinnedBuf h_ring[2];                 // pinned staging for streamed windows
    bool alt_out = false;                // next_window expands into out2 instead of out
    bool want_host_offsets = false;      // host-out entry points: fetch the per-topic offsets with the chunk's totals
    bool host_out = false;               // the pass streams its windows to the host (pinned staging is sized per window)
    // Hits per window of THIS pass.  The handle's default (2^30, r4) is for device-resident passes, where a window is only a unit of
    // launches: at config 3 the pass needs 139 windows instead of 553 and saves ~8 ms of tiles_kernel launches and gaps
    // (profiles/r04a_packed_vs_window_size.jsonl).  Passes that stage windows in pinned host memory, and the delivery stage (whose
    // candidate lists are sized per window), keep 2^28 (device-resident delivery passes: 2^27, r5r) unless the caller configured something smaller.
    uint64_t window_cap() const {
        uint64_t c = h->cfg.window_hits;
        // RGR_WINDOW_HITS (A/B switch of bench.py --ab-env, read per window): hits per window of a device-resident pass whose handle took the default
        if (!h->cfg_window_explicit && !host_out && !deliver)
            if (const char* e = std::getenv("RGR_WINDOW_HITS")) { const unsigned long long v = std::strtoull(e, nullptr, 10); if (v >= 4096 && v <= (1ull << 32) - 4096) c = v; }
        if (deliver && !host_out && !h->cfg_window_explicit) {    // RGR_DELIVER_WINDOW_HITS (A/B switch): device-resident delivery passes
            if (const char* e = std::getenv("RGR_DELIVER_WINDOW_HITS")) { const unsigned long long v = std::strtoull(e, nullptr, 10); if (v >= 4096 && v <= (1ull << 32) - 4096) return v; }
            // r5q / r5r: 2^27 — the delivery expansion runs 0.328 ms per 2^27-hit window against 0.733 per 2^28 (its tile records, count
            // words and candidate slices of a window stay closer to L2), the dedup's four launches per window cost 0.187 against 0.352:
            // 16.31 -> 16.84 M matches/s; 2^26: 15.0 M, 2^29: 15.4 M, 2^30: 15.0 M (profiles/r05r_*, r05q_*, r05i_*)
            // r8o: with this round's dedup passes (0.188 -> 0.129 ms per 2^27-hit window) and 8-byte hits the balance moved: 2^28-hit windows run the
            // 8-byte form at 21.42 M against 21.04 M (2^29: 20.65, 2^30: 20.03; profiles/r08o_*).  12-byte delivery tuples keep 2^27.
            return std::min<uint64_t>(c, format == kFmtDeliver8 ? 1ull << 28 : 1ull << 27);
        }
        return (host_out || deliver) ? std::min<uint64_t>(c, h->cfg_window_explicit ? c : (1ull << 28)) : c;
    }
    // streamed passes (rgr_batch_run_to_host, rgr_match_batch): copy stream + per-slot events, created once
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_expanded[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr};
    void ensure_stream_state() {
        if (copy_stream) return;
        RGR_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) { RGR_HIP(hipEventCreateWithFlags(&ev_expanded[k], hipEventDisableTiming)); RGR_HIP(hipEventCreateWithFlags(&ev_copied[k], hipEventDisableTiming)); }
    }
    // delivery stage (rgr_batch_set_publish_attrs)
    bool deliver = false;
    bool group_by_node = false;          // host-out deliver calls: partition every topic's tuples by node + directory (rgr_node_groups)
    DevBuf node_tmp[2], grp_cnt, grp_off, grp_node, grp_begin;      // node_tmp: one per window slot (the D2H copy of window k overlaps window k+1)
    std::vector<uint64_t> h_grp_off, h_grp_begin;     // directory of the window rgr_batch_next_window returned last
    std::vector<uint32_t> h_grp_node;
    int format = kFmtTuple;              // rgr_batch_set_format
    bool has_topic_ids = false;          // rgr_batch_set_topic_ids
    DevBuf d_topic_ids;
    DevBuf d_pub_in;                     // the attributes as the caller gave them (batch order); d_pub is what the kernels index: the same, or gathered into walk order
    // dedup_scalars: [0] unused, [1] the two work-item counters; dedup_stat: candidates seen, one u64 per block of the tile pass (kDedupStatSlots of them:
    // a block adds to ITS slot with a plain read-modify-write — launches are stream-ordered — instead of 2 048 atomics on one address per window)
    DevBuf d_pub, cand, cand_count, dedup_items, dedup_scalars, dedup_stat;
    PinnedBuf h_pub_stage;             // rgr_batch_set_publish_attrs of a small batch: the caller's array, staged for an upload nobody waits for
    DevBuf rf_filter[2], rf_node[2], r_cnt, r_payload, r_ecnt, r_e0, r_e1, r_out_off, r_epos, r_end, r_depth;   // retain frontier rounds
    // pass state
    bool retain = false;             // batch of SUBSCRIBE filters against the retained-topic trie
    bool retain_positions = false;   // rgr_batch_set_retain_positions: tuples carry positions in the epoch's value array instead of topic ids
    uint32_t tier = 0;               // two-tier mode: 0 = base epoch, 1 = delta epoch
    std::shared_ptr<Epoch> epoch;
    std::shared_ptr<RetainEpoch> repoch;
    std::shared_ptr<const std::vector<SubEntry>> retain_vals_pin;      // mirror handed out by rgr_batch_retain_vals
    bool in_pass = false;
    uint32_t cursor = 0;
    uint64_t hits_before = 0;        // hits emitted by earlier windows of this pass
    // timing
    struct Span { hipEvent_t a, b; int kind; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> event_pool;
    rgr_stats local{};               // accumulated over the pass, merged into the handle at the end

    hipEvent_t get_event() {
        if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
        hipEvent_t e;
        // timing only: no system-scope fence when the event completes (the default flags make every record a cache-flushing release;
        // 4 records per window were 1.8 ms of a 116 ms packed pass, profiles/r04b_*).  RGR_SPAN_FENCE=1 restores the default events.
        static const bool fence = std::getenv("RGR_SPAN_FENCE") != nullptr;
        RGR_HIP(hipEventCreateWithFlags(&e, fence ? hipEventDefault : hipEventDisableSystemFence));
        return e;
    }
    // RGR_SPAN_SAMPLE=0 (diagnostic): no event pairs around the per-window launches — what the per-kernel timing itself costs a pass
    static bool spans_on() { static const bool on = [] { const char* e = std::getenv("RGR_SPAN_SAMPLE"); return !(e && e[0] == '0'); }(); return on; }
    static constexpr size_t kNoSpan = ~size_t(0);
    size_t span_begin(int kind, hipStream_t on = nullptr) {
        if (!spans_on()) return kNoSpan;
        Span s{get_event(), get_event(), kind};
        RGR_HIP(hipEventRecord(s.a, on ? on : stream));
        spans.push_back(s);
        return spans.size() - 1;
    }
    void span_end(size_t i, hipStream_t on = nullptr) { if (i != kNoSpan) RGR_HIP(hipEventRecord(spans[i].b, on ? on : stream)); }
    ChunkSlot* other_slot() { return c == &cs[0] ? &cs[1] : &cs[0]; }
    void resolve_spans() {   // stream must be synchronised
        for (auto& s : spans) {
            float ms = 0;
            RGR_HIP(hipEventElapsedTime(&ms, s.a, s.b));
            if (s.kind == kSpanWalk) local.walk_ms += ms;
            else if (s.kind == kSpanScan) local.scan_ms += ms;
            else if (s.kind == kSpanDedup) local.dedup_ms += ms;
            else local.expand_ms += ms;
            event_pool.push_back(s.a); event_pool.push_back(s.b);
        }
        spans.clear();
    }
    ~rgr_batch() {
        for (auto& s : spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
        for (auto e : event_pool) (void)hipEventDestroy(e);
        for (int k = 0; k < 2; ++k) { if (ev_expanded[k]) (void)hipEventDestroy(ev_expanded[k]); if (ev_copied[k]) (void)hipEventDestroy(ev_copied[k]); }
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        if (prep_stream) { (void)hipStreamSynchronize(prep_stream); (void)hipStreamDestroy(prep_stream); }
        if (ev_windows_done) (void)hipEventDestroy(ev_windows_done);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

// scalars layout (device, 32 bytes): [0] u32 ovf_count, [1] u32 error, [2..3] u64 ovf_cursor, [4..5] u64 visited
struct Scalars { uint32_t ovf_count, error; unsigned long long ovf_cursor, visited; uint32_t big_count, pad; };

std::shared_ptr<Epoch> current_epoch(rgr_handle* h) {
    std::lock_guard<std::mutex> g(h->epoch_mu);
    return h->epoch;
}

void merge_stats(rgr_handle* h, rgr_stats& l) {
    std::lock_guard<std::mutex> g(h->stats_mu);
    rgr_stats& s = h->stats;
    s.topics += l.topics; s.invalid_topics += l.invalid_topics; s.levels += l.levels; s.pairs += l.pairs; s.hits += l.hits;
    s.visited_nodes += l.visited_nodes; s.overflow_topics += l.overflow_topics;
    s.walk_launches += l.walk_launches; s.expand_launches += l.expand_launches;
    s.walk_ms += l.walk_ms; s.scan_ms += l.scan_ms; s.expand_ms += l.expand_ms;
    s.tokenize_ms += l.tokenize_ms; s.h2d_ms += l.h2d_ms; s.d2h_ms += l.d2h_ms;
    s.alg_bytes_walk += l.alg_bytes_walk; s.alg_bytes_expand += l.alg_bytes_expand;
    s.dedup_candidates += l.dedup_candidates; s.dedup_launches += l.dedup_launches; s.dedup_ms += l.dedup_ms;
    l = rgr_stats{};
}

const TrieView& batch_view(const rgr_batch* b) { return b->retain ? b->repoch->tv : b->epoch->view; }

void tokenize_batch(rgr_handle* h, rgr_batch* b, const uint8_t* blob, const uint64_t* offs, uint32_t n) {
    const double t0 = now_ms();
    unsigned nt = h->cfg.host_threads ? h->cfg.host_threads : std::max(1u, std::thread::hardware_concurrency());
    nt = std::min<unsigned>(nt, std::max<uint32_t>(1, n / 4096));
    struct Part { std::vector<uint32_t> toks; std::vector<uint32_t> lens; std::vector<uint8_t> flags; };
    std::vector<Part> parts(nt);
    {
        std::shared_lock<std::shared_mutex> lk(b->retain ? h->retain_mu : h->table_mu);
        auto work = [&](unsigned k) {
            Part& p = parts[k];
            const uint64_t lo = uint64_t(n) * k / nt, hi = uint64_t(n) * (k + 1) / nt;
            p.lens.reserve(hi - lo); p.flags.reserve(hi - lo); p.toks.reserve((hi - lo) * 10);
            for (uint64_t i = lo; i < hi; ++i) {
                const size_t mark = p.toks.size();
                const std::string_view sv(reinterpret_cast<const char*>(blob) + offs[i], offs[i + 1] - offs[i]);
                const uint8_t fl = b->retain ? h->retain_tokenizer(b->tier).tokenize_filter(sv, p.toks) : h->table.tokenize_topic(sv, p.toks);
                p.flags.push_back(fl);
                p.lens.push_back(uint32_t(p.toks.size() - mark));
            }
        };
        if (nt == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (unsigned k = 0; k < nt; ++k) th.emplace_back(work, k);
            for (auto& t : th) t.join();
        }
    }
    uint64_t total = 0;
    for (auto& p : parts) total += p.toks.size();
    std::vector<uint32_t> toks;
    toks.reserve(total + 1);
    std::vector<uint64_t> toff(size_t(n) + 1, 0);
    std::vector<uint8_t> flags;
    flags.reserve(n);
    b->status.assign(n, RGR_TOPIC_OK);
    b->valid_topics = 0; b->valid_levels = 0;
    uint64_t ti = 0;
    for (auto& p : parts) {
        toks.insert(toks.end(), p.toks.begin(), p.toks.end());
        for (size_t j = 0; j < p.lens.size(); ++j, ++ti) {
            toff[ti + 1] = toff[ti] + p.lens[j];
            flags.push_back(p.flags[j]);
            if (p.flags[j] & kTopicInvalid) b->status[ti] = RGR_TOPIC_INVALID;
            else { b->valid_topics++; b->valid_levels += p.lens[j]; }
        }
    }
    b->total_tokens = total;
    {
        std::shared_lock<std::shared_mutex> lk(b->retain ? h->retain_mu : h->table_mu);
        b->dict_tokens = b->retain ? h->retain_tokenizer(b->tier).dict().size() : h->table.dict_stamp();
    }
    b->local.tokenize_ms += now_ms() - t0;
    const double t1 = now_ms();
    b->d_tokens.ensure(std::max<uint64_t>(1, total) * 4);
    b->d_tok_off.ensure((size_t(n) + 1) * 8);
    b->d_tflags.ensure(std::max<uint32_t>(1, n));
    b->d_path.ensure(std::max<uint64_t>(1, total) * (b->retain ? 8 : 4));
    if (total) RGR_HIP(hipMemcpyAsync(b->d_tokens.p, toks.data(), total * 4, hipMemcpyHostToDevice, b->stream));
    RGR_HIP(hipMemcpyAsync(b->d_tok_off.p, toff.data(), (size_t(n) + 1) * 8, hipMemcpyHostToDevice, b->stream));
    if (n) RGR_HIP(hipMemcpyAsync(b->d_tflags.p, flags.data(), n, hipMemcpyHostToDevice, b->stream));
    RGR_HIP(hipStreamSynchronize(b->stream));
    b->local.h2d_ms += now_ms() - t1;
}

// Device tokeniser: Topic::from_str + dictionary lookup on the GPU against `dict`.
void tokenize_batch_device(rgr_batch* b, const DictImage& dict) {
    const double t0 = now_ms();
    const uint32_t n = b->n;
    b->d_level_cnt.ensure(std::max<uint32_t>(1, n) * 4);
    b->d_tflags.ensure(std::max<uint32_t>(1, n));
    b->d_tok_off.ensure((size_t(n) + 1) * 8);
    b->scan_tmp.ensure((size_t(n) / scan_block_topics() + 3) * 16);
    const uint8_t* blob = b->d_blob.as<uint8_t>();
    const uint64_t* offs = b->d_offs.as<uint64_t>();
    launch_tok_count(blob, offs, n, b->d_level_cnt.as<uint32_t>(), b->d_tflags.as<uint8_t>(), b->stream,
                     b->from_publish ? b->d_force.as<uint8_t>() : nullptr);
    launch_scan_u32(b->d_level_cnt.as<uint32_t>(), b->d_tok_off.as<uint64_t>(), n, b->scan_tmp.as<uint64_t>(), b->stream);
    uint64_t total = 0;
    std::vector<uint8_t> flags(n);
    // (r7) A micro-batch does not wait for its token count between the two kernels: a topic of b bytes has at most b + 1 levels, so the arrays are
    // sized by that bound and count, scan and fill run back to back — one host round trip less in a pass that is made of them (a few thousand
    // publishes: 7.5 synchronisations per pass, profiles/r07e_*).  Larger batches keep the exact size (10 M topics: 0.3 GB instead of 2 GB).
    const uint64_t bound = (b->blob_bytes) + n;
    const bool small = n != 0 && bound * 12 <= (64ull << 20);
    if (!small) {
        RGR_HIP(hipMemcpyAsync(&total, b->d_tok_off.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, b->stream));
        if (n) RGR_HIP(hipMemcpyAsync(flags.data(), b->d_tflags.p, n, hipMemcpyDeviceToHost, b->stream));
        RGR_HIP(hipStreamSynchronize(b->stream));
    }
    const uint64_t cap_tokens = small ? bound : total;
    b->d_tokens.ensure(std::max<uint64_t>(1, cap_tokens) * 4);
    b->d_path.ensure(std::max<uint64_t>(1, cap_tokens) * (b->retain ? 8 : 4));
    launch_tok_fill(dict.view, blob, offs, n, b->d_tok_off.as<uint64_t>(), b->d_tflags.as<uint8_t>(), b->d_tokens.as<uint32_t>(), b->stream);
    if (small) {
        RGR_HIP(hipMemcpyAsync(&total, b->d_tok_off.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, b->stream));
        RGR_HIP(hipMemcpyAsync(flags.data(), b->d_tflags.p, n, hipMemcpyDeviceToHost, b->stream));
    }
    RGR_HIP(hipStreamSynchronize(b->stream));
    RGR_HIP(hipGetLastError());
    b->status.assign(n, RGR_TOPIC_OK);
    b->valid_topics = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (flags[i] & kTopicInvalid) b->status[i] = RGR_TOPIC_INVALID; else b->valid_topics++;
    }
    if (b->from_publish)
        for (uint32_t i = 0; i < n; ++i) if (b->pub_info[i].error) b->status[i] = RGR_PACKET_MALFORMED;
    b->total_tokens = total;
    b->valid_levels = total;
    b->dict_tokens = dict.n_tokens;
    b->local.tokenize_ms += now_ms() - t0;
}

// Walk order (rgr_batch_set_order): sort the batch's topics by their leading tokens and gather the token arrays into that order.  Runs after every
// (re)tokenisation of an ordered batch; the stream is synchronised on return (the permutation is mirrored on the host for rgr_batch_topic_order).
void apply_order(rgr_batch* b) {
    const uint32_t n = b->n;
    b->h_order.assign(n, 0);
    if (!n) return;
    const size_t temp_bytes = order_sort_temp_bytes(n);
    b->order_tmp[0].ensure(size_t(n) * 8); b->order_tmp[1].ensure(size_t(n) * 8); b->order_tmp[2].ensure(size_t(n) * 4); b->order_tmp[3].ensure(std::max<size_t>(16, temp_bytes));
    b->d_order.ensure(size_t(n) * 4);
    if (launch_order_sort(b->d_tokens.as<uint32_t>(), b->d_tok_off.as<uint64_t>(), b->d_tflags.as<uint8_t>(), n, b->order_tmp[0].as<unsigned long long>(),
                          b->order_tmp[1].as<unsigned long long>(), b->order_tmp[2].as<uint32_t>(), b->d_order.as<uint32_t>(), b->order_tmp[3].p, temp_bytes, b->stream) != 0)
        throw std::runtime_error("rgr_batch_set_order: the radix sort failed");
    // lengths in walk order -> offsets -> tokens / flags
    b->s_tok_off.ensure((size_t(n) + 1) * 8);
    b->s_tflags.ensure(n);
    b->s_tokens.ensure(std::max<uint64_t>(1, b->total_tokens) * 4);
    b->scan_tmp.ensure((size_t(n) / scan_block_topics() + 3) * 16);
    uint32_t* len = b->order_tmp[2].as<uint32_t>();                      // (the sort is done with its index scratch)
    launch_order_len(b->d_order.as<uint32_t>(), b->d_tok_off.as<uint64_t>(), n, len, b->stream);
    launch_scan_u32(len, b->s_tok_off.as<uint64_t>(), n, b->scan_tmp.as<uint64_t>(), b->stream);
    launch_order_gather(b->d_order.as<uint32_t>(), n, b->d_tok_off.as<uint64_t>(), b->d_tokens.as<uint32_t>(), b->d_tflags.as<uint8_t>(), b->s_tok_off.as<uint64_t>(),
                        b->s_tokens.as<uint32_t>(), b->s_tflags.as<uint8_t>(), b->stream);
    if (b->has_topic_ids) {
        b->d_order_ids.ensure(size_t(n) * 4);
        launch_order_compose(b->d_order.as<uint32_t>(), b->d_topic_ids.as<uint32_t>(), n, b->d_order_ids.as<uint32_t>(), b->stream);
    }
    if (b->deliver && b->d_pub_in.p) {           // the delivery stage indexes its attributes by walk position
        b->d_pub.ensure(size_t(n) * sizeof(PublishAttr));
        launch_order_gather_attrs(b->d_order.as<uint32_t>(), b->d_pub_in.as<PublishAttr>(), n, b->d_pub.as<PublishAttr>(), b->stream);
    }
    RGR_HIP(hipMemcpyAsync(b->h_order.data(), b->d_order.p, size_t(n) * 4, hipMemcpyDeviceToHost, b->stream));
    RGR_HIP(hipStreamSynchronize(b->stream));
    RGR_HIP(hipGetLastError());
}

WalkArgs make_walk_args(rgr_batch* b, uint32_t n) {
    WalkArgs a{};
    Scalars* sc = b->c->scalars.as<Scalars>();
    a.tokens = (b->ordered ? b->s_tokens : b->d_tokens).as<uint32_t>();
    a.tok_off = (b->ordered ? b->s_tok_off : b->d_tok_off).as<uint64_t>();
    a.tflags = (b->ordered ? b->s_tflags : b->d_tflags).as<uint8_t>();
    a.topic_base = b->c->begin;
    a.n = n;
    a.slot_cap = b->retain ? 0 : b->h->cfg.slot_cap;
    a.slots = b->c->slots.as<uint32_t>();
    a.pair_cnt = b->c->pair_cnt.as<uint32_t>();
    a.path_scratch = b->d_path.as<uint32_t>();
    a.visited = b->h->cfg.collect_walk_stats ? &sc->visited : nullptr;
    a.ovf_list = b->c->ovf_list.as<uint32_t>();
    a.ovf_count = &sc->ovf_count;
    a.ovf_base = b->c->ovf_base.as<uint64_t>();
    a.ovf_cursor = &sc->ovf_cursor;
    a.ovf_arena = b->c->arena.as<uint32_t>();
    a.ovf_arena_cap = b->c->arena_cap;
    return a;
}

ChunkArrays make_chunk_arrays(rgr_batch* b, uint32_t n) {
    ChunkArrays c{};
    Scalars* sc = b->c->scalars.as<Scalars>();
    c.n = n;
    c.slot_cap = b->retain ? 0 : b->h->cfg.slot_cap;   // retain descriptor lists always live in the arena
    c.slots = b->c->slots.as<uint32_t>();
    c.pair_cnt = b->c->pair_cnt.as<uint32_t>();
    c.hit_cnt = b->c->hit_cnt.as<uint32_t>();
    c.pair_live = b->c->pair_live.as<uint32_t>();
    c.hit_off = b->c->hit_off.as<uint64_t>();
    c.pair_base = b->c->pair_base.as<uint64_t>();
    c.ovf_base = b->c->ovf_base.as<uint64_t>();
    c.ovf_arena = b->c->arena.as<uint32_t>();
    c.ovf_arena_cap = b->c->arena_cap;
    c.error_flag = &sc->error;
    c.big_list = b->c->r_big.as<uint32_t>();
    c.big_count = &sc->big_count;
    c.pair_src = b->c->pair_src.as<uint32_t>();
    c.pair_topic = b->c->pair_topic.as<uint32_t>();
    c.pair_off = b->c->pair_off.as<uint64_t>();
    if (b->deliver && !b->retain) { c.pub = b->d_pub.as<PublishAttr>(); c.pair_qr = b->c->pair_qr.as<uint8_t>(); }
    else if (b->ordered) c.topic_ids = (b->has_topic_ids ? b->d_order_ids : b->d_order).as<uint32_t>();      // walk position -> the caller's index (or the caller's id of it)
    else if (b->has_topic_ids) c.topic_ids = b->d_topic_ids.as<uint32_t>();
    return c;
}

void ensure_chunk_buffers(rgr_batch* b, uint32_t n) {
    const uint32_t C = b->retain ? 0 : b->h->cfg.slot_cap;
    b->c->slots.ensure(std::max<size_t>(16, size_t(C) * n * 4));
    b->c->pair_cnt.ensure(size_t(n) * 4);
    b->c->hit_cnt.ensure(size_t(n) * 4);
    b->c->pair_live.ensure(size_t(n) * 4);
    b->c->hit_off.ensure((size_t(n) + 1) * 8);
    b->c->pair_base.ensure((size_t(n) + 1) * 8);
    b->c->ovf_list.ensure(size_t(n) * 4);
    b->c->ovf_base.ensure(size_t(n) * 8);
    b->c->scalars.ensure(sizeof(Scalars));
    b->c->r_big.ensure(size_t(n) * 4);
    if (b->c->arena_cap == 0) {
        const char* e = std::getenv("RGR_ARENA_INIT");        // tests shrink it to exercise the grow-and-redo paths
        b->c->arena_cap = e && std::atoll(e) > 0 ? uint64_t(std::atoll(e)) : (1u << 20);
        b->c->arena.ensure(b->c->arena_cap * 4);
    }
    const uint32_t nb = (n + scan_block_topics() - 1) / scan_block_topics();
    b->scan_tmp.ensure((size_t(nb) + 1) * 16);
    b->c->scan_tmp.ensure((size_t(nb) + 1) * 16);
    b->c->h_hit_off.ensure((size_t(n) + 1) * 8);
    b->c->h_pair_base.ensure((size_t(n) + 1) * 8);
    b->c->h_scalars.ensure(sizeof(Scalars) + 16);      // + chunk totals (hits, pairs)
}

// RetainTree::matches for the chunk's filters: level-synchronous frontier rounds (kernels.hip).
// Fills the arena with every filter's run descriptors (in trie preorder), ovf_base / pair_cnt.
void retain_rounds(rgr_batch* b, uint32_t begin, uint32_t n) {
    const RetainView& rv = b->repoch->view;
    Scalars* sc = b->c->scalars.as<Scalars>();
    b->r_end.ensure(size_t(n) * 8);
    b->r_depth.ensure(std::max<size_t>(1, n) * 4);
    RGR_HIP(hipMemsetAsync(b->r_depth.p, 0, size_t(n) * 4, b->stream));
    uint64_t g_total = 0, m = n, visited = 0;
    int cur = 0;
    for (uint32_t d = 0; m > 0; ++d) {
        if (m > (1ull << 31)) throw std::runtime_error("retain frontier exceeds 2^31 items");
        const uint32_t mm = uint32_t(m);
        for (DevBuf* p : {&b->r_cnt, &b->r_payload, &b->r_ecnt, &b->r_e0, &b->r_e1, &b->c->r_big}) p->ensure(size_t(mm) * 4);
        b->r_out_off.ensure((size_t(mm) + 1) * 8);
        b->r_epos.ensure((size_t(mm) + 1) * 8);
        b->scan_tmp.ensure((size_t(mm) / scan_block_topics() + 3) * 16);
        RetainRound r{};
        r.tokens = b->d_tokens.as<uint32_t>(); r.tok_off = b->d_tok_off.as<uint64_t>(); r.tflags = b->d_tflags.as<uint8_t>();
        r.topic_base = begin; r.fdepth = b->r_depth.as<uint32_t>(); r.m = mm;
        r.f_filter = d == 0 ? nullptr : b->rf_filter[cur].as<uint32_t>();
        r.f_node = d == 0 ? nullptr : b->rf_node[cur].as<uint32_t>();
        r.cnt = b->r_cnt.as<uint32_t>(); r.payload = b->r_payload.as<uint32_t>();
        r.ecnt = b->r_ecnt.as<uint32_t>(); r.e0 = b->r_e0.as<uint32_t>(); r.e1 = b->r_e1.as<uint32_t>();
        launch_retain_step(rv, r, b->stream);
        launch_scan_u32(r.cnt, b->r_out_off.as<uint64_t>(), mm, b->scan_tmp.as<uint64_t>(), b->stream);
        launch_scan_u32(r.ecnt, b->r_epos.as<uint64_t>(), mm, b->scan_tmp.as<uint64_t>(), b->stream);
        uint64_t tot[2] = {0, 0};
        RGR_HIP(hipMemcpyAsync(&tot[0], b->r_out_off.as<uint64_t>() + mm, 8, hipMemcpyDeviceToHost, b->stream));
        RGR_HIP(hipMemcpyAsync(&tot[1], b->r_epos.as<uint64_t>() + mm, 8, hipMemcpyDeviceToHost, b->stream));
        RGR_HIP(hipStreamSynchronize(b->stream));
        const uint64_t m_next = tot[0], e_r = tot[1];
        if (g_total + e_r > b->c->arena_cap) {
            b->c->arena_cap = (g_total + e_r) * 2;
            b->c->arena.ensure_preserve(b->c->arena_cap * 4, g_total * 4);
        }
        launch_retain_emit(r, b->r_epos.as<uint64_t>(), g_total, b->c->arena.as<uint32_t>(), b->c->ovf_base.as<uint64_t>(),
                           b->r_end.as<uint64_t>(), b->stream);
        g_total += e_r;
        if (m_next) {
            b->rf_filter[cur ^ 1].ensure(m_next * 4);
            b->rf_node[cur ^ 1].ensure(m_next * 4);
            RGR_HIP(hipMemsetAsync(&sc->big_count, 0, 4, b->stream));
            launch_retain_next(rv, r, b->r_out_off.as<uint64_t>(), b->rf_filter[cur ^ 1].as<uint32_t>(), b->rf_node[cur ^ 1].as<uint32_t>(),
                               b->c->r_big.as<uint32_t>(), &sc->big_count, b->stream);
        }
        launch_retain_advance(r, n, b->r_depth.as<uint32_t>(), b->stream);
        visited += m;
        cur ^= 1;
        m = m_next;
    }
    launch_retain_finish(n, b->c->ovf_base.as<uint64_t>(), b->r_end.as<uint64_t>(), b->c->pair_cnt.as<uint32_t>(), b->stream);
    RGR_HIP(hipGetLastError());
    b->local.visited_nodes += visited;
    b->local.alg_bytes_walk += 24 * visited;
}

// per-topic hit offsets / pair bases of the current chunk on the host (window planning, host-side offsets)
void fetch_host_arrays(rgr_batch* b) {
    if (b->c->host_arrays) return;
    const uint32_t n = b->c->n;
    RGR_HIP(hipMemcpyAsync(b->c->h_hit_off.p, b->c->hit_off.p, (size_t(n) + 1) * 8, hipMemcpyDeviceToHost, b->stream));
    RGR_HIP(hipMemcpyAsync(b->c->h_pair_base.p, b->c->pair_base.p, (size_t(n) + 1) * 8, hipMemcpyDeviceToHost, b->stream));
    RGR_HIP(hipStreamSynchronize(b->stream));
    b->c->host_arrays = true;
}

// accounting of a prepared chunk
void account_chunk(rgr_batch* b) {
    const uint32_t n = b->c->n;
    const Scalars* hs = b->c->h_scalars.as<Scalars>();
    const uint64_t P = b->c->total_pairs, H = b->c->total_hits;
    (void)n;
    b->local.pairs += P;
    b->local.hits += H;
    b->local.visited_nodes += hs->visited;
    b->local.overflow_topics += hs->ovf_count;
    b->local.alg_bytes_walk += 24 * hs->visited + 8 * P;
    b->local.alg_bytes_expand += 20 * H;
}

// Topics of the chunk that starts at `begin`.  The first chunk of a pass is the one piece of preparation nothing can overlap with
// (every later chunk is walked on the prefetch stream behind the previous chunk's expansions): when more chunks follow it is an
// eighth of the configured size, so the first window starts expanding after ~0.4 ms instead of ~3 ms at config 3 (r4).
uint32_t chunk_len(const rgr_batch* b, uint32_t begin) {
    uint32_t c = b->h->cfg.chunk_topics;
    if (begin == 0 && !b->retain && b->n > c) c = std::max<uint32_t>(c / 8, 4096u);
    return std::min<uint32_t>(c, b->n - begin);
}

// Walk the chunk starting at `begin`; on return the host has hit_off / pair_base and the
// dense pair arrays are built.  With walk_only the pipeline stops after the walk passes.
void prepare_chunk(rgr_batch* b, uint32_t begin, bool walk_only) {
    const uint32_t n = chunk_len(b, begin);
    b->c->begin = begin;
    b->c->n = n;
    ensure_chunk_buffers(b, n);
    const TrieView& tv = batch_view(b);
    for (;;) {
        RGR_HIP(hipMemsetAsync(b->c->scalars.p, 0, sizeof(Scalars), b->stream));
        WalkArgs wa = make_walk_args(b, n);
        size_t sp = b->span_begin(kSpanWalk);
        if (b->retain) {
            retain_rounds(b, begin, n);
        } else {
            launch_walk(tv, wa, false, b->stream);
            launch_walk(tv, wa, true, b->stream);
        }
        b->span_end(sp);
        b->local.walk_launches++;
        RGR_HIP(hipGetLastError());
        if (walk_only) {
            RGR_HIP(hipMemcpyAsync(b->c->h_scalars.p, b->c->scalars.p, sizeof(Scalars), hipMemcpyDeviceToHost, b->stream));
            RGR_HIP(hipStreamSynchronize(b->stream));
            const Scalars* hs = b->c->h_scalars.as<Scalars>();
            if (hs->ovf_cursor > b->c->arena_cap) {   // arena too small: grow and redo
                b->c->arena_cap = hs->ovf_cursor * 2;
                b->c->arena.ensure(b->c->arena_cap * 4);
                b->resolve_spans();
                continue;
            }
            b->resolve_spans();
            b->c->ready = true;
            return;
        }
        ChunkArrays ca = make_chunk_arrays(b, n);
        sp = b->span_begin(kSpanScan);
        launch_count(tv, ca, b->stream);
        launch_scan(ca, b->c->scan_tmp.as<uint64_t>(), b->stream);
        b->span_end(sp);
        // the totals first (24 bytes); the per-topic offset arrays (16 B per topic) cross PCIe only if the chunk needs
        // more than one window — at low fan-out (config 2: 0.46 hits per topic) they were a third of the pass
        uint64_t* tot = b->c->h_scalars.as<uint64_t>() + sizeof(Scalars) / 8;
        RGR_HIP(hipMemcpyAsync(b->c->h_scalars.p, b->c->scalars.p, sizeof(Scalars), hipMemcpyDeviceToHost, b->stream));
        RGR_HIP(hipMemcpyAsync(&tot[0], b->c->hit_off.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, b->stream));
        RGR_HIP(hipMemcpyAsync(&tot[1], b->c->pair_base.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, b->stream));
        // (r7) a small chunk whose caller wants the offsets on the host anyway (the host-staged calls) takes them in THIS round trip
        const bool arrays_now = b->want_host_offsets && n <= 65536u;
        if (arrays_now) {
            RGR_HIP(hipMemcpyAsync(b->c->h_hit_off.p, b->c->hit_off.p, (size_t(n) + 1) * 8, hipMemcpyDeviceToHost, b->stream));
            RGR_HIP(hipMemcpyAsync(b->c->h_pair_base.p, b->c->pair_base.p, (size_t(n) + 1) * 8, hipMemcpyDeviceToHost, b->stream));
        }
        RGR_HIP(hipStreamSynchronize(b->stream));
        RGR_HIP(hipGetLastError());
        b->resolve_spans();
        const Scalars* hs = b->c->h_scalars.as<Scalars>();
        if (hs->error || hs->ovf_cursor > b->c->arena_cap) {
            b->c->arena_cap = std::max<uint64_t>(b->c->arena_cap * 2, hs->ovf_cursor * 2);
            b->c->arena.ensure(b->c->arena_cap * 4);
            continue;
        }
        const uint64_t H = tot[0], P = tot[1];
        b->c->total_hits = H; b->c->total_pairs = P;
        b->c->host_arrays = arrays_now;
        if (H > b->window_cap() || b->want_host_offsets) fetch_host_arrays(b);
        b->c->pair_src.ensure(std::max<uint64_t>(1, P) * 4);
        b->c->pair_topic.ensure(std::max<uint64_t>(1, P) * 4);
        b->c->pair_off.ensure((P + 1) * 8);
        if (b->deliver && !b->retain) b->c->pair_qr.ensure(std::max<uint64_t>(1, P));
        ca = make_chunk_arrays(b, n);
        sp = b->span_begin(kSpanScan);
        launch_compact(tv, ca, begin, b->stream);
        b->span_end(sp);
        account_chunk(b);
        b->c->ready = true;
        return;
    }
}

// The NEXT chunk, prepared while the current chunk's windows are being expanded: walk -> count -> scan -> compact are
// enqueued on prep_stream into the idle slot with NO host synchronisation — the pair arrays are sized by their upper
// bound (n * slot_cap + arena capacity) instead of the scanned total — and the host-side arrays (hit offsets, pair
// bases, scalars) follow by asynchronous copies; `done` fires when all of it has landed.  rgr_batch_next_window
// adopts the slot when the cursor reaches it (an arena overflow, detected then, redoes the chunk synchronously).
// Walk (random gathers) and expansion (streaming stores) overlap well: at config 3 the per-chunk preparation was
// ~10 % of a 12-byte pass and ~28 % of a 4-byte one.
void prefetch_chunk(rgr_batch* b, uint32_t begin) {
    rgr_handle* h = b->h;
    if (b->retain || begin >= b->n || std::getenv("RGR_NO_PREFETCH")) return;
    ChunkSlot* cur = b->c;
    struct Back { rgr_batch* b; ChunkSlot* c; ~Back() { b->c = c; } } back{b, cur};
    b->c = b->other_slot();
    ChunkSlot& nx = *b->c;
    const uint32_t n = chunk_len(b, begin);
    nx.begin = begin; nx.n = n; nx.ready = false; nx.inflight = false;
    if (!b->prep_stream) RGR_HIP(hipStreamCreateWithFlags(&b->prep_stream, hipStreamNonBlocking));
    if (!b->ev_windows_done) RGR_HIP(hipEventCreateWithFlags(&b->ev_windows_done, hipEventDisableTiming));
    if (!nx.done) RGR_HIP(hipEventCreateWithFlags(&nx.done, hipEventDisableTiming));
    ensure_chunk_buffers(b, n);
    const uint64_t pmax = uint64_t(n) * h->cfg.slot_cap + nx.arena_cap;
    nx.pair_src.ensure(std::max<uint64_t>(1, pmax) * 4);
    nx.pair_topic.ensure(std::max<uint64_t>(1, pmax) * 4);
    nx.pair_off.ensure((pmax + 1) * 8);
    if (b->deliver) nx.pair_qr.ensure(std::max<uint64_t>(1, pmax));
    hipStream_t ps = b->prep_stream;
    // expansions of the chunk this slot held before may still be in flight on the main stream
    RGR_HIP(hipEventRecord(b->ev_windows_done, b->stream));
    RGR_HIP(hipStreamWaitEvent(ps, b->ev_windows_done, 0));
    const TrieView& tv = batch_view(b);
    RGR_HIP(hipMemsetAsync(nx.scalars.p, 0, sizeof(Scalars), ps));
    WalkArgs wa = make_walk_args(b, n);
    size_t sp = b->span_begin(kSpanWalk, ps);
    launch_walk(tv, wa, false, ps);
    launch_walk(tv, wa, true, ps);
    b->span_end(sp, ps);
    b->local.walk_launches++;
    ChunkArrays ca = make_chunk_arrays(b, n);
    sp = b->span_begin(kSpanScan, ps);
    launch_count(tv, ca, ps);
    launch_scan(ca, nx.scan_tmp.as<uint64_t>(), ps);
    launch_compact(tv, ca, begin, ps);
    b->span_end(sp, ps);
    RGR_HIP(hipMemcpyAsync(nx.h_scalars.p, nx.scalars.p, sizeof(Scalars), hipMemcpyDeviceToHost, ps));
    RGR_HIP(hipMemcpyAsync(nx.h_hit_off.p, nx.hit_off.p, (size_t(n) + 1) * 8, hipMemcpyDeviceToHost, ps));
    RGR_HIP(hipMemcpyAsync(nx.h_pair_base.p, nx.pair_base.p, (size_t(n) + 1) * 8, hipMemcpyDeviceToHost, ps));
    RGR_HIP(hipEventRecord(nx.done, ps));
    RGR_HIP(hipGetLastError());
    nx.inflight = true;
}

// Make the chunk starting at `begin` the current one: adopt the prefetched slot when there is one, else prepare it now.
void enter_chunk(rgr_batch* b, uint32_t begin) {
    ChunkSlot* o = b->other_slot();
    if (o->inflight && o->begin == begin) {
        RGR_HIP(hipEventSynchronize(o->done));
        o->inflight = false;
        b->c = o;
        const Scalars* hs = o->h_scalars.as<Scalars>();
        if (hs->error || hs->ovf_cursor > o->arena_cap) {      // overflow arena too small: grow it and redo the chunk on the main stream
            o->arena_cap = std::max<uint64_t>(o->arena_cap * 2, hs->ovf_cursor * 2);
            o->arena.ensure(o->arena_cap * 4);
            b->resolve_spans();
            prepare_chunk(b, begin, false);
        } else {
            o->total_hits = o->h_hit_off.as<uint64_t>()[o->n];
            o->total_pairs = o->h_pair_base.as<uint64_t>()[o->n];
            o->host_arrays = true;
            account_chunk(b);
            o->ready = true;
        }
    } else {
        if (o->inflight) { RGR_HIP(hipEventSynchronize(o->done)); o->inflight = false; }   // (a prefetch nobody came for)
        b->c->ready = false;
        prepare_chunk(b, begin, false);
    }
    prefetch_chunk(b, b->c->begin + b->c->n);
}

}  // namespace

// =============================================================================== C ABI
extern "C" {

const char* rgr_last_error(void) { return g_last_error.c_str(); }
const char* rgr_version(void) {
    static const std::string v = std::string("rmqtt_gpu_router 0.2 (gfx950; tuple expansion: ") + expand_tuple_kernel_name() + "; ids24 expansion: " + expand_ids24_kernel_name() + ")";
    return v.c_str();
}

int32_t rgr_create(const rgr_config* cfg, rgr_handle** out) {
    return guarded([&]() -> int32_t {
        if (!out) return fail(RGR_EINVAL, "rgr_create: out is NULL");
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0) return fail(RGR_EDEVICE, "rgr_create: no HIP device available (there is no CPU fallback)");
        auto h = std::make_unique<rgr_handle>();
        if (cfg) h->cfg = *cfg;
        if (h->cfg.device < 0 || h->cfg.device >= ndev) return fail(RGR_EDEVICE, "rgr_create: bad device ordinal");
        if (!h->cfg.slot_cap) h->cfg.slot_cap = 64;
        h->cfg_window_explicit = h->cfg.window_hits != 0;
        if (!h->cfg.window_hits) h->cfg.window_hits = 1ull << 30;
        if (!h->cfg.chunk_topics) h->cfg.chunk_topics = 1u << 21;
        RGR_HIP(hipSetDevice(h->cfg.device));
        // epoch 0: the empty table
        *out = h.release();
        int32_t rc = rgr_commit(*out);
        if (rc != RGR_OK) { delete *out; *out = nullptr; }
        return rc;
    });
}

void rgr_destroy(rgr_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    for (rgr_batch* b : h->pool) delete b;
    delete h;
}

int32_t rgr_filter_add(rgr_handle* h, const char* filter, uint32_t len, uint32_t* filter_id) {
    return guarded([&]() -> int32_t {
        if (!h || !filter_id || (!filter && len)) return fail(RGR_EINVAL, "rgr_filter_add: bad argument");
        std::unique_lock<std::shared_mutex> lk(h->table_mu);
        int32_t rc = h->table.filter_add(std::string_view(filter, len), filter_id);
        if (rc != RGR_OK) return fail(rc, "rgr_filter_add: invalid topic filter");
        return RGR_OK;
    });
}

int32_t rgr_filter_find(rgr_handle* h, const char* filter, uint32_t len, uint32_t* filter_id) {
    return guarded([&]() -> int32_t {
        if (!h || !filter_id) return fail(RGR_EINVAL, "rgr_filter_find: bad argument");
        std::shared_lock<std::shared_mutex> lk(h->table_mu);
        int32_t rc = h->table.filter_find(std::string_view(filter, len), filter_id);
        if (rc != RGR_OK) return fail(rc, "rgr_filter_find: not found");
        return RGR_OK;
    });
}

int32_t rgr_filter_remove(rgr_handle* h, uint32_t filter_id) {
    return guarded([&]() -> int32_t {
        if (!h) return fail(RGR_EINVAL, "rgr_filter_remove: bad argument");
        std::unique_lock<std::shared_mutex> lk(h->table_mu);
        int32_t rc = h->table.filter_remove(filter_id);
        if (rc == RGR_ESTATE) return fail(rc, "rgr_filter_remove: filter still has subscriptions");
        if (rc != RGR_OK) return fail(rc, "rgr_filter_remove: unknown filter id");
        return RGR_OK;
    });
}

int32_t rgr_sub_add(rgr_handle* h, uint32_t filter_id, uint32_t sub_id, uint8_t qos, uint8_t flags) {
    return guarded([&]() -> int32_t {
        if (!h) return fail(RGR_EINVAL, "rgr_sub_add: bad argument");
        std::unique_lock<std::shared_mutex> lk(h->table_mu);
        int32_t rc = h->table.sub_add(filter_id, sub_id, qos, flags);
        if (rc != RGR_OK) return fail(rc, "rgr_sub_add: unknown filter id");
        return RGR_OK;
    });
}

int32_t rgr_sub_add_ex(rgr_handle* h, uint32_t filter_id, uint32_t sub_id, uint8_t qos, uint8_t flags, uint16_t node_idx,
                       uint32_t owner_id, uint32_t client_idx) {
    return guarded([&]() -> int32_t {
        if (!h) return fail(RGR_EINVAL, "rgr_sub_add_ex: bad argument");
        std::unique_lock<std::shared_mutex> lk(h->table_mu);
        int32_t rc = h->table.sub_add(filter_id, sub_id, qos, flags, node_idx);
        if (rc != RGR_OK) return fail(rc, "rgr_sub_add_ex: unknown filter id");
        h->table.sub_set_attr(sub_id, owner_id, client_idx);
        return RGR_OK;
    });
}

int32_t rgr_sub_attrs_bulk(rgr_handle* h, const uint32_t* sub_ids, const uint32_t* owner_ids, const uint32_t* client_idx, uint64_t n) {
    return guarded([&]() -> int32_t {
        if (!h || (n && (!owner_ids || !client_idx))) return fail(RGR_EINVAL, "rgr_sub_attrs_bulk: bad argument");
        std::unique_lock<std::shared_mutex> lk(h->table_mu);
        for (uint64_t i = 0; i < n; ++i) h->table.sub_set_attr(sub_ids ? sub_ids[i] : uint32_t(i), owner_ids[i], client_idx[i]);
        if (n) h->table.mark_attrs_all_dirty();
        return RGR_OK;
    });
}

int32_t rgr_sub_remove(rgr_handle* h, uint32_t filter_id, uint32_t sub_id) {
    return guarded([&]() -> int32_t {
        if (!h) return fail(RGR_EINVAL, "rgr_sub_remove: bad argument");
        std::unique_lock<std::shared_mutex> lk(h->table_mu);
        int32_t rc = h->table.sub_remove(filter_id, sub_id);
        if (rc != RGR_OK) return fail(rc, "rgr_sub_remove: unknown filter / subscription id");
        return RGR_OK;
    });
}

int32_t rgr_subscribe_bulk(rgr_handle* h, const uint8_t* blob, const uint64_t* offsets, uint64_t n, const uint32_t* sub_ids,
                           const uint8_t* qos, const uint8_t* flags, uint32_t* filter_ids_out, uint64_t* n_rejected) {
    return guarded([&]() -> int32_t {
        if (!h || (n && (!blob || !offsets))) return fail(RGR_EINVAL, "rgr_subscribe_bulk: bad argument");
        std::unique_lock<std::shared_mutex> lk(h->table_mu);
        h->table.subscribe_bulk(blob, offsets, n, sub_ids, qos, flags, filter_ids_out, n_rejected, h->cfg.host_threads);
        return RGR_OK;
    });
}

int32_t rgr_snapshot_save(rgr_handle* h, const char* path) {
    return guarded([&]() -> int32_t {
        if (!h || !path) return fail(RGR_EINVAL, "rgr_snapshot_save: bad argument");
        std::shared_lock<std::shared_mutex> lk(h->table_mu);
        std::string err;
        if (!h->table.save(path, &err)) return fail(RGR_EINVAL, "rgr_snapshot_save: " + err);
        return RGR_OK;
    });
}

int32_t rgr_snapshot_load(rgr_handle* h, const char* path) {
    return guarded([&]() -> int32_t {
        if (!h || !path) return fail(RGR_EINVAL, "rgr_snapshot_load: bad argument");
        std::lock_guard<std::mutex> cg(h->commit_mu);
        std::unique_lock<std::shared_mutex> lk(h->table_mu);
        std::string err;
        if (!h->table.load(path, &err)) return fail(RGR_EINVAL, "rgr_snapshot_load: " + err);
        // every device image is rebuilt by the next commit; epochs still pinned by passes keep theirs alive
        for (int k = 0; k < 2; ++k) { h->edge_img[k].reset(); h->filt_img[k].reset(); }
        h->sub_pool.reset();
        h->host_desc.clear();
        h->pool_garbage = 0;
        return RGR_OK;
    });
}

int32_t rgr_commit(rgr_handle* h) {
    return guarded([&]() -> int32_t {
        if (!h) return fail(RGR_EINVAL, "rgr_commit: bad argument");
        RGR_HIP(hipSetDevice(h->cfg.device));
        std::lock_guard<std::mutex> cg(h->commit_mu);
        auto prev = current_epoch(h);
        auto ep = std::make_shared<Epoch>();
        // A commit that fails half way (device out of memory, HIP error) has already consumed the table's delta:
        // every image is then marked for a full rebuild, so the NEXT commit publishes the complete table instead
        // of silently missing the lost changes.
        struct Recover {
            rgr_handle* h; bool armed = true;
            ~Recover() {
                if (!armed) return;
                for (int k = 0; k < 2; ++k) { if (h->edge_img[k]) h->edge_img[k]->need_full = true; if (h->filt_img[k]) h->filt_img[k]->need_full = true; }
                h->sub_pool.reset(); h->host_desc.clear(); h->pool_garbage = 0;
                    }
        } recover{h};
        const bool prof = std::getenv("RGR_COMMIT_PROFILE") != nullptr;
        double t_prev = now_ms();
        auto lap = [&](const char* what) { if (prof) { const double t = now_ms(); std::fprintf(stderr, "[commit] %-14s %.3f ms\n", what, t - t_prev); t_prev = t; } };
        {
            std::shared_lock<std::shared_mutex> lk(h->table_mu);
            lap("lock");
            HostTable::Delta delta;
            h->table.take_delta(delta);              // (mutators hold table_mu exclusively; this is the only reader of the delta)
            const auto& edges = h->table.edges();
            // ---- dictionary: append-only; re-uploaded only when it grew
            if (prev && prev->dict && prev->dict->n_tokens == h->table.dict_stamp()) ep->dict = prev->dict;
            else { ep->dict = std::make_shared<DictImage>(); ep->dict->upload(h->table.dict(), h->table.dict_stamp()); }
            // ---- every image accumulates the delta; the one not serving the current epoch is patched
            for (int k = 0; k < 2; ++k) {
                if (!h->edge_img[k]) h->edge_img[k] = std::make_shared<EdgeImage>();
                if (!h->filt_img[k]) h->filt_img[k] = std::make_shared<FiltImage>();
                if (delta.relocated) h->edge_img[k]->need_full = true;
                else h->edge_img[k]->pending.insert(h->edge_img[k]->pending.end(), delta.slots.begin(), delta.slots.end());
                h->filt_img[k]->pending.insert(h->filt_img[k]->pending.end(), delta.fids.begin(), delta.fids.end());
            }
            const int tgt = h->cur_img ^ 1;
            // an in-flight pass may still read the target image through an older epoch: give it a fresh one
            if (h->edge_img[tgt].use_count() > 1) h->edge_img[tgt] = std::make_shared<EdgeImage>();
            if (h->filt_img[tgt].use_count() > 1) h->filt_img[tgt] = std::make_shared<FiltImage>();
            EdgeImage& ei = *h->edge_img[tgt];
            bool full = false;
            if (ei.need_full || ei.cap != edges.size() || ei.pending.size() > edges.size() / 8) {
                ei.buf.ensure(edges.size() * sizeof(EdgeEntry));
                RGR_HIP(hipMemcpy(ei.buf.p, edges.data(), edges.size() * sizeof(EdgeEntry), hipMemcpyHostToDevice));
                ei.cap = edges.size();
                full = true;
            } else if (!ei.pending.empty()) {
                std::sort(ei.pending.begin(), ei.pending.end());
                ei.pending.erase(std::unique(ei.pending.begin(), ei.pending.end()), ei.pending.end());
                std::vector<EdgeEntry> recs(ei.pending.size());
                for (size_t i = 0; i < recs.size(); ++i) recs[i] = edges[ei.pending[i]];
                DevBuf d_slots, d_recs;
                d_slots.ensure(recs.size() * 4); d_recs.ensure(recs.size() * sizeof(EdgeEntry));
                RGR_HIP(hipMemcpy(d_slots.p, ei.pending.data(), recs.size() * 4, hipMemcpyHostToDevice));
                RGR_HIP(hipMemcpy(d_recs.p, recs.data(), recs.size() * sizeof(EdgeEntry), hipMemcpyHostToDevice));
                launch_scatter_edges(ei.buf.as<EdgeEntry>(), d_slots.as<uint32_t>(), d_recs.as<EdgeEntry>(), uint32_t(recs.size()), nullptr);
                RGR_HIP(hipDeviceSynchronize());
            }
            ei.pending.clear(); ei.need_full = false;
            lap("dict + edges");
            // ---- subscriber runs: dirty filters get a fresh run appended to the pool
            const uint64_t nf = h->table.filter_capacity();
            if (h->host_desc.size() < nf) h->host_desc.resize(nf, FilterDesc{0, 0});
            std::vector<uint32_t> fids = delta.fids;
            std::sort(fids.begin(), fids.end());
            fids.erase(std::unique(fids.begin(), fids.end()), fids.end());
            uint64_t add = 0;
            for (uint32_t f : fids) { const auto* v = h->table.filter_subs(f); add += v ? v->size() : 0; }
            const bool want_attrs = h->table.has_attrs();
            const bool attrs_dirty = h->table.take_attrs_all_dirty();     // (same single-reader rule as the delta)
            auto gather_attrs = [&](const std::vector<SubEntry>& run) {
                std::vector<SubAttr> at(run.size());
                for (size_t i = 0; i < run.size(); ++i) at[i] = h->table.sub_attr(run[i].sub_id);
                return at;
            };
            bool rebuild = !h->sub_pool || h->sub_pool->used + add > h->sub_pool->cap || h->sub_pool->used + add > 0xFFFFFF00ull ||
                           h->pool_garbage > h->table.n_subs() + (1u << 16) || attrs_dirty || (want_attrs && !h->sub_pool->has_attrs);
            if (rebuild) {
                std::vector<FilterDesc> filt;
                std::vector<SubEntry> subs;
                h->table.flatten_filters(filt, subs);
                auto np = std::make_shared<SubsPool>();
                np->cap = subs.size() + subs.size() / 2 + (1u << 16);      // head-room for appended runs
                np->buf.ensure(np->cap * sizeof(SubEntry));
                if (!subs.empty()) RGR_HIP(hipMemcpy(np->buf.p, subs.data(), subs.size() * sizeof(SubEntry), hipMemcpyHostToDevice));
                np->used = subs.size();
                np->packed_buf.ensure((np->cap + kPackedPad) * 4);     // (+ padding: expand_compact_lp_kernel reads four entries at a run's start without asking how long the run is)
                launch_pack_subs(np->buf.as<SubEntry>(), np->used, np->packed_buf.as<uint32_t>(), nullptr);
                if (want_attrs) {
                    np->attr_buf.ensure(np->cap * sizeof(SubAttr));
                    const auto at = gather_attrs(subs);
                    if (!at.empty()) RGR_HIP(hipMemcpy(np->attr_buf.p, at.data(), at.size() * sizeof(SubAttr), hipMemcpyHostToDevice));
                    np->has_attrs = true;
                }
                h->sub_pool = np;
                h->host_desc = filt;
                if (h->host_desc.size() < nf) h->host_desc.resize(nf, FilterDesc{0, 0});
                h->pool_garbage = 0;
                for (int k = 0; k < 2; ++k) h->filt_img[k]->need_full = true;
            } else if (!fids.empty()) {
                std::vector<SubEntry> stage;
                stage.reserve(add);
                for (uint32_t f : fids) {
                    const auto* v = h->table.filter_subs(f);
                    h->pool_garbage += h->host_desc[f].count;
                    h->host_desc[f] = FilterDesc{uint32_t(h->sub_pool->used + stage.size()), v ? uint32_t(v->size()) : 0u};
                    if (v) stage.insert(stage.end(), v->begin(), v->end());
                }
                if (!stage.empty())
                {
                    RGR_HIP(hipMemcpy(h->sub_pool->buf.as<SubEntry>() + h->sub_pool->used, stage.data(), stage.size() * sizeof(SubEntry), hipMemcpyHostToDevice));
                    launch_pack_subs(h->sub_pool->buf.as<SubEntry>() + h->sub_pool->used, stage.size(), h->sub_pool->packed_buf.as<uint32_t>() + h->sub_pool->used, nullptr);
                }
                if (!stage.empty() && h->sub_pool->has_attrs) {
                    const auto at = gather_attrs(stage);
                    RGR_HIP(hipMemcpy(h->sub_pool->attr_buf.as<SubAttr>() + h->sub_pool->used, at.data(), at.size() * sizeof(SubAttr), hipMemcpyHostToDevice));
                }
                h->sub_pool->used += stage.size();
            }
            lap("subscriber runs");
            FiltImage& fi = *h->filt_img[tgt];
            if (fi.need_full || fi.cap < nf || fi.pending.size() > nf / 4) {
                fi.cap = nf + nf / 4 + 1024;
                fi.buf.ensure(fi.cap * sizeof(FilterDesc));
                if (nf) RGR_HIP(hipMemcpy(fi.buf.p, h->host_desc.data(), nf * sizeof(FilterDesc), hipMemcpyHostToDevice));
                full = true;
            } else if (!fi.pending.empty()) {
                std::sort(fi.pending.begin(), fi.pending.end());
                fi.pending.erase(std::unique(fi.pending.begin(), fi.pending.end()), fi.pending.end());
                std::vector<FilterDesc> recs(fi.pending.size());
                for (size_t i = 0; i < recs.size(); ++i) recs[i] = h->host_desc[fi.pending[i]];
                DevBuf d_f, d_r;
                d_f.ensure(recs.size() * 4); d_r.ensure(recs.size() * sizeof(FilterDesc));
                RGR_HIP(hipMemcpy(d_f.p, fi.pending.data(), recs.size() * 4, hipMemcpyHostToDevice));
                RGR_HIP(hipMemcpy(d_r.p, recs.data(), recs.size() * sizeof(FilterDesc), hipMemcpyHostToDevice));
                launch_scatter_desc(fi.buf.as<FilterDesc>(), d_f.as<uint32_t>(), d_r.as<FilterDesc>(), uint32_t(recs.size()), nullptr);
                RGR_HIP(hipDeviceSynchronize());
            }
            fi.pending.clear(); fi.need_full = false;
            lap("filter descs");
            (full ? h->commits_full : h->commits_delta)++;
            h->cur_img = tgt;
            ep->edges = h->edge_img[tgt];
            ep->filt = h->filt_img[tgt];
            ep->subs = h->sub_pool;
            ep->view.edges = ei.buf.as<EdgeEntry>();
            ep->view.mask = uint32_t(edges.size() - 1);
            ep->view.root = h->table.root_header();
            ep->view.filt = fi.buf.as<FilterDesc>();
            ep->view.subs = h->sub_pool->buf.as<SubEntry>();
            ep->view.attrs = h->sub_pool->has_attrs ? h->sub_pool->attr_buf.as<SubAttr>() : nullptr;
            ep->n_v5 = h->table.n_v5_subs();
            ep->max_sub_id = h->table.max_sub_id();
            ep->view.subs_packed = ep->max_sub_id < (1u << 30) ? h->sub_pool->packed_buf.as<uint32_t>() : nullptr;
            RGR_HIP(hipDeviceSynchronize());            // (the pack kernels above ran on the null stream; the epoch is published below)
            ep->max_node_idx = h->table.max_node_idx();
            ep->n_filters = h->table.n_filters();
            ep->n_subs = h->table.n_subs();
            ep->n_nodes = h->table.n_nodes();
            ep->edge_slots = edges.size();
            ep->bytes = ei.buf.bytes + fi.buf.bytes + h->sub_pool->buf.bytes + h->sub_pool->attr_buf.bytes + h->sub_pool->packed_buf.bytes;
        }
        recover.armed = false;
        std::lock_guard<std::mutex> g(h->epoch_mu);
        ep->id = ++h->epoch_counter;
        h->epoch = ep;
        return RGR_OK;
    });
}

// ------------------------------------------------------------------ device-resident batches
// `recycle`: take the workspace from the handle's pool (one-shot entry points); it goes back with
// batch_release().  Public rgr_batch_create always builds a fresh one owned by the caller.
static int32_t batch_create_impl(rgr_handle* h, const uint8_t* blob, const uint64_t* offs, uint32_t n, bool retain, rgr_batch** out,
                                 bool recycle = false, uint32_t tier = 0) {
    return guarded([&]() -> int32_t {
        if (!h || !out || (n && (!blob || !offs))) return fail(RGR_EINVAL, "rgr_batch_create: bad argument");
        RGR_HIP(hipSetDevice(h->cfg.device));
        std::unique_ptr<rgr_batch> b;
        if (recycle) {
            std::lock_guard<std::mutex> g(h->pool_mu);
            if (!h->pool.empty()) { b.reset(h->pool.back()); h->pool.pop_back(); }
        }
        if (!b) {
            b = std::make_unique<rgr_batch>();
            RGR_HIP(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
        }
        b->h = h;
        b->n = n;
        b->retain = retain;
        b->tier = retain ? tier : 0;
        b->deliver = false;
        b->ordered = false;
        b->retain_positions = false;
        b->format = kFmtTuple;
        b->has_topic_ids = false;
        b->from_publish = false;
        b->want_host_offsets = false;
        for (ChunkSlot& cs : b->cs) { if (cs.inflight) { RGR_HIP(hipEventSynchronize(cs.done)); cs.inflight = false; } cs.ready = false; }
        b->c = &b->cs[0];
        b->in_pass = false; b->cursor = 0; b->hits_before = 0;
        b->dict_tokens = ~0ull;
        b->epoch.reset(); b->repoch.reset();
        // two-tier retain batches are always tokenised on the device, against the dictionary image of the epoch
        // they run on: no retain_mu on the query path (lock order, retain_abi.inc) and no host dictionary that a
        // merge may replace under a reusable batch
        b->host_tok = h->cfg.host_tokenize && !(retain && h->retain_tiered());
        if (b->host_tok) {
            if (n) { b->h_blob.assign(blob + offs[0], blob + offs[n]); b->h_offs.assign(offs, offs + n + 1); for (auto& o : b->h_offs) o -= offs[0]; }
            else b->h_offs.assign(1, 0);
            tokenize_batch(h, b.get(), b->h_blob.data(), b->h_offs.data(), n);
        } else {
            const double t1 = now_ms();
            const uint64_t nbytes = n ? offs[n] - offs[0] : 0;
            std::vector<uint64_t> rel(size_t(n) + 1, 0);
            // the framing comes from the network side: offsets must not go backwards (publish_scan computes offs[i+1] - offs[i] and reads
        // that many bytes of the uploaded blob)
        for (uint32_t i = 0; i < n; ++i)
            if (offs[i] > offs[i + 1]) return fail(RGR_EINVAL, "rgr_batch_create_from_publish: packet_offsets are not monotonic");
        for (uint32_t i = 0; i <= n && n; ++i) rel[i] = offs[i] - offs[0];
            b->d_blob.ensure(std::max<uint64_t>(16, nbytes + 16));
            b->blob_bytes = nbytes;
            b->d_offs.ensure((size_t(n) + 1) * 8);
            if (nbytes) RGR_HIP(hipMemcpyAsync(b->d_blob.p, blob + offs[0], nbytes, hipMemcpyHostToDevice, b->stream));
            RGR_HIP(hipMemcpyAsync(b->d_offs.p, rel.data(), (size_t(n) + 1) * 8, hipMemcpyHostToDevice, b->stream));
            // (no synchronisation here since r7: `rel` and the caller's blob outlive tokenize_batch_device below, which synchronises the stream before
            // it returns; a micro-batch pass is made of such round trips.  h2d_ms is what the copies cost the HOST; their device time is in tokenize_ms.)
            b->local.h2d_ms += now_ms() - t1;
            if (retain) {
                std::shared_ptr<RetainEpoch> ep;
                { std::lock_guard<std::mutex> g(h->epoch_mu); ep = tier ? h->retain_delta_epoch : h->retain_epoch; }
                if (!ep && tier) return fail(RGR_ENOENT, "rgr_retain_batch_create_tier: the delta tier is empty");
                if (!ep) return fail(RGR_ESTATE, "rgr_retain_batch_create: rgr_retain_commit has not been called");
                tokenize_batch_device(b.get(), ep->dict);
            } else {
                tokenize_batch_device(b.get(), *current_epoch(h)->dict);
            }
        }
        *out = b.release();
        return RGR_OK;
    });
}

int32_t rgr_batch_create(rgr_handle* h, const uint8_t* blob, const uint64_t* offs, uint32_t n, rgr_batch** out) {
    return batch_create_impl(h, blob, offs, n, false, out);
}

// PUBLISH-packet form: the codec's topic extraction and the tokeniser both run on the device.
int32_t rgr_batch_create_from_publish(rgr_handle* h, const uint8_t* packets, const uint64_t* offs, uint32_t n, uint32_t version,
                                      const uint32_t* from_ids, rgr_batch** out) {
    return guarded([&]() -> int32_t {
        if (!h || !out || (n && (!packets || !offs))) return fail(RGR_EINVAL, "rgr_batch_create_from_publish: bad argument");
        if (version != 3 && version != 4 && version != 5) return fail(RGR_EINVAL, "rgr_batch_create_from_publish: version must be 3, 4 (MQTT 3.1 / 3.1.1) or 5");
        RGR_HIP(hipSetDevice(h->cfg.device));
        auto b = std::make_unique<rgr_batch>();
        RGR_HIP(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
        b->h = h; b->n = n; b->from_publish = true;
        const double t1 = now_ms();
        const uint64_t nbytes = n ? offs[n] - offs[0] : 0;
        std::vector<uint64_t> rel(size_t(n) + 1, 0);
        // the framing comes from the network side: offsets must not go backwards (publish_scan computes offs[i+1] - offs[i] and reads
        // that many bytes of the uploaded blob)
        for (uint32_t i = 0; i < n; ++i)
            if (offs[i] > offs[i + 1]) return fail(RGR_EINVAL, "rgr_batch_create_from_publish: packet_offsets are not monotonic");
        for (uint32_t i = 0; i <= n && n; ++i) rel[i] = offs[i] - offs[0];
        b->d_pkts.ensure(std::max<uint64_t>(16, nbytes + 16));
        b->d_pkt_offs.ensure((size_t(n) + 1) * 8);
        b->d_pubinfo.ensure(std::max<size_t>(1, n) * sizeof(PubInfo));
        b->d_level_cnt.ensure(std::max<uint32_t>(1, n) * 4);
        b->d_force.ensure(std::max<uint32_t>(1, n));
        b->d_offs.ensure((size_t(n) + 1) * 8);
        b->scan_tmp.ensure((size_t(n) / scan_block_topics() + 3) * 16);
        DevBuf d_from;
        if (nbytes) RGR_HIP(hipMemcpyAsync(b->d_pkts.p, packets + offs[0], nbytes, hipMemcpyHostToDevice, b->stream));
        RGR_HIP(hipMemcpyAsync(b->d_pkt_offs.p, rel.data(), (size_t(n) + 1) * 8, hipMemcpyHostToDevice, b->stream));
        if (from_ids) {
            d_from.ensure(std::max<size_t>(1, n) * 4);
            b->d_pub.ensure(std::max<size_t>(1, n) * sizeof(PublishAttr));
            if (n) RGR_HIP(hipMemcpyAsync(d_from.p, from_ids, size_t(n) * 4, hipMemcpyHostToDevice, b->stream));
        }
        b->local.h2d_ms += now_ms() - t1;
        const double t2 = now_ms();
        // scan every packet; the topic lengths are exclusive-scanned into the offsets of a dense topic blob
        launch_publish_scan(b->d_pkts.as<uint8_t>(), b->d_pkt_offs.as<uint64_t>(), n, int(version), b->d_pubinfo.as<PubInfo>(),
                            b->d_level_cnt.as<uint32_t>(), b->d_force.as<uint8_t>(), from_ids ? d_from.as<uint32_t>() : nullptr,
                            from_ids ? b->d_pub.as<PublishAttr>() : nullptr, b->stream);
        launch_scan_u32(b->d_level_cnt.as<uint32_t>(), b->d_offs.as<uint64_t>(), n, b->scan_tmp.as<uint64_t>(), b->stream);
        uint64_t topic_bytes = 0;
        b->pub_info.resize(n);
        RGR_HIP(hipMemcpyAsync(&topic_bytes, b->d_offs.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, b->stream));
        if (n) RGR_HIP(hipMemcpyAsync(b->pub_info.data(), b->d_pubinfo.p, size_t(n) * sizeof(PubInfo), hipMemcpyDeviceToHost, b->stream));
        RGR_HIP(hipStreamSynchronize(b->stream));
        b->d_blob.ensure(std::max<uint64_t>(16, topic_bytes + 16));
        b->blob_bytes = topic_bytes;
        launch_publish_topics(b->d_pkts.as<uint8_t>(), b->d_pubinfo.as<PubInfo>(), n, b->d_offs.as<uint64_t>(), b->d_blob.as<uint8_t>(), b->stream);
        RGR_HIP(hipGetLastError());
        b->local.tokenize_ms += now_ms() - t2;
        tokenize_batch_device(b.get(), *current_epoch(h)->dict);
        b->deliver = from_ids != nullptr;
        *out = b.release();
        return RGR_OK;
    });
}

const rgr_publish_info* rgr_batch_publish_info(const rgr_batch* b) {
    static_assert(sizeof(rgr_publish_info) == sizeof(PubInfo), "rgr_publish_info layout");
    return b && b->from_publish ? reinterpret_cast<const rgr_publish_info*>(b->pub_info.data()) : nullptr;
}

// Return a recycled workspace to the pool (bounded; extras are destroyed).
static void batch_release(rgr_batch* b) {
    if (!b) return;
    rgr_handle* h = b->h;
    (void)hipSetDevice(h->cfg.device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    if (b->prep_stream) (void)hipStreamSynchronize(b->prep_stream);
    for (ChunkSlot& cs : b->cs) cs.inflight = false;
    try { b->resolve_spans(); } catch (...) {
        for (auto& sp : b->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
        b->spans.clear();
    }
    merge_stats(h, b->local);
    b->epoch.reset(); b->repoch.reset();
    {
        std::lock_guard<std::mutex> g(h->pool_mu);
        if (h->pool.size() < 16) { h->pool.push_back(b); return; }
    }
    delete b;
}

void rgr_batch_destroy(rgr_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->h->cfg.device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    if (b->prep_stream) (void)hipStreamSynchronize(b->prep_stream);
    merge_stats(b->h, b->local);
    delete b;
}

const int32_t* rgr_batch_status(const rgr_batch* b) { return b ? b->status.data() : nullptr; }

int32_t rgr_batch_set_publish_attrs(rgr_batch* b, const rgr_publish_attr* attrs) {
    return guarded([&]() -> int32_t {
        if (!b) return fail(RGR_EINVAL, "rgr_batch_set_publish_attrs: bad argument");
        if (b->in_pass) return fail(RGR_ESTATE, "rgr_batch_set_publish_attrs: inside a pass");
        if (b->retain) return fail(RGR_ESTATE, "rgr_batch_set_publish_attrs: not a publish batch");
        if (!attrs) { b->deliver = false; if (b->format == kFmtDeliver8) b->format = kFmtTuple; return RGR_OK; }
        if (b->format != kFmtTuple && b->format != kFmtDeliver8) return fail(RGR_ESTATE, "rgr_batch_set_publish_attrs: the delivery stage needs RGR_FORMAT_TUPLE or RGR_FORMAT_DELIVER8");
        if (b->has_topic_ids) return fail(RGR_ESTATE, "rgr_batch_set_publish_attrs: not together with rgr_batch_set_topic_ids");
        RGR_HIP(hipSetDevice(b->h->cfg.device));
        static_assert(sizeof(rgr_publish_attr) == sizeof(PublishAttr), "rgr_publish_attr layout");
        b->d_pub.ensure(std::max<size_t>(1, b->n) * sizeof(PublishAttr));
        b->d_pub_in.ensure(std::max<size_t>(1, b->n) * sizeof(PublishAttr));
        if (b->n) {
            // (r7) a small batch stages the caller's array in pinned memory of its own and does not wait for the upload: the pass's kernels are ordered
            // behind it on the stream, and a micro-batch pass is made of host round trips
            const bool staged = size_t(b->n) * sizeof(PublishAttr) <= (1u << 20);
            const void* src = attrs;
            if (staged) {
                if (b->h_pub_stage.p) RGR_HIP(hipStreamSynchronize(b->stream));          // (an earlier upload from this buffer may still be reading it)
                b->h_pub_stage.ensure(size_t(b->n) * sizeof(PublishAttr));
                std::memcpy(b->h_pub_stage.p, attrs, size_t(b->n) * sizeof(PublishAttr));
                src = b->h_pub_stage.p;
            }
            RGR_HIP(hipMemcpyAsync(b->d_pub_in.p, src, size_t(b->n) * sizeof(PublishAttr), hipMemcpyHostToDevice, b->stream));
            // a batch in walk order (rgr_batch_set_order): the delivery stage indexes its attributes by walk position
            if (b->ordered) launch_order_gather_attrs(b->d_order.as<uint32_t>(), b->d_pub_in.as<PublishAttr>(), b->n, b->d_pub.as<PublishAttr>(), b->stream);
            else RGR_HIP(hipMemcpyAsync(b->d_pub.p, b->d_pub_in.p, size_t(b->n) * sizeof(PublishAttr), hipMemcpyDeviceToDevice, b->stream));
            if (!staged) RGR_HIP(hipStreamSynchronize(b->stream));
        }
        b->deliver = true;
        return RGR_OK;
    });
}

int32_t rgr_batch_set_order(rgr_batch* b, uint32_t order) {
    return guarded([&]() -> int32_t {
        if (!b || order > RGR_ORDER_WALK) return fail(RGR_EINVAL, "rgr_batch_set_order: bad argument");
        if (b->in_pass) return fail(RGR_ESTATE, "rgr_batch_set_order: inside a pass");
        RGR_HIP(hipSetDevice(b->h->cfg.device));
        if (order == RGR_ORDER_CALLER) {
            if (b->ordered && b->deliver && b->d_pub_in.p && b->n) {        // the kernels index the attributes by batch position again
                RGR_HIP(hipMemcpyAsync(b->d_pub.p, b->d_pub_in.p, size_t(b->n) * sizeof(PublishAttr), hipMemcpyDeviceToDevice, b->stream));
                RGR_HIP(hipStreamSynchronize(b->stream));
            }
            b->ordered = false;
            return RGR_OK;
        }
        if (b->retain) return fail(RGR_ESTATE, "rgr_batch_set_order: not a publish batch");
        if (b->deliver && b->from_publish) return fail(RGR_ESTATE, "rgr_batch_set_order: not on a PUBLISH-packet batch that carries its own attributes");
        b->ordered = true;
        struct Undo { rgr_batch* b; bool armed = true; ~Undo() { if (armed) b->ordered = false; } } undo{b};
        apply_order(b);
        undo.armed = false;
        return RGR_OK;
    });
}

const uint32_t* rgr_batch_topic_order(const rgr_batch* b) { return (b && b->ordered) ? b->h_order.data() : nullptr; }

int32_t rgr_batch_set_retain_positions(rgr_batch* b, int32_t on) {
    if (!b) return fail(RGR_EINVAL, "rgr_batch_set_retain_positions: bad argument");
    if (b->in_pass) return fail(RGR_ESTATE, "rgr_batch_set_retain_positions: inside a pass");
    if (!b->retain) return fail(RGR_ESTATE, "rgr_batch_set_retain_positions: not a retain batch");
    b->retain_positions = on != 0;
    return RGR_OK;
}

int32_t rgr_batch_retain_vals(const rgr_batch* b, const rgr_retain_val** vals, uint64_t* n) {
    if (!b || !vals || !n) return fail(RGR_EINVAL, "rgr_batch_retain_vals: bad argument");
    if (!b->retain || !b->repoch) return fail(RGR_ESTATE, "rgr_batch_retain_vals: no retain pass has begun on this batch");
    static_assert(sizeof(rgr_retain_val) == sizeof(SubEntry), "rgr_retain_val layout");
    std::shared_ptr<const std::vector<SubEntry>> hv;
    { std::lock_guard<std::mutex> g(b->h->epoch_mu); hv = b->repoch->h_vals; }
    const_cast<rgr_batch*>(b)->retain_vals_pin = hv;        // (the two-tier commit replaces the epoch's mirror copy-on-write: keep the one handed out)
    *vals = hv ? reinterpret_cast<const rgr_retain_val*>(hv->data()) : nullptr;
    *n = hv ? hv->size() : 0;
    return RGR_OK;
}

int32_t rgr_batch_set_topic_ids(rgr_batch* b, const uint32_t* ids) {
    return guarded([&]() -> int32_t {
        if (!b) return fail(RGR_EINVAL, "rgr_batch_set_topic_ids: bad argument");
        if (b->in_pass) return fail(RGR_ESTATE, "rgr_batch_set_topic_ids: inside a pass");
        if (!ids) { b->has_topic_ids = false; return RGR_OK; }
        if (b->deliver) return fail(RGR_ESTATE, "rgr_batch_set_topic_ids: not together with publish attributes");
        RGR_HIP(hipSetDevice(b->h->cfg.device));
        b->d_topic_ids.ensure(std::max<size_t>(1, b->n) * 4);
        if (b->n) RGR_HIP(hipMemcpyAsync(b->d_topic_ids.p, ids, size_t(b->n) * 4, hipMemcpyHostToDevice, b->stream));
        b->has_topic_ids = true;
        if (b->ordered && b->n) {
            b->d_order_ids.ensure(size_t(b->n) * 4);
            launch_order_compose(b->d_order.as<uint32_t>(), b->d_topic_ids.as<uint32_t>(), b->n, b->d_order_ids.as<uint32_t>(), b->stream);
        }
        RGR_HIP(hipStreamSynchronize(b->stream));
        return RGR_OK;
    });
}

int32_t rgr_batch_set_format(rgr_batch* b, uint32_t format) {
    if (!b || format > RGR_FORMAT_DELIVER8) return fail(RGR_EINVAL, "rgr_batch_set_format: bad argument");
    if (b->in_pass) return fail(RGR_ESTATE, "rgr_batch_set_format: inside a pass");
    if (format != RGR_FORMAT_TUPLE && format != RGR_FORMAT_DELIVER8 && b->deliver) return fail(RGR_ESTATE, "rgr_batch_set_format: the delivery stage needs RGR_FORMAT_TUPLE or RGR_FORMAT_DELIVER8");
    if (format == RGR_FORMAT_DELIVER8 && (!b->deliver || b->retain)) return fail(RGR_ESTATE, "rgr_batch_set_format: RGR_FORMAT_DELIVER8 is the delivery stage's format (attach publish attributes first)");
    static_assert(RGR_FORMAT_TUPLE == kFmtTuple && RGR_FORMAT_SOA == kFmtSoa && RGR_FORMAT_PACKED == kFmtPacked && RGR_FORMAT_RUNS == kFmtRuns && RGR_FORMAT_IDS24 == kFmtIds24 &&
                  RGR_FORMAT_DELIVER8 == kFmtDeliver8, "format constants");
    static_assert(sizeof(rgr_hit8) == 8, "rgr_hit8 layout");
    b->format = int(format);
    return RGR_OK;
}

int32_t rgr_batch_begin(rgr_batch* b) {
    return guarded([&]() -> int32_t {
        if (!b) return fail(RGR_EINVAL, "rgr_batch_begin: bad argument");
        RGR_HIP(hipSetDevice(b->h->cfg.device));
        if (b->retain) {
            std::lock_guard<std::mutex> g(b->h->epoch_mu);
            b->repoch = b->tier ? b->h->retain_delta_epoch : b->h->retain_epoch;
            if (!b->repoch && b->tier) return fail(RGR_ENOENT, "rgr_batch_begin: the delta tier is empty");
            if (!b->repoch) return fail(RGR_ESTATE, "rgr_batch_begin: rgr_retain_commit has not been called");
        } else {
            b->epoch = current_epoch(b->h);
        }
        const uint64_t want = b->retain ? b->repoch->dict.n_tokens : b->epoch->dict->n_tokens;
        if (want != b->dict_tokens) {      // the dictionary grew since this batch was tokenised
            if (b->host_tok) tokenize_batch(b->h, b, b->h_blob.data(), b->h_offs.data(), b->n);
            else tokenize_batch_device(b, b->retain ? b->repoch->dict : *b->epoch->dict);
            if (b->ordered) apply_order(b);                  // (new token ids: a new order)
        }
        if (b->deliver && b->ordered && b->format != kFmtDeliver8)
            return fail(RGR_ESTATE, "rgr_batch_begin: a delivery pass in walk order answers in RGR_FORMAT_DELIVER8 (the 12-byte tuple's topic column would name walk positions)");
        if (b->format == kFmtPacked && (b->retain ? b->repoch->max_id : b->epoch->max_sub_id) >= (1u << 30))
            return fail(RGR_ECAPACITY, "rgr_batch_begin: RGR_FORMAT_PACKED needs ids below 2^30");
        if (b->format == kFmtIds24 && (b->retain ? b->repoch->max_id : b->epoch->max_sub_id) >= (1u << 24))
            return fail(RGR_ECAPACITY, "rgr_batch_begin: RGR_FORMAT_IDS24 needs ids below 2^24");
        for (ChunkSlot& cs : b->cs) {          // a pass abandoned midway may have left a prefetch behind
            if (cs.inflight) { RGR_HIP(hipEventSynchronize(cs.done)); cs.inflight = false; }
            cs.ready = false;
        }
        b->c = &b->cs[0];
        if (b->dedup_scalars.p) RGR_HIP(hipMemsetAsync(b->dedup_scalars.p, 0, 16, b->stream));     // (a pass abandoned midway leaves its count behind)
        if (b->dedup_stat.p) RGR_HIP(hipMemsetAsync(b->dedup_stat.p, 0, size_t(dedup_stat_slots()) * 8, b->stream));
        b->in_pass = true;
        b->cursor = 0;
        b->hits_before = 0;
        b->tile_plan.valid = false;
        b->win_seq = 0;
        b->dedup_seq = 0;
        b->local.topics += b->n;
        b->local.invalid_topics += b->n - b->valid_topics;
        b->local.levels += b->valid_levels;
        b->local.alg_bytes_walk += 4 * (b->valid_levels + b->valid_topics);
        return RGR_OK;
    });
}

int32_t rgr_batch_next_window(rgr_batch* b, rgr_window* w) {
    return guarded([&]() -> int32_t {
        if (!b || !w) return fail(RGR_EINVAL, "rgr_batch_next_window: bad argument");
        if (!b->in_pass) return fail(RGR_ESTATE, "rgr_batch_next_window: call rgr_batch_begin first");
        rgr_handle* h = b->h;
        RGR_HIP(hipSetDevice(h->cfg.device));
        if (b->cursor >= b->n) {
            unsigned long long n_cand = 0;
            std::vector<unsigned long long> per_block;
            if (b->dedup_stat.p) {                    // candidates the pass's dedup kernels saw (accumulated on the device, one slot per block of the tile pass)
                per_block.assign(dedup_stat_slots(), 0);
                RGR_HIP(hipMemcpyAsync(per_block.data(), b->dedup_stat.p, per_block.size() * 8, hipMemcpyDeviceToHost, b->stream));
                RGR_HIP(hipMemsetAsync(b->dedup_stat.p, 0, per_block.size() * 8, b->stream));
            }
            RGR_HIP(hipStreamSynchronize(b->stream));
            for (unsigned long long v : per_block) n_cand += v;
            b->local.dedup_candidates += n_cand;
            RGR_HIP(hipGetLastError());
            b->resolve_spans();
            b->in_pass = false;
            merge_stats(h, b->local);
            return RGR_EOF;
        }
        if (!b->c->ready || b->cursor >= b->c->begin + b->c->n) enter_chunk(b, b->cursor);
        const uint32_t n = b->c->n;
        const uint32_t lc = b->cursor - b->c->begin;
        const uint64_t cap = b->window_cap();
        uint32_t le;
        uint64_t hit_lo, hit_hi, pair_lo, pair_hi;
        if (!b->c->host_arrays) {            // the whole chunk is one window (its totals fit): no per-topic arrays needed
            le = n; hit_lo = 0; hit_hi = b->c->total_hits; pair_lo = 0; pair_hi = b->c->total_pairs;
        } else {
            const uint64_t* ho = b->c->h_hit_off.as<uint64_t>();
            const uint64_t* pb = b->c->h_pair_base.as<uint64_t>();
            if (ho[n] - ho[lc] <= cap) le = n;
            else {
                le = uint32_t(std::upper_bound(ho + lc, ho + n + 1, ho[lc] + cap) - ho) - 1;
                if (le <= lc) le = lc + 1;       // a single topic larger than the window: give it its own window
            }
            hit_lo = ho[lc]; hit_hi = ho[le]; pair_lo = pb[lc]; pair_hi = pb[le];
        }
        const uint64_t nh = hit_hi - hit_lo;
        w->n_runs = 0; w->d_run_src = nullptr; w->d_run_topic = nullptr; w->d_run_off = nullptr; w->d_subs = nullptr;
        if (b->format == kFmtRuns) {
            // nothing to expand: the dense pair arrays of the chunk ARE the run list (built by compact_kernel)
            w->n_runs = pair_hi - pair_lo;
            w->d_run_src = b->c->pair_src.as<uint32_t>() + pair_lo;
            w->d_run_topic = b->c->pair_topic.as<uint32_t>() + pair_lo;
            w->d_run_off = b->c->pair_off.as<uint64_t>() + pair_lo;
            w->d_subs = reinterpret_cast<const uint64_t*>(batch_view(b).subs);
            b->local.alg_bytes_expand -= nh * 20;            // (the chunk's accounting charged 20 B per hit)
            b->local.alg_bytes_expand += (pair_hi - pair_lo) * 16;
        } else if (nh) {
            const uint32_t T = expand_tile_hits();
            DevBuf& outbuf = b->alt_out ? b->out2 : b->out;
            const uint64_t ids_bytes = (nh * 4 + 255) & ~uint64_t(255);           // compact formats: sub ids, then the qos bytes
            outbuf.ensure(b->format == kFmtTuple ? nh * sizeof(Tuple) : b->format == kFmtDeliver8 ? nh * 8 : b->format == kFmtIds24 ? nh * 3 + 16 : ids_bytes + (b->format == kFmtSoa ? nh + 16 : 0));
            ChunkArrays ca = make_chunk_arrays(b, n);
            size_t sp;
            const char* fused_env = std::getenv("RGR_TILES_FUSED");
            const bool fused = fused_env && fused_env[0] == '1' && !b->deliver && b->format == kFmtIds24;
            TileRec* tile_recs;
            NextTiles next_tiles{};
            rgr_batch::TilePlan next_plan;
            if (fused) {
                const int slot = int(b->win_seq & 1u);
                const rgr_batch::TilePlan& tp = b->tile_plan;
                if (!(tp.valid && tp.chunk == b->c && tp.lc == lc && tp.le == le && tp.hit_lo == hit_lo && tp.pair_lo == pair_lo && tp.pair_hi == pair_hi && tp.slot == slot)) {
                    b->tile_buf[slot].ensure(((nh + T - 1) / T) * sizeof(TileRec));         // (first window of a chunk, or a plan that does not fit)
                    sp = b->span_begin(kSpanScan);
                    launch_tiles(ca, pair_lo, pair_hi, hit_lo, b->tile_buf[slot].as<TileRec>(), b->stream);
                    b->span_end(sp);
                }
                b->tile_plan.valid = false;
                tile_recs = b->tile_buf[slot].as<TileRec>();
                if (b->c->host_arrays && le < n) {             // the next window of this chunk: planned now, its records ride on this window's expansion
                    const uint64_t* ho = b->c->h_hit_off.as<uint64_t>();
                    const uint64_t* pb = b->c->h_pair_base.as<uint64_t>();
                    uint32_t nle;
                    if (ho[n] - ho[le] <= cap) nle = n;
                    else {
                        nle = uint32_t(std::upper_bound(ho + le, ho + n + 1, ho[le] + cap) - ho) - 1;
                        if (nle <= le) nle = le + 1;
                    }
                    if (ho[nle] > ho[le] && pb[nle] > pb[le]) {
                        b->tile_buf[slot ^ 1].ensure(((ho[nle] - ho[le] + T - 1) / T) * sizeof(TileRec));
                        next_tiles = NextTiles{pb[le], pb[nle], ho[le], b->tile_buf[slot ^ 1].as<TileRec>()};
                        next_plan.valid = true; next_plan.chunk = b->c; next_plan.lc = le; next_plan.le = nle;
                        next_plan.hit_lo = ho[le]; next_plan.pair_lo = pb[le]; next_plan.pair_hi = pb[nle]; next_plan.slot = slot ^ 1;
                    }
                }
            } else {
                b->tile_first.ensure(((nh + T - 1) / T) * sizeof(TileRec));
                sp = b->span_begin(kSpanScan);
                launch_tiles(ca, pair_lo, pair_hi, hit_lo, b->tile_first.as<TileRec>(), b->stream);
                b->span_end(sp);
                tile_recs = b->tile_first.as<TileRec>();
            }
            // delivery stage: fused into the expansion; v5 hits additionally go through the per-client dedup
            DeliverArgs da{};
            const bool dedup = b->deliver && !b->retain && b->epoch->n_v5 > 0 && b->epoch->view.attrs != nullptr;
            if (b->deliver && !b->retain) {
                if (nh > 0xFFFFFFFFull) return fail(RGR_ECAPACITY, "delivery stage: window larger than 2^32 hits");
                da.pub = b->d_pub.as<PublishAttr>();
                da.attrs = b->epoch->view.attrs;
                if (dedup) {
                    // every tile owns a tile-sized slice of the candidate list (no global cursor): sized for the
                    // worst case, touched only where candidates exist
                    const uint32_t nt = le - lc;
                    const uint64_t ntl = (nh + T - 1) / T;
                    b->cand.ensure(ntl * T * sizeof(Cand));
                    b->cand_count.ensure(ntl * 4 * 3);                   // per-tile counts (every tile writes its own), then the tiles' topic ranges
                    b->dedup_items.ensure((size_t(nt) + nh / dedup_topic_cap() + 2) * sizeof(DedupItem));
                    if (!b->dedup_scalars.p) { b->dedup_scalars.ensure(16); RGR_HIP(hipMemsetAsync(b->dedup_scalars.p, 0, 16, b->stream)); }
                    if (!b->dedup_stat.p) { b->dedup_stat.ensure(size_t(dedup_stat_slots()) * 8); RGR_HIP(hipMemsetAsync(b->dedup_stat.p, 0, size_t(dedup_stat_slots()) * 8, b->stream)); }
                    da.cand = b->cand.as<Cand>();
                    da.tile_ncand = b->cand_count.as<uint32_t>();
                    da.tile_trange = b->cand_count.as<uint32_t>() + ntl;
                    da.topic_lo = b->c->begin + lc;
                }
            }
            sp = b->span_begin(kSpanExpand);
            if (b->format == kFmtTuple || b->format == kFmtDeliver8)
            {
                // where the retained path's tuples take their ids from (kernels.hpp launch_expand): positions need no read at all; a single-tier
                // epoch's entries carry no flags, so its packed 4-byte side array serves (RGR_RETAIN_PACKED_READS=0, read per window: the 8-byte
                // entries as until r5)
                int id_source = 0;
                if (b->retain && b->retain_positions) id_source = 2;
                else if (b->retain && !h->retain_tiered() && batch_view(b).subs_packed) { const char* e = std::getenv("RGR_RETAIN_PACKED_READS"); id_source = (e && e[0] == '0') ? 0 : 1; }
                launch_expand(batch_view(b), ca, pair_lo, pair_hi, hit_lo, hit_hi, tile_recs, outbuf.as<Tuple>(), b->stream,
                              (b->deliver && !b->retain) ? &da : nullptr, b->format == kFmtDeliver8, id_source);
            }
            else if (launch_expand_compact(batch_view(b), ca, pair_lo, pair_hi, hit_lo, hit_hi, tile_recs, b->format,
                                           outbuf.as<uint32_t>(), outbuf.as<uint8_t>() + ids_bytes, b->stream, next_tiles.out ? &next_tiles : nullptr))
                b->tile_plan = next_plan;                      // the expansion wrote the next window's records as well
            b->span_end(sp);
            b->win_seq++;
            // the chunk's accounting charged 20 B per hit (8 read + 12 written); the compact formats write 5 / 4
            if (b->format != kFmtTuple) b->local.alg_bytes_expand -= nh * (b->format == kFmtSoa ? 7 : b->format == kFmtIds24 ? 9 : b->format == kFmtDeliver8 ? 4 : 8);
            b->local.expand_launches++;
            if (dedup) {
                // LDS tables (tile-local, then one block per spanning topic); stream-ordered, no host synchronisation
                sp = b->span_begin(kSpanDedup);
                launch_dedup(b->cand.as<Cand>(), b->cand_count.as<uint32_t>(), b->cand_count.as<uint32_t>() + (nh + T - 1) / T, uint32_t((nh + T - 1) / T),
                             b->format == kFmtDeliver8 ? hit8_words(outbuf.p) : tuple_words(outbuf.as<Tuple>()), le - lc, b->c->hit_off.as<uint64_t>() + lc, hit_lo, b->dedup_items.as<DedupItem>(),
                             reinterpret_cast<uint32_t*>(b->dedup_scalars.as<unsigned long long>() + 1), b->dedup_seq & 1u, b->dedup_stat.as<unsigned long long>(), b->stream);
                b->dedup_seq++;            // (its own counter, advanced exactly where a launch consumed the parity: ADVICE r5)
                b->span_end(sp);
                b->local.dedup_launches++;
            }
        }
        void* grouped_out = nullptr;
        if (b->group_by_node && b->deliver && !b->retain && b->format == kFmtTuple) {
            const uint32_t nt = le - lc;
            const uint64_t* d_ho = b->c->hit_off.as<uint64_t>() + lc;
            Tuple* cur = (b->alt_out ? b->out2 : b->out).as<Tuple>();
            if (nh && b->epoch->max_node_idx > 0) {          // one radix-256 pass per byte of the largest node index
                DevBuf& tmp = b->node_tmp[b->alt_out ? 1 : 0];
                tmp.ensure(nh * sizeof(Tuple));
                Tuple* other = tmp.as<Tuple>();
                size_t sp = b->span_begin(kSpanExpand);
                for (uint32_t shift = 16; shift < 32 && (b->epoch->max_node_idx >> (shift - 16)) != 0; shift += 8) {
                    launch_node_partition(cur, other, d_ho, hit_lo, nt, shift, b->stream);
                    std::swap(cur, other);
                }
                b->span_end(sp);
            }
            grouped_out = cur;
            // directory: groups per topic -> scan -> (node, first position) per group
            b->grp_cnt.ensure(std::max<size_t>(1, nt) * 4);
            b->grp_off.ensure((size_t(nt) + 1) * 8);
            b->scan_tmp.ensure((size_t(nt) / scan_block_topics() + 3) * 16);
            launch_node_groups(cur, d_ho, hit_lo, nt, b->grp_cnt.as<uint32_t>(), nullptr, nullptr, nullptr, 0, b->stream);
            launch_scan_u32(b->grp_cnt.as<uint32_t>(), b->grp_off.as<uint64_t>(), nt, b->scan_tmp.as<uint64_t>(), b->stream);
            b->h_grp_off.assign(size_t(nt) + 1, 0);
            RGR_HIP(hipMemcpyAsync(b->h_grp_off.data(), b->grp_off.p, (size_t(nt) + 1) * 8, hipMemcpyDeviceToHost, b->stream));
            RGR_HIP(hipStreamSynchronize(b->stream));
            const uint64_t ng = b->h_grp_off[nt];
            b->h_grp_node.assign(ng, 0);
            b->h_grp_begin.assign(ng, 0);
            if (ng) {
                b->grp_node.ensure(ng * 4);
                b->grp_begin.ensure(ng * 8);
                launch_node_groups(cur, d_ho, hit_lo, nt, nullptr, b->grp_off.as<uint64_t>(), b->grp_node.as<uint32_t>(), b->grp_begin.as<uint64_t>(), 0, b->stream);
                RGR_HIP(hipMemcpyAsync(b->h_grp_node.data(), b->grp_node.p, ng * 4, hipMemcpyDeviceToHost, b->stream));
                RGR_HIP(hipMemcpyAsync(b->h_grp_begin.data(), b->grp_begin.p, ng * 8, hipMemcpyDeviceToHost, b->stream));
                RGR_HIP(hipStreamSynchronize(b->stream));
            }
        }
        w->topic_begin = b->cursor;
        w->topic_end = b->c->begin + le;
        w->n_hits = nh;
        w->hit_base = b->hits_before;
        {
            void* op = grouped_out ? grouped_out : (b->alt_out ? b->out2.p : b->out.p);
            const uint64_t ids_bytes = (nh * 4 + 255) & ~uint64_t(255);
            w->d_tuples = b->format == kFmtTuple ? reinterpret_cast<const rgr_tuple*>(op) : nullptr;
            w->d_sub_ids = (b->format == kFmtSoa || b->format == kFmtPacked) && nh ? static_cast<const uint32_t*>(op) : nullptr;
            w->d_qos = b->format == kFmtSoa && nh ? static_cast<const uint8_t*>(op) + ids_bytes : nullptr;
            w->d_ids24 = b->format == kFmtIds24 && nh ? static_cast<const uint8_t*>(op) : nullptr;
            w->d_hits8 = b->format == kFmtDeliver8 && nh ? static_cast<const rgr_hit8*>(op) : nullptr;
            w->d_topic_order = b->ordered ? b->d_order.as<uint32_t>() + b->cursor : nullptr;
        }
        w->d_hit_offsets = b->c->hit_off.as<uint64_t>() + lc;
        w->offsets_bias = hit_lo;
        b->hits_before += nh;
        b->cursor = b->c->begin + le;
        return RGR_OK;
    });
}

int32_t rgr_window_to_host(rgr_batch* b, const rgr_window* w, rgr_tuple* host_tuples, uint64_t* host_hit_offsets) {
    return guarded([&]() -> int32_t {
        if (!b || !w) return fail(RGR_EINVAL, "rgr_window_to_host: bad argument");
        if (host_tuples && b->format != kFmtTuple) return fail(RGR_ESTATE, "rgr_window_to_host: tuples exist only in RGR_FORMAT_TUPLE");
        RGR_HIP(hipSetDevice(b->h->cfg.device));
        const double t0 = now_ms();
        if (host_tuples && w->n_hits)
            RGR_HIP(hipMemcpyAsync(host_tuples, w->d_tuples, w->n_hits * sizeof(rgr_tuple), hipMemcpyDeviceToHost, b->stream));
        RGR_HIP(hipStreamSynchronize(b->stream));
        if (host_hit_offsets) {
            fetch_host_arrays(b);
            const uint64_t* ho = b->c->h_hit_off.as<uint64_t>() + (w->topic_begin - b->c->begin);
            const uint32_t m = w->topic_end - w->topic_begin;
            for (uint32_t i = 0; i <= m; ++i) host_hit_offsets[i] = ho[i] - w->offsets_bias;
        }
        b->local.d2h_ms += now_ms() - t0;
        return RGR_OK;
    });
}

int32_t rgr_batch_run(rgr_batch* b, uint64_t* n_hits, uint32_t* n_windows) {
    int32_t rc = rgr_batch_begin(b);
    if (rc != RGR_OK) return rc;
    uint64_t hits = 0;
    uint32_t nw = 0;
    for (;;) {
        rgr_window w;
        rc = rgr_batch_next_window(b, &w);
        if (rc == RGR_EOF) break;
        if (rc != RGR_OK) return rc;
        hits += w.n_hits;
        nw++;
    }
    if (n_hits) *n_hits = hits;
    if (n_windows) *n_windows = nw;
    return RGR_OK;
}

// One pass with every window copied to the host while the next one is being expanded: window i goes into
// device buffer i & 1, its D2H copy runs on the copy stream behind an event, and the expansion into a
// buffer waits for that buffer's previous copy.  dst(w) names the (pinned) destination of window w's tuples —
// it is called after rgr_batch_next_window returned w, so the caller may size its storage from w — and
// done(w, host_ptr) runs once the copy has landed.  The caller has called rgr_batch_begin.
extern "C++" {
template <class Dst, class Done>
static int32_t stream_windows(rgr_batch* b, Dst dst, Done done, uint64_t* n_hits, uint32_t* n_windows) {
    b->ensure_stream_state();
    struct HostOut { rgr_batch* b; ~HostOut() { b->host_out = false; } } host_out_guard{b};
    struct Pending { bool live = false; rgr_window w{}; rgr_tuple* host = nullptr; } pend[2];
    uint64_t hits = 0;
    uint32_t nw = 0;
    int32_t r = RGR_OK;
    const double t0 = now_ms();
    auto drain = [&](int k) {
        if (!pend[k].live) return;
        RGR_HIP(hipEventSynchronize(b->ev_copied[k]));
        pend[k].live = false;
        done(pend[k].w, pend[k].host);
    };
    struct Restore { rgr_batch* b; ~Restore() { b->alt_out = false; } } restore{b};
    for (int k = 0;; k ^= 1) {
        drain(k);                                   // slot k's device buffer is free again
        b->alt_out = k == 1;
        rgr_window w;
        r = rgr_batch_next_window(b, &w);
        if (r != RGR_OK) break;
        RGR_HIP(hipEventRecord(b->ev_expanded[k], b->stream));
        rgr_tuple* host = dst(w, k);                 // (also for an empty window: the caller keeps its offsets there)
        RGR_HIP(hipStreamWaitEvent(b->copy_stream, b->ev_expanded[k], 0));
        if (w.n_hits) RGR_HIP(hipMemcpyAsync(host, w.d_tuples, w.n_hits * sizeof(rgr_tuple), hipMemcpyDeviceToHost, b->copy_stream));
        RGR_HIP(hipEventRecord(b->ev_copied[k], b->copy_stream));
        pend[k].live = true; pend[k].w = w; pend[k].host = host;
        hits += w.n_hits;
        nw++;
    }
    if (r != RGR_EOF) { (void)hipStreamSynchronize(b->copy_stream); return r; }
    drain(0); drain(1);
    b->local.d2h_ms += now_ms() - t0;
    if (n_hits) *n_hits = hits;
    if (n_windows) *n_windows = nw;
    return RGR_OK;
}
}  // extern "C++"

int32_t rgr_batch_run_to_host(rgr_batch* b, rgr_window_consumer consume, void* user, uint64_t* n_hits, uint32_t* n_windows) {
    if (b && b->format != kFmtTuple) return fail(RGR_ESTATE, "rgr_batch_run_to_host: RGR_FORMAT_TUPLE only");
    if (b) b->host_out = true;           // (before the pass is planned: its windows are sized for pinned staging; stream_windows clears it)
    int32_t rc = rgr_batch_begin(b);
    if (rc != RGR_OK && b) b->host_out = false;
    if (rc != RGR_OK) return rc;
    return guarded([&]() -> int32_t {
        return stream_windows(
            b,
            [&](const rgr_window& w, int k) {
                b->h_ring[k].ensure(std::max<uint64_t>(1, w.n_hits) * sizeof(rgr_tuple));
                return b->h_ring[k].as<rgr_tuple>();
            },
            [&](const rgr_window& w, rgr_tuple* host) { if (consume) consume(user, w.topic_begin, w.topic_end, host, w.n_hits); },
            n_hits, n_windows);
    });
}

// ------------------------------------------------------------------ host in / host out
namespace {
// Tuple storage of a host result: one pinned block from the handle's pool (the D2H copies run at PCIe speed
// and hipHostMalloc is paid once per size class, not per call).
struct TupleStore {
    std::shared_ptr<PinnedPool> pool;
    rgr_tuple* p = nullptr;
    size_t n = 0, cap = 0, cap_bytes = 0;
    ~TupleStore() { if (p && pool) pool->give(p, cap_bytes); }
    void grow(size_t want) {       // keeps contents; no copy may be in flight into the old block
        if (want <= cap) return;
        const size_t ncap = std::max(want, cap * 2);
        size_t nbytes = 0;
        rgr_tuple* np = static_cast<rgr_tuple*>(pool->take(std::max<size_t>(1, ncap) * sizeof(rgr_tuple), &nbytes));
        if (n) std::memcpy(np, p, n * sizeof(rgr_tuple));
        if (p) pool->give(p, cap_bytes);
        p = np; cap = nbytes / sizeof(rgr_tuple); cap_bytes = nbytes;
    }
};
struct ResultOwner {
    std::vector<int32_t> status;
    std::vector<uint64_t> offsets;
    TupleStore tuples;
    std::vector<uint32_t> ids;
    std::vector<uint64_t> grp_off, grp_begin;     // rgr_node_groups
    std::vector<uint32_t> grp_node;
};
}  // namespace

static int32_t match_batch_impl(rgr_handle* h, const uint8_t* blob, const uint64_t* offs, uint32_t n, const rgr_publish_attr* attrs,
                                rgr_result* out, const uint32_t* topic_ids = nullptr, rgr_node_groups* groups = nullptr) {
    if (!out) return fail(RGR_EINVAL, "rgr_match_batch: out is NULL");
    std::memset(out, 0, sizeof *out);
    if (groups) std::memset(groups, 0, sizeof *groups);
    rgr_batch* b = nullptr;
    int32_t rc = batch_create_impl(h, blob, offs, n, false, &b, true);
    if (rc != RGR_OK) return rc;
    rc = guarded([&]() -> int32_t {
        auto own = std::make_unique<ResultOwner>();
        own->tuples.pool = h->pinned;
        own->status.assign(b->status.begin(), b->status.end());
        own->offsets.assign(size_t(n) + 1, 0);
        int32_t r = attrs ? rgr_batch_set_publish_attrs(b, attrs) : RGR_OK;
        if (r != RGR_OK) return r;
        if (topic_ids) { r = rgr_batch_set_topic_ids(b, topic_ids); if (r != RGR_OK) return r; }
        b->want_host_offsets = true;
        b->group_by_node = groups != nullptr && attrs != nullptr;
        struct ClearGrouping { rgr_batch* b; ~ClearGrouping() { b->group_by_node = false; } } clear_grouping{b};
        if (b->group_by_node) own->grp_off.assign(size_t(n) + 1, 0);
        b->host_out = true;
        r = rgr_batch_begin(b);
        if (r != RGR_OK) { b->host_out = false; return r; }
        // Windows are expanded and copied in a two-deep pipeline straight into the result block.  The block is
        // sized when a chunk's totals are known (one chunk = up to rgr_config.chunk_topics topics: the whole
        // batch for any realistic call); only a batch of several chunks can make it grow, and it grows with no
        // copy in flight (both slots drained first).
        uint32_t sized_chunk = ~0u;
        bool in_flight[2] = {false, false};
        r = stream_windows(
            b,
            [&](const rgr_window& w, int k) {
                if (b->c->begin != sized_chunk) {
                    sized_chunk = b->c->begin;
                    const uint64_t chunk_hits = b->c->total_hits;
                    const size_t need = own->tuples.n + chunk_hits;
                    if (need > own->tuples.cap) {
                        for (int j = 0; j < 2; ++j) if (in_flight[j]) { RGR_HIP(hipEventSynchronize(b->ev_copied[j])); }
                        own->tuples.grow(need);
                    }
                }
                const size_t base = own->tuples.n;
                own->tuples.n = base + w.n_hits;
                fetch_host_arrays(b);
                const uint64_t* ho = b->c->h_hit_off.as<uint64_t>() + (w.topic_begin - b->c->begin);
                for (uint32_t i = 0; i <= w.topic_end - w.topic_begin; ++i) own->offsets[w.topic_begin + i] = base + (ho[i] - w.offsets_bias);
                if (b->group_by_node) {                        // the window's node directory, rebased to the result block
                    const uint64_t g0 = own->grp_node.size();
                    for (uint32_t i = 0; i <= w.topic_end - w.topic_begin; ++i) own->grp_off[w.topic_begin + i] = g0 + b->h_grp_off[i];
                    own->grp_node.insert(own->grp_node.end(), b->h_grp_node.begin(), b->h_grp_node.end());
                    for (uint64_t gb : b->h_grp_begin) own->grp_begin.push_back(base + gb);
                }
                in_flight[k] = true;
                return own->tuples.p + base;
            },
            [&](const rgr_window&, rgr_tuple*) {}, nullptr, nullptr);
        if (r != RGR_OK) return r;
        out->n_topics = n;
        out->n_hits = own->tuples.n;
        out->status = own->status.data();
        out->hit_offsets = own->offsets.data();
        out->tuples = own->tuples.p;
        if (groups && attrs) {
            for (uint32_t i = 1; i <= n; ++i) if (own->grp_off[i] < own->grp_off[i - 1]) own->grp_off[i] = own->grp_off[i - 1];   // (topics after the last window)
            own->grp_begin.push_back(own->tuples.n);
            groups->n_groups = own->grp_node.size();
            groups->group_offsets = own->grp_off.data();
            groups->group_node = own->grp_node.data();
            groups->group_begin = own->grp_begin.data();
        }
        out->_owner = own.release();
        return RGR_OK;
    });
    batch_release(b);
    return rc;
}

int32_t rgr_match_batch(rgr_handle* h, const uint8_t* blob, const uint64_t* offs, uint32_t n, rgr_result* out) {
    return match_batch_impl(h, blob, offs, n, nullptr, out);
}

int32_t rgr_match_batch_deliver(rgr_handle* h, const uint8_t* blob, const uint64_t* offs, uint32_t n, const rgr_publish_attr* attrs,
                                rgr_result* out) {
    if (n && !attrs) return fail(RGR_EINVAL, "rgr_match_batch_deliver: attrs is NULL");
    return match_batch_impl(h, blob, offs, n, attrs, out);
}

int32_t rgr_match_batch_deliver_grouped(rgr_handle* h, const uint8_t* blob, const uint64_t* offs, uint32_t n, const rgr_publish_attr* attrs,
                                        rgr_result* out, rgr_node_groups* groups) {
    if ((n && !attrs) || !groups) return fail(RGR_EINVAL, "rgr_match_batch_deliver_grouped: bad argument");
    return match_batch_impl(h, blob, offs, n, attrs, out, nullptr, groups);
}

void rgr_result_free(rgr_result* r) {
    if (!r || !r->_owner) return;
    delete static_cast<ResultOwner*>(r->_owner);
    std::memset(r, 0, sizeof *r);
}

static int32_t match_filters_impl(rgr_handle* h, const uint8_t* blob, const uint64_t* offs, uint32_t n, rgr_filters_result* out, bool reps) {
    if (!out) return fail(RGR_EINVAL, "rgr_match_filters: out is NULL");
    std::memset(out, 0, sizeof *out);
    rgr_batch* b = nullptr;
    int32_t rc = batch_create_impl(h, blob, offs, n, false, &b, true);
    if (rc != RGR_OK) return rc;
    rc = guarded([&]() -> int32_t {
        auto own = std::make_unique<ResultOwner>();
        own->status.assign(b->status.begin(), b->status.end());
        own->offsets.assign(size_t(n) + 1, 0);
        int32_t r = rgr_batch_begin(b);
        if (r != RGR_OK) return r;
        // Per chunk: walk, exclusive scan of the per-topic matched-filter counts, one kernel that writes the
        // filter ids densely in iteration order, and two copies (offsets, ids) into pinned staging.  What
        // crosses PCIe is 4 bytes per matched filter + 8 per topic — not the slot arrays.
        for (uint32_t begin = 0; begin < n; begin += b->c->n) {
            prepare_chunk(b, begin, true);
            const uint32_t cn = b->c->n;
            const Scalars* hs = b->c->h_scalars.as<Scalars>();
            b->c->pair_base.ensure((size_t(cn) + 1) * 8);
            b->c->h_pair_base.ensure((size_t(cn) + 1) * 8);
            b->scan_tmp.ensure((size_t(cn) / scan_block_topics() + 3) * 16);
            uint64_t* d_off = b->c->pair_base.as<uint64_t>();
            launch_scan_u32(b->c->pair_cnt.as<uint32_t>(), d_off, cn, b->scan_tmp.as<uint64_t>(), b->stream);
            RGR_HIP(hipMemcpyAsync(b->c->h_pair_base.p, d_off, (size_t(cn) + 1) * 8, hipMemcpyDeviceToHost, b->stream));
            RGR_HIP(hipStreamSynchronize(b->stream));
            const uint64_t* ho = b->c->h_pair_base.as<uint64_t>();
            const uint64_t P = ho[cn];
            const size_t base = own->ids.size();
            if (P) {
                b->c->pair_src.ensure(P * 4);
                b->h_ring[0].ensure(P * 4);
                launch_pairs_dense(batch_view(b), make_chunk_arrays(b, cn), d_off, b->c->pair_src.as<uint32_t>(), reps, b->stream);
                RGR_HIP(hipMemcpyAsync(b->h_ring[0].p, b->c->pair_src.p, P * 4, hipMemcpyDeviceToHost, b->stream));
                RGR_HIP(hipStreamSynchronize(b->stream));
                RGR_HIP(hipGetLastError());
                own->ids.resize(base + P);
                std::memcpy(own->ids.data() + base, b->h_ring[0].p, P * 4);
            }
            for (uint32_t t = 0; t <= cn; ++t) own->offsets[begin + t] = base + ho[t];
            b->local.pairs += P;
            b->local.visited_nodes += hs->visited;
            b->local.overflow_topics += hs->ovf_count;
            b->local.alg_bytes_walk += 24 * hs->visited + 8 * P;
        }
        RGR_HIP(hipStreamSynchronize(b->stream));
        b->resolve_spans();
        b->in_pass = false;
        out->n_topics = n;
        out->n_pairs = own->ids.size();
        out->status = own->status.data();
        out->pair_offsets = own->offsets.data();
        out->filter_ids = own->ids.data();
        out->_owner = own.release();
        return RGR_OK;
    });
    batch_release(b);
    return rc;
}

int32_t rgr_match_filters(rgr_handle* h, const uint8_t* blob, const uint64_t* offs, uint32_t n, rgr_filters_result* out) {
    return match_filters_impl(h, blob, offs, n, out, false);
}
int32_t rgr_match_filter_subs(rgr_handle* h, const uint8_t* blob, const uint64_t* offs, uint32_t n, rgr_filters_result* out) {
    return match_filters_impl(h, blob, offs, n, out, true);
}

void rgr_filters_result_free(rgr_filters_result* r) {
    if (!r || !r->_owner) return;
    delete static_cast<ResultOwner*>(r->_owner);
    std::memset(r, 0, sizeof *r);
}

// ------------------------------------------------------------------ sharding rule
int32_t rgr_shard_assign(const uint8_t* blob, const uint64_t* offsets, uint64_t n, uint32_t n_shards, int32_t is_filter,
                         uint32_t key_levels, int32_t* out) {
    return guarded([&]() -> int32_t {
        if ((n && (!blob || !offsets || !out)) || n_shards == 0 || key_levels > 8) return fail(RGR_EINVAL, "rgr_shard_assign: bad argument");
        const uint32_t K = key_levels ? key_levels : 3;
        for (uint64_t i = 0; i < n; ++i) {
            const std::string_view s(reinterpret_cast<const char*>(blob) + offsets[i], offsets[i + 1] - offsets[i]);
            uint64_t h = 0xcbf29ce484222325ull;
            bool replicate = false;
            size_t start = 0;
            for (uint32_t l = 0; l < K; ++l) {
                const size_t pos = s.find('/', start);
                const std::string_view lv = s.substr(start, pos == std::string_view::npos ? std::string_view::npos : pos - start);
                if (is_filter && (lv == "+" || lv == "#")) { replicate = true; break; }
                for (unsigned char ch : lv) { h ^= ch; h *= 0x100000001b3ull; }
                h ^= 0x2f; h *= 0x100000001b3ull;                 // level terminator
                if (pos == std::string_view::npos) break;          // shorter than K levels
                start = pos + 1;
            }
            if (replicate) { out[i] = -1; continue; }
            h ^= h >> 32; h *= 0xd6e8feb86659fd93ull; h ^= h >> 32;
            out[i] = int32_t(h % n_shards);
        }
        return RGR_OK;
    });
}

// ------------------------------------------------------------------ observability
int32_t rgr_stats_get(rgr_handle* h, rgr_stats* out) {
    return guarded([&]() -> int32_t {
        if (!h || !out) return fail(RGR_EINVAL, "rgr_stats_get: bad argument");
        {
            std::lock_guard<std::mutex> g(h->stats_mu);
            *out = h->stats;
        }
        auto ep = current_epoch(h);
        out->n_filters = ep->n_filters; out->n_subs = ep->n_subs; out->n_nodes = ep->n_nodes;
        out->n_edge_slots = ep->edge_slots; out->epoch = ep->id; out->table_bytes_device = ep->bytes;
        {
            std::lock_guard<std::mutex> g(h->epoch_mu);
            out->retain_epoch = h->retain_epoch ? h->retain_epoch->id : 0;
            out->retain_topics = h->retain_epoch ? h->retain_epoch->n_topics : 0;
            out->retain_merges = h->retain_merges;
        }
        if (h->retain_tiered()) {
            std::shared_lock<std::shared_mutex> lk(h->retain_mu);
            out->retain_topics = h->tiers.n_topics();
            out->retain_delta_topics = h->tiers.n_delta();
            out->retain_dead = h->tiers.n_dead();
        }
        {
            std::lock_guard<std::mutex> cg(h->commit_mu);
            out->commits_full = h->commits_full; out->commits_delta = h->commits_delta;
        }
        std::shared_lock<std::shared_mutex> lk(h->table_mu);
        out->n_tokens = h->table.n_tokens();
        return RGR_OK;
    });
}

int32_t rgr_stats_reset(rgr_handle* h) {
    if (!h) return fail(RGR_EINVAL, "rgr_stats_reset: bad argument");
    std::lock_guard<std::mutex> g(h->stats_mu);
    h->stats = rgr_stats{};
    return RGR_OK;
}

}  // extern "C"

#include "retain_abi.inc"
#include "group_abi.inc"
