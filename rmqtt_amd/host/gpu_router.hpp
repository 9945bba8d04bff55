// C++ host-side mirror of rmqtt's `Router` trait / `DefaultRouter` surface over the C ABI.
//
// The reference's host language is Rust (not in this image), so this is the host side that is
// actually compiled and tested; rust/rmqtt-gpu-router/ holds the equivalent Rust source.
// Same method names, argument meaning and error behaviour as rmqtt/src/router.rs:65-112:
//
//   add / remove / matches / is_online / gets / get / topics_tree / topics / routes /
//   merge_topics / merge_routes / list_topics / list_relations / relations
//
// GpuRouter keeps the subscription table exactly where DefaultRouter keeps it — a relations map
// keyed by the filter string (router.rs:121-127, types.rs:476) — and mirrors every add/remove
// into the device table through rgr_filter_add/rgr_sub_add/...  Only `matches` changes: the
// trie walk + relation expansion run on the GPU and come back as (topic_idx, sub_id, qos)
// tuples, which are mapped back through a sub_id slab.  The per-hit decisions of _matches —
// No Local (router.rs:196-201: whole-`Id` equality) and the v5 collector's first-hit-per-client
// rule (types.rs:524-539) — are taken by the device's delivery stage (rgr_match_batch_deliver):
// every relation is registered with a dense id of its `Id`, of its (node, ClientId) and of its
// node (rgr_sub_add_ex), and `matches` reads the RGR_HIT_* flags.  Shared-group members
// (router.rs:202-213) are flagged RGR_SUB_SHARED, collected per (filter, group) while the hits of
// one filter go by, and one of them is chosen through the `SharedSubscription` the broker installs
// (router.rs:236-255) — the same place, order and arguments as the reference.
//
// The table lives in an rgr_group: one shard per device (`devices`), filters and publishes routed
// by rgr_shard_assign — with one device this is exactly a single handle.  `Batcher` is the
// deadline micro-batcher that sits between the per-publish trait call and the batched device
// pass (the twin of rust/rmqtt-gpu-router/src/batcher.rs).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <optional>
#include <shared_mutex>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "rmqtt_gpu_router.h"

namespace rmqtt {

using NodeId = uint64_t;
using ClientId = std::string;
using TopicFilter = std::string;
using TopicName = std::string;
// types.rs:1899-1911; equality over every field (types.rs:1841-1851).
struct Id {
    NodeId node_id = 0;
    uint16_t lid = 0;
    std::string local_addr, remote_addr;
    ClientId client_id;
    std::string username;
    int64_t create_time = 0;
    bool operator==(const Id& o) const {
        return node_id == o.node_id && lid == o.lid && client_id == o.client_id && local_addr == o.local_addr &&
               remote_addr == o.remote_addr && username == o.username && create_time == o.create_time;
    }
    bool operator!=(const Id& o) const { return !(*this == o); }
};

// The table's reader / writer lock.  std::shared_mutex is pthread_rwlock with glibc's default: readers are preferred, and a writer waits until NO reader
// holds the lock — with two passes in flight and 32 completion threads taking it in turns that is practically never: measured at full publish load
// (bench.py --router-e2e --e2e-churn, profiles/r07x_*) a subscribe waited 1.1 s.  Here a waiting writer stops NEW readers (they yield until it has
// been through), so it gets the lock as soon as the current holders are done.  No thread takes the lock shared twice (GpuRouter::SharedHold).
class TableMutex {
   public:
    void lock() { writers_.fetch_add(1, std::memory_order_acq_rel); m_.lock(); writers_.fetch_sub(1, std::memory_order_acq_rel); }
    bool try_lock() { return m_.try_lock(); }
    void unlock() { m_.unlock(); }
    void lock_shared() {
        // (a writer is usually through in a fraction of a millisecond — an add, a remove, a commit; behind a long one (restore) the readers sleep instead of spinning)
        for (unsigned spins = 0; writers_.load(std::memory_order_acquire) > 0; ++spins) {
            if (spins < 256) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
        m_.lock_shared();
    }
    bool try_lock_shared() { return writers_.load(std::memory_order_acquire) == 0 && m_.try_lock_shared(); }
    void unlock_shared() { m_.unlock_shared(); }

   private:
    std::shared_mutex m_;
    std::atomic<int> writers_{0};
};

// types.rs:607-827 (fields the matching path carries).
struct SubscriptionOptions {
    bool v5 = false;
    uint8_t qos = 0;
    bool no_local = false, retain_as_published = false;
    uint8_t retain_handling = 0;
    uint32_t subscription_identifier = 0;   // 0 = None
    std::optional<std::string> shared_group;     // types.rs:776,815: <group> of $share/<group>/<filter>
    bool is_v3() const { return !v5; }
    std::optional<bool> opt_no_local() const { return v5 ? std::optional<bool>(no_local) : std::nullopt; }
};

struct SharedGroupType { std::string group; bool is_online = true; std::vector<ClientId> group_cids; };   // types.rs:474
// Rows hold their own copies of the filter and the client id.  (r3 measured the alternative — ref-counted strings like the reference's
// ByteString clones — on 256 threads at config-3 fan-out: 8.8 M rows/s against 38 M rows/s for plain copies; every thread bumps the reference
// counts of the same few hot filters and clients, and those cache lines ping-pong.  profiles/r03f_router_e2e_refcounted_rows_slower.jsonl)
struct SubRelation {   // types.rs:478-484
    TopicFilter topic_filter;
    ClientId client_id;
    SubscriptionOptions opts;
    std::optional<std::vector<uint32_t>> sub_ids;
    std::optional<SharedGroupType> group;        // Some for the member a shared group selected
};
// rmqtt/src/subscribe.rs:69-95.  The default selects nobody, like DefaultSharedSubscription
// (subscribe.rs:107): without the shared-subscription plugin $share members receive nothing.
struct SharedCandidate { NodeId node_id; ClientId client_id; SubscriptionOptions opts; bool is_online; };
struct SharedSubscription {
    virtual ~SharedSubscription() = default;
    virtual bool is_supported() const { return false; }
    virtual std::optional<std::pair<size_t, bool>> choice(const std::string& /*group*/, const struct Id& /*publisher*/, const TopicName& /*topic*/,
                                                          const std::vector<SharedCandidate>& /*ncs*/) { return std::nullopt; }
};
using SubRelations = std::vector<SubRelation>;
using SubRelationsMap = std::map<NodeId, SubRelations>;   // types.rs:486

struct Route { NodeId node_id; TopicFilter topic; };

// anyhow::Result stand-in: ok() or an error string; never throws across the trait.
template <class T> struct Result {
    std::optional<T> value;
    std::string error;
    bool ok() const { return value.has_value(); }
    static Result Ok(T v) { Result r; r.value = std::move(v); return r; }
    static Result Err(std::string e) { Result r; r.error = std::move(e); return r; }
};

namespace raft { struct Snapshot; }   // raft_snapshot.hpp

struct Counter {   // rmqtt-utils/src/counter.rs:39 (count, max)
    int64_t count = 0, max = 0;
    void inc() { if (++count > max) max = count; }
    void dec() { --count; }
};

class Router {   // rmqtt/src/router.rs:65-112
   public:
    virtual ~Router() = default;
    virtual Result<bool> add(const std::string& topic_filter, const Id& id, const SubscriptionOptions& opts) = 0;
    virtual Result<bool> remove(const std::string& topic_filter, const Id& id) = 0;
    virtual Result<SubRelationsMap> matches(const Id& id, const TopicName& topic) = 0;
    virtual bool is_online(NodeId node_id, const std::string& client_id) = 0;
    virtual std::vector<Route> gets(size_t limit) = 0;
    virtual Result<std::vector<Route>> get(const std::string& topic) = 0;
    virtual size_t topics_tree() = 0;
    virtual Counter topics() = 0;
    virtual Counter routes() = 0;
    virtual std::vector<std::string> list_topics(size_t top) = 0;
};

class GpuRouter final : public Router {
   public:
    explicit GpuRouter(NodeId this_node, int device = 0);
    // one shard per entry (ShardedGpuRouter of the Rust crate): rgr_group over these devices
    GpuRouter(NodeId this_node, const std::vector<int>& devices);
    ~GpuRouter() override;
    // extends.shared_subscription() and Router::is_online of the broker (router.rs:204-213, 236-245)
    void set_shared_subscription(std::shared_ptr<SharedSubscription> s) { shared_ = std::move(s); }
    void set_is_online(std::function<bool(NodeId, const ClientId&)> f) { is_online_ = std::move(f); }
    // v5 hits whose RGR_HIT_V5_DUP flag disagreed with the host collector although no shared member had been
    // chosen for that publish (must stay 0: the device's first-hit-per-client rule equals types.rs:524-539)
    uint64_t flag_mismatches() const { return flag_mismatches_; }
    uint32_t shards() const;
    GpuRouter(const GpuRouter&) = delete;
    GpuRouter& operator=(const GpuRouter&) = delete;
    bool usable() const { return g_ != nullptr; }
    const std::string& create_error() const { return create_error_; }

    Result<bool> add(const std::string& topic_filter, const Id& id, const SubscriptionOptions& opts) override;
    Result<bool> remove(const std::string& topic_filter, const Id& id) override;
    Result<SubRelationsMap> matches(const Id& id, const TopicName& topic) override;
    // Batched form for a micro-batcher in front of the trait: one device pass for many publishes.
    // out[i] is nullopt where the reference would return Err (invalid topic name).
    Result<bool> matches_batch(const std::vector<Id>& ids, const std::vector<TopicName>& topics,
                               std::vector<std::optional<SubRelationsMap>>& out);
    // How `matches` asks the device (r3):
    //   Filters  rgr_group_match_filter_subs: the device walks the trie and returns, per publish, one sub id per matched filter
    //            (4 B per FILTER over PCIe); the per-client loop of router.rs:194-231 — No Local, $share, the v3/v5 collector —
    //            runs on the host over this router's own relations map, exactly where the reference runs it.  This is the
    //            default: at config-3 fan-out the 12-byte tuples of the Deliver path are 178 KB per publish over PCIe.
    //   Deliver  rgr_group_match_batch_deliver: 12-byte tuples carrying the device's delivery words (No Local, v5 first-hit
    //            flags, node index); the host only folds them into the map.  Kept for low fan-out and as the cross-check of the
    //            device's delivery stage (flag_mismatches()).
    //   Auto     Filters, unless the running mean of hits per publish is below kAutoDeliverBelow.
    enum class MatchMode { Auto, Filters, Deliver };
    void set_match_mode(MatchMode m) { mode_ = m; }
    static constexpr double kAutoDeliverBelow = 8.0;
    // The two halves of the Filters path, for a batcher that lets every CALLER expand its own publish in parallel (as every
    // tokio worker does in the reference): one device pass for the whole batch, then expand() per publish on any thread.
    struct FilterPass {
        rgr_filters_result res{};
        uint64_t epoch = 0;                 // restore epoch the pass saw (removals do not make it stale since r8j: Slot / limbo_)
        GpuRouter* lease_of = nullptr;      // counted in the router's pass generation while it lives (set by device_pass)
        unsigned lease_parity = 0;
        FilterPass() = default;
        FilterPass(const FilterPass&) = delete;
        FilterPass& operator=(const FilterPass&) = delete;
        ~FilterPass() { if (lease_of) lease_of->live_passes_[lease_parity].fetch_sub(1, std::memory_order_acq_rel); rgr_filters_result_free(&res); }
    };
    Result<bool> filters_pass(const std::vector<TopicName>& topics, FilterPass& pass);
    Result<bool> filters_pass(const std::string& blob, const std::vector<uint64_t>& offs, FilterPass& pass);      // topics already packed
    // Err: what the reference returns for an invalid topic name ("invalid topic ..."), or a device failure of the re-match
    Result<SubRelationsMap> expand(const FilterPass& pass, size_t t, const Id& id, const TopicName& topic);
    // n publishes of one pass (index[i] = position inside the pass) under ONE acquisition of the table lock
    void expand_chunk(const FilterPass& pass, const size_t* index, const Id* const* ids, const TopicName* const* topics, size_t n,
                      std::vector<Result<SubRelationsMap>>& out, bool* stale = nullptr);      // stale (optional): set instead of re-matching one by one when a removal overtook the pass
    Result<SubRelationsMap> rematch_public(const Id& id, const TopicName& topic) { return rematch(id, topic); }
    uint64_t stale_expansions() const { return stale_expansions_; }
    // ---- the delivery stage as a pass of its own (r6): what Shared::forwards consumes WITHOUT a SubRelationsMap in between (gpu_shared.hpp).
    // One device pass (rgr_group_match_batch_deliver) for a batch of publishes with their real qos / retain bits: per hit the device has
    // already taken forwards_to's per-recipient decisions (shared.rs:886-903: qos' = min, Retain-As-Published) and _matches' (router.rs:196-201
    // No Local; types.rs:524-539 first hit per v5 client).  `ids` are the publishers (No Local compares whole Ids).
    struct DeliverPass {
        rgr_result res{};
        uint64_t epoch = 0;                 // restore epoch the pass saw (a removal does NOT make a delivery pass stale: see limbo_ below)
        GpuRouter* lease_of = nullptr;      // the router whose generation counter this pass is counted in while it lives (set by deliver_pass)
        unsigned lease_parity = 0;
        DeliverPass() = default;
        DeliverPass(const DeliverPass&) = delete;
        DeliverPass& operator=(const DeliverPass&) = delete;
        ~DeliverPass() { if (lease_of) lease_of->live_passes_[lease_parity].fetch_sub(1, std::memory_order_acq_rel); rgr_result_free(&res); }
    };
    // A publisher's dense owner id, looked up where the publish is SUBMITTED (many threads) instead of once per publish inside the pass (one driver
    // thread: 0.5 ms of a pass of 3 600 publishes) — or not at all: a session keeps the hint of its own Id (gpu_shared.hpp `From`).  The hint carries
    // the epoch of the owner index it was read at (bumped whenever an Id gets or loses its owner id): deliver_pass looks the Id up again when the
    // index has changed since, so a recycled or newly assigned owner id is never missed.
    // (r8k) The epoch is PER BUCKET of the Id's hash (kOwnerBuckets of them): one client's first subscribe or last unsubscribe invalidates the cached
    // hints of 1 / 4 096 of the publishers, not of all of them — with one global epoch a subscriber coming and going 420 times a second sent every
    // publish of 4 M/s back to the locked lookup (profiles/r08j_*: 1.6 M).  `bucket` travels with the hint (the hash of an Id never changes).
    static constexpr uint32_t kOwnerBuckets = 4096;
    struct OwnerHint { uint32_t owner = RGR_ID_NONE; uint32_t bucket = 0; uint64_t epoch = ~0ull; };
    static uint32_t owner_bucket_of(const Id& id) { return uint32_t(IdHash{}(id) >> 7) % kOwnerBuckets; }
    uint64_t owners_epoch(uint32_t bucket) const { return owner_bucket_epoch_[bucket].load(std::memory_order_acquire); }
    bool owner_hint_current(const OwnerHint& h) const { return h.bucket < kOwnerBuckets && h.epoch == owners_epoch(h.bucket); }
    Result<bool> deliver_pass(const std::string& blob, const std::vector<uint64_t>& offs, const Id* const* ids, const uint8_t* qos_retain, DeliverPass& pass,
                              const OwnerHint* hints = nullptr);
    OwnerHint owner_hint(const Id& id) {
        std::shared_lock<TableMutex> g(mu_, std::defer_lock);
        if (held_by_this_thread() != this) g.lock();                  // (a resubmission from inside a worker's SharedHold)
        const uint32_t b = owner_bucket_of(id);
        return OwnerHint{owners_.find(id), b, owners_epoch(b)};
    }
    // One recipient of one publish, as the device decided it; pointers into the router's relations map (valid while the visitor runs).
    struct Delivery {
        const ClientId* client_id; const TopicFilter* topic_filter; NodeId node_id;
        uint8_t qos; bool retain;                      // shared.rs:886-902
        bool is_v5;                                    // SubscriptionOptions::V5 (the collector treats v3 and v5 hits differently, types.rs:519-539)
        uint32_t subscription_identifier;              // 0 = none; of THIS hit
        bool v5_duplicate;                             // later hit of a v5 client already delivered to: only its identifier counts (types.rs:526-534)
    };
    // NeedsHostPath: $share members among the hits — the caller takes the reference's own path for this publish.  Stale: the pass is older than the last
    // removal (a sub id may have been recycled): the publish has to be matched again — through another batch (GpuShared resubmits it), not one by one.
    enum class DeliverOutcome { Done, InvalidTopic, NeedsHostPath, Stale };
    // Every hit of publish `t` of the pass that is not dropped by No Local, in TopicTree::matches order; `visit` returns nothing.  Holds the table's
    // shared lock while it runs (like expand()).
    template <class Visit> DeliverOutcome visit_deliveries(const DeliverPass& pass, size_t t, Visit&& visit) {
        std::shared_lock<TableMutex> g(mu_, std::defer_lock);
        if (held_by_this_thread() != this) g.lock();              // (a worker's run of publishes holds it once: SharedHold)
        if (pass.epoch != restore_epoch_.load(std::memory_order_acquire)) { stale_expansions_++; return DeliverOutcome::Stale; }
        if (pass.res.status[t] != RGR_TOPIC_OK) return DeliverOutcome::InvalidTopic;
        const uint64_t lo = pass.res.hit_offsets[t], hi = pass.res.hit_offsets[t + 1];
        if (shared_rels_)
            for (uint64_t k = lo; k < hi; ++k) if ((pass.res.tuples[k].qos_flags >> 8) & RGR_SUB_SHARED) return DeliverOutcome::NeedsHostPath;
        for (uint64_t k = lo; k < hi; ++k) {
            const uint32_t w = pass.res.tuples[k].qos_flags;
            if (w & RGR_HIT_NO_LOCAL) continue;
            const Slot& sl = slab_[pass.res.tuples[k].sub_id];
            if (!sl.rel) continue;                                    // the relation was removed after the pass matched it: nobody to deliver to (its id is not reused while this pass lives)
            const Rel& rel = *sl.rel;
            visit(Delivery{&rel.id.client_id, sl.filter, bulk_loaded_ ? rel.id.node_id : nodes_[w >> 16], uint8_t(w & RGR_HIT_QOS_MASK), (w & RGR_HIT_RETAIN) != 0,
                           rel.opts.v5, rel.opts.v5 ? rel.opts.subscription_identifier : 0u, (w & RGR_HIT_V5_DUP) != 0});
        }
        return DeliverOutcome::Done;
    }
    NodeId this_node() const { return this_node_; }
    // The table's shared lock for a RUN of publishes of one pass (the batcher's worker tasks): one acquisition per run instead of one per publish —
    // 2.7 M lock / unlock pairs per second from 32 threads are 5 M read-modify-writes of ONE cache line.  visit_deliveries sees the hold through a
    // thread-local and does not lock again (a second shared acquisition behind a waiting writer could deadlock); a completion that has to leave the
    // device path (Shared::forwards' host path takes the lock itself, possibly exclusively) pauses the hold around it.
    class SharedHold {
       public:
        explicit SharedHold(GpuRouter& r) : r_(r) { r_.mu_.lock_shared(); held_by_this_thread() = &r_; }
        ~SharedHold() { held_by_this_thread() = nullptr; r_.mu_.unlock_shared(); }
        SharedHold(const SharedHold&) = delete;
        SharedHold& operator=(const SharedHold&) = delete;
       private:
        GpuRouter& r_;
    };
    class SharedPause {                  // inside a SharedHold of this thread: unlocked for the scope (no-op when there is no hold)
       public:
        explicit SharedPause(GpuRouter& r) : r_(r), was_(held_by_this_thread() == &r) { if (was_) { held_by_this_thread() = nullptr; r_.mu_.unlock_shared(); } }
        ~SharedPause() { if (was_) { r_.mu_.lock_shared(); held_by_this_thread() = &r_; } }
        SharedPause(const SharedPause&) = delete;
        SharedPause& operator=(const SharedPause&) = delete;
       private:
        GpuRouter& r_;
        bool was_;
    };
    bool is_online(NodeId node, const std::string& client) override { return is_online_ ? is_online_(node, client) : true; }   // session state lives in the broker
    std::vector<Route> gets(size_t limit) override;
    Result<std::vector<Route>> get(const std::string& topic) override;       // router.rs:157-170 (.unique())
    Result<bool> has_matches(const std::string& topic);                     // router.rs:151-154
    size_t topics_tree() override;
    Counter topics() override { return topics_count_; }
    Counter routes() override { return relations_count_; }
    std::vector<std::string> list_topics(size_t top) override;
    // ClusterRouter::restore (rmqtt-cluster-raft/src/router.rs:466-580) from a decoded Raft snapshot
    // (raft::decode_snapshot): the relations map is replaced and the device table is rebuilt through the bulk
    // path (rgr_group_subscribe_bulk: one sort + compile instead of one trie insert per relation).
    // Unlike the reference, which stops half-way at the first invalid filter (router.rs:559 `?` after
    // relations.clear()), every filter is validated first and an Err leaves the router as it was.
    Result<bool> restore(const raft::Snapshot& snap);

   private:
    static const GpuRouter*& held_by_this_thread() { static thread_local const GpuRouter* held = nullptr; return held; }
    struct Rel { Id id; SubscriptionOptions opts; uint32_t sub_id; uint32_t owner_id; };
    struct Dense {                       // string key -> dense u32 id with reference counts
        std::unordered_map<std::string, std::pair<uint32_t, uint32_t>> ids;   // key -> (id, refs)
        std::vector<uint32_t> free;
        uint32_t next = 0;
        uint32_t acquire(const std::string& k);
        void release(const std::string& k);
        uint32_t find(const std::string& k) const;                             // RGR_ID_NONE if absent
    };
    // A filter's relations.  Shared: a slot whose relation has been removed keeps its filter's entry (and through it the filter string) until the slot
    // is handed out again — so a FILTER pass that holds the id of a removed relation as its filter's representative still expands the filter's
    // CURRENT relations instead of going stale (r8j; ids are not handed out again while a pass that may hold them lives: limbo_).
    struct FilterEntry { std::string filter; std::unordered_map<ClientId, Rel> rels; };
    struct Slot { const std::string* filter = nullptr; const Rel* rel = nullptr; std::shared_ptr<const FilterEntry> entry; };      // filter = &entry->filter

    rgr_group* g_ = nullptr;
    std::shared_ptr<SharedSubscription> shared_;
    std::function<bool(NodeId, const ClientId&)> is_online_;
    uint64_t flag_mismatches_ = 0;
    std::string create_error_;
    NodeId this_node_;
    std::vector<int32_t> devices_;
    uint64_t shared_rels_ = 0;           // relations that are $share members (their picks need the ungrouped hit order)
    bool bulk_loaded_ = false;           // relations loaded by restore(): the tuples' node bits are not populated
    // add / remove / restore / commit: exclusive; a device pass and the host expansion of its result: shared (the reference:
    // DashMap + a trie RwLock) — so several passes walk the same committed table at once.  A sub id freed by remove() is
    // kept out of circulation until the device table has dropped it AND no pass that may hold it lives (limbo_ below), so a recycled id can never
    // resolve to a relation the device did not match; only restore() — which renumbers everything — makes a pass stale: expand() re-runs its publishes.
    // (r7z) DELIVERY passes carry one sub id per hit, so a removal does not invalidate them: a hit whose relation is gone is skipped, and a freed sub id
    // is not handed out again while a delivery pass that may hold it lives.  Two generations: removals go to limbo_[current]; a delivery pass is counted
    // in live_passes_[generation it started in]; a successful commit with no pass of the PREVIOUS generation alive frees that generation's limbo (those
    // ids left the device table at the commit that ended it, and every pass that started before is gone) and starts a new generation.  Only restore()
    // — which renumbers everything — makes passes stale (restore_epoch_).  With the epoch rule alone 410 unsubscribes a second sent a quarter
    // of 4 M publishes/s round again (profiles/r07y_*).  Filter passes (their ids stand for whole filters) live by the same rule since r8j: Slot.
    TableMutex mu_;
    std::atomic<uint64_t> restore_epoch_{0};
    std::atomic<uint64_t> stale_expansions_{0};
    std::vector<uint32_t> limbo_[2];
    uint64_t pass_generation_ = 0;               // (changed under the exclusive lock only; passes read it under the shared lock)
    std::atomic<int64_t> live_passes_[2] = {{0}, {0}};
    MatchMode mode_ = MatchMode::Auto;
    std::atomic<double> mean_hits_{1e9};         // running mean of hits per publish (Auto)
    std::unordered_map<TopicFilter, std::shared_ptr<FilterEntry>> relations_;   // AllRelationsMap
    std::vector<Slot> slab_;           // sub_id -> relation
    std::vector<uint32_t> free_sub_ids_;
    // Id -> owner_id, keyed by the Id itself (every field Id equality looks at, types.rs:1841-1851): the publisher of EVERY publish of a delivery
    // pass is looked up here, and a joined string key cost ~0.4 us per publish of allocations (r7c: 1.1 ms of a 1.5 ms pass of 2 763 publishes)
    struct IdHash {
        size_t operator()(const Id& id) const {
            const std::hash<std::string_view> h;
            uint64_t x = (uint64_t(id.node_id) << 16 | id.lid) * 0x9E3779B97F4A7C15ull ^ uint64_t(id.create_time);
            for (const std::string* f : {&id.client_id, &id.local_addr, &id.remote_addr, &id.username}) x = (x ^ h(std::string_view(*f))) * 0xFF51AFD7ED558CCDull, x ^= x >> 29;
            return size_t(x);
        }
    };
    struct OwnerIndex {                  // Id -> dense u32 id with reference counts (Dense, keyed by Id)
        std::unordered_map<Id, std::pair<uint32_t, uint32_t>, IdHash> ids;
        std::vector<uint32_t> free;
        uint32_t next = 0;
        // changed (optional): set when the key appeared / disappeared — the caller bumps the epoch of the key's bucket (GpuRouter::owner_bucket_epoch_)
        uint32_t acquire(const Id& k, bool* changed = nullptr) {
            auto it = ids.find(k);
            if (it != ids.end()) { it->second.second++; return it->second.first; }
            uint32_t id;
            if (!free.empty()) { id = free.back(); free.pop_back(); } else id = next++;
            ids.emplace(k, std::make_pair(id, 1u));
            if (changed) *changed = true;
            return id;
        }
        void release(const Id& k, bool* changed = nullptr) {
            auto it = ids.find(k);
            if (it == ids.end()) return;
            if (--it->second.second == 0) { free.push_back(it->second.first); ids.erase(it); if (changed) *changed = true; }
        }
        uint32_t find(const Id& k) const { auto it = ids.find(k); return it == ids.end() ? RGR_ID_NONE : it->second.first; }
    };
    OwnerIndex owners_;
    // bumped (under the exclusive lock) when an Id of the bucket gets or loses its owner id; restore() bumps every bucket.  Starts at 1: 0 = "never read".
    std::atomic<uint64_t> owner_bucket_epoch_[kOwnerBuckets];
    void bump_owner_bucket(const Id& id) { owner_bucket_epoch_[owner_bucket_of(id)].fetch_add(1, std::memory_order_acq_rel); }
    Dense clients_;                      // (node, ClientId) -> client_idx
    std::vector<NodeId> nodes_;          // node_idx -> NodeId
    std::unordered_map<NodeId, uint16_t> node_idx_;
    Counter topics_count_, relations_count_;
    std::atomic<bool> dirty_{false};

    int32_t commit_if_dirty();           // caller holds mu_ exclusively
    Result<bool> filters_pass_locked(const std::string& blob, const std::vector<uint64_t>& offs, FilterPass& pass);   // mu_ held exclusively
    Result<bool> device_pass(const std::string& blob, const std::vector<uint64_t>& offs, FilterPass& pass);           // mu_ held, table committed
    Result<SubRelationsMap> rematch(const Id& id, const TopicName& topic);
    Result<bool> matches_batch_deliver(const std::vector<Id>& ids, const std::vector<TopicName>& topics, std::vector<std::optional<SubRelationsMap>>& out);
    std::optional<SubRelationsMap> expand_locked(const rgr_filters_result& res, size_t t, const Id& id, const TopicName& topic, uint64_t* hits);
};

// Deadline micro-batcher in front of the batched device pass: Router::matches is called once per PUBLISH (rmqtt/src/shared.rs:772).
//   matches()  blocking form: the caller enqueues (id, topic), sleeps on its own slot and expands its publish itself.
//   submit()   asynchronous form — what a tokio task awaiting `matches` is: enqueue and return; the completion (the publish's
//              SubRelationsMap) is delivered through `cb` on one of `workers` pool threads (the tokio workers' stand-in), which
//              expand runs of publishes of one pass under one acquisition of the table lock.  Tens of thousands of publishes
//              can be outstanding from a handful of threads, so batches fill to max_batch instead of to the thread count.
// `passes_in_flight` driver threads each collect a batch (max_batch publishes, or max_delay after its first one) and run ONE
// device pass for it; the library gives every pass its own stream and workspace, so the next batch is collected and walked while
// the previous one is still on the device (r4; r3 ran one pass at a time: 241 k publishes/s at config 2).  No lock of the router
// is held while callers wait.  Rust twin: rust/rmqtt-gpu-router/src/batcher.rs (MAX_IN_FLIGHT; tokio's own workers).
class Batcher {
   public:
    // completion of an asynchronous publish: `user` and `tag` are the caller's (a plain function pointer + two words instead of a
    // std::function: no heap allocation per publish — r4d: allocator and cache-line traffic between submitters and workers, not the
    // device, capped the boundary at 2.3 M publishes/s)
    using Callback = void (*)(void* user, uint64_t tag, Result<SubRelationsMap>&& result);
    Batcher(GpuRouter& router, size_t max_batch, std::chrono::microseconds max_delay, unsigned passes_in_flight = 3, unsigned workers = 0);
    ~Batcher();
    Result<SubRelationsMap> matches(const Id& id, const TopicName& topic);
    void submit(const Id& id, std::string_view topic, Callback cb, void* user, uint64_t tag);
    // (r6) the same queueing for Shared::forwards (gpu_shared.hpp): a request carries the publish's qos / retain, its batch runs ONE delivery
    // pass (GpuRouter::deliver_pass), and the completion — on a pool thread, or on the driver without a pool — receives the pass and the
    // request's index in it instead of a SubRelationsMap (err != nullptr: the pass failed).  A batcher serves one kind of request.
    using DeliverCallback = void (*)(void* user, uint64_t tag, const std::shared_ptr<GpuRouter::DeliverPass>& pass, size_t index, const Id& from, const std::string* err);
    // hint: the publisher's cached OwnerHint (checked against the owner index's epoch here and again inside the pass); null = looked up here
    void submit_deliver(const Id& from, std::string_view topic, uint8_t qos_retain, DeliverCallback cb, void* user, uint64_t tag, const GpuRouter::OwnerHint* hint = nullptr);
    uint64_t passes() const { return passes_; }
    uint64_t requests() const { uint64_t n = 0; for (const Shard& sh : shards_) { std::lock_guard<std::mutex> g(sh.m); n += sh.requests; } return n; }
    // where the drivers' and workers' time went (nanoseconds summed over threads): collecting a batch (incl. the deadline wait's tail
    // and packing the topics), the device pass (GpuRouter::filters_pass), handing the results on, and the workers' expansion tasks
    struct Timing { uint64_t collect_ns, pass_ns, dispatch_ns, task_ns, tasks, max_task_queue, requeued; };
    Timing timing() const { return Timing{collect_ns_, pass_ns_, dispatch_ns_, task_ns_, tasks_run_, max_task_queue_, requeued_}; }

   private:
    // blocking requests live on their caller's stack and are woken through their own condition variable (one shared cv made 256
    // callers fight for one mutex per pass); asynchronous ones are heap objects that end with their callback
    struct Req { Id id; TopicName topic; Callback cb = nullptr; DeliverCallback dcb = nullptr; uint8_t qos_retain = 0; void* user = nullptr; uint64_t tag = 0; uint32_t shard = 0;
                 GpuRouter::OwnerHint owner; unsigned tries = 0;
                 std::shared_ptr<GpuRouter::FilterPass> pass; size_t index = 0; std::string err; bool done = false;
                 std::mutex m; std::condition_variable cv; };
    // a submitter sticks to one shard: its queue, and the free list its asynchronous requests are recycled through (a finished
    // request goes back to the shard it came from, so request objects — and the capacity of their strings — stay with their submitter
    // instead of crossing the allocator's arenas on every publish)
    struct alignas(64) Shard { mutable std::mutex m; std::vector<Req*> q; std::vector<Req*> free; uint64_t requests = 0; };
    struct Task { std::shared_ptr<GpuRouter::FilterPass> pass; std::shared_ptr<GpuRouter::DeliverPass> dpass; std::vector<Req*> reqs; };
    static constexpr size_t kShards = 16;      // submission queues (a submitter sticks to one)
    static constexpr size_t kTaskRun = 64;     // publishes per worker task
    static constexpr unsigned kMaxRequeues = 8;  // a publish whose FILTER pass a removal overtook joins another batch that many times at most (then the one-publish re-match)
    std::atomic<uint64_t> requeued_{0};
    GpuRouter& router_;
    size_t max_batch_;
    std::chrono::microseconds max_delay_;
    Shard shards_[kShards];
    std::atomic<size_t> pending_{0};
    std::mutex mu_;                            // drivers sleep here
    std::condition_variable cv_req_;
    std::atomic<bool> stop_{false};
    std::atomic<int> sleepers_{0};             // drivers inside a condition-variable wait (submitters only notify when there is one)
    std::atomic<uint64_t> passes_{0};
    std::atomic<uint64_t> collect_ns_{0}, pass_ns_{0}, dispatch_ns_{0}, task_ns_{0}, tasks_run_{0};
    std::atomic<size_t> next_shard_{0};
    size_t max_task_queue_ = 0;                // (under task_mu_)
    std::mutex task_mu_;
    std::condition_variable task_cv_;
    std::deque<Task> tasks_;
    bool task_stop_ = false;
    std::vector<std::thread> drivers_, workers_;
    static constexpr size_t kFreeMax = 1 << 16;      // recycled request objects kept per shard
    void enqueue(Req* r);
    void recycle(std::vector<Req*>& reqs);
    void run();
    void work();
    void run_task(Task& t);
};

}  // namespace rmqtt
