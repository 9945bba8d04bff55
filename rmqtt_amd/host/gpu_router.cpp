// GpuRouter — see gpu_router.hpp.  Uses nothing but the C ABI of rmqtt_gpu_router.h.
#include "gpu_router.hpp"
#include "raft_snapshot.hpp"

#include <algorithm>
#include <chrono>
#include <string_view>

namespace rmqtt {

namespace {
uint8_t flags_of(const SubscriptionOptions& o) {
    return uint8_t((o.v5 ? RGR_SUB_V5 : 0) | (o.v5 && o.no_local ? RGR_SUB_NO_LOCAL : 0) | (o.v5 && o.retain_as_published ? RGR_SUB_RAP : 0) |
                   (o.shared_group ? RGR_SUB_SHARED : 0));
}
std::string client_key(NodeId node, const ClientId& c) { return std::to_string(node) + '|' + c; }
}  // namespace

uint32_t GpuRouter::Dense::acquire(const std::string& k) {
    auto it = ids.find(k);
    if (it != ids.end()) { it->second.second++; return it->second.first; }
    uint32_t id;
    if (!free.empty()) { id = free.back(); free.pop_back(); } else id = next++;
    ids.emplace(k, std::make_pair(id, 1u));
    return id;
}
void GpuRouter::Dense::release(const std::string& k) {
    auto it = ids.find(k);
    if (it == ids.end()) return;
    if (--it->second.second == 0) { free.push_back(it->second.first); ids.erase(it); }
}
uint32_t GpuRouter::Dense::find(const std::string& k) const {
    auto it = ids.find(k);
    return it == ids.end() ? RGR_ID_NONE : it->second.first;
}

GpuRouter::GpuRouter(NodeId this_node, int device) : GpuRouter(this_node, std::vector<int>{device}) {}

GpuRouter::GpuRouter(NodeId this_node, const std::vector<int>& devices) : this_node_(this_node) {
    for (auto& e : owner_bucket_epoch_) e.store(1, std::memory_order_relaxed);
    rgr_config cfg{};
    devices_.assign(devices.begin(), devices.end());
    if (rgr_group_create(&cfg, devices_.data(), uint32_t(devices_.size()), &g_) != RGR_OK) { g_ = nullptr; create_error_ = rgr_last_error(); }
}

GpuRouter::~GpuRouter() { if (g_) rgr_group_destroy(g_); }

uint32_t GpuRouter::shards() const { return g_ ? rgr_group_size(g_) : 0; }

int32_t GpuRouter::commit_if_dirty() {
    if (!dirty_) return RGR_OK;
    int32_t rc = rgr_group_commit(g_);
    if (rc == RGR_OK) {
        dirty_ = false;
        // the device table no longer holds the ids removed before this commit.  Those of the PREVIOUS generation may be handed out again once no delivery
        // pass of that generation lives; then this commit also ends the current generation (gpu_router.hpp, limbo_).
        const unsigned old = 1u - unsigned(pass_generation_ & 1u);
        if (live_passes_[old].load(std::memory_order_acquire) == 0) {
            free_sub_ids_.insert(free_sub_ids_.end(), limbo_[old].begin(), limbo_[old].end());
            limbo_[old].clear();
            ++pass_generation_;
        }
    }
    return rc;
}

// Topic::from_str (topic.rs:357-394, 231-243): '+' / '#' only as whole levels, '#' only last, '$' only first.
static bool valid_topic(const std::string& s) {
    size_t start = 0, level = 0;
    bool hash_seen = false;
    for (;;) {
        const size_t pos = s.find('/', start);
        const std::string_view lv(s.data() + start, (pos == std::string::npos ? s.size() : pos) - start);
        if (hash_seen) return false;
        if (lv == "#") hash_seen = true;
        else if (lv != "+") {
            if (lv.find('+') != std::string_view::npos || lv.find('#') != std::string_view::npos) return false;
            if (!lv.empty() && lv[0] == '$' && level != 0) return false;
        }
        if (pos == std::string::npos) return true;
        start = pos + 1; ++level;
    }
}

// router.rs:434-453
Result<bool> GpuRouter::add(const std::string& topic_filter, const Id& id, const SubscriptionOptions& opts) {
    if (!g_) return Result<bool>::Err(create_error_);
    if (!valid_topic(topic_filter)) return Result<bool>::Err("invalid topic filter `" + topic_filter + "`");   // router.rs:436 (`?`)
    std::unique_lock<TableMutex> g(mu_);
    // (an add cannot make a sub id held by a pass in flight resolve to another relation: ids are handed out again only out of the limbo that
    // remove() fills, when no pass that may hold them lives — gpu_router.hpp limbo_)
    auto it = relations_.find(topic_filter);
    if (it == relations_.end()) {
        topics_count_.inc();
        it = relations_.emplace(topic_filter, std::make_shared<FilterEntry>(FilterEntry{topic_filter, {}})).first;
    }
    auto& rels = it->second->rels;
    auto old = rels.find(id.client_id);
    uint32_t sub_id;
    bool owner_appeared = false;
    const uint32_t owner_id = owners_.acquire(id, &owner_appeared);
    if (owner_appeared) bump_owner_bucket(id);
    if (opts.shared_group) shared_rels_++;
    if (old != rels.end() && old->second.opts.shared_group) shared_rels_--;
    if (old == rels.end()) {
        relations_count_.inc();
        if (!free_sub_ids_.empty()) { sub_id = free_sub_ids_.back(); free_sub_ids_.pop_back(); }
        else { sub_id = uint32_t(slab_.size()); slab_.emplace_back(); }
        old = rels.emplace(id.client_id, Rel{id, opts, sub_id, owner_id}).first;
    } else {
        sub_id = old->second.sub_id;
        bool owner_left = false;
        owners_.release(old->second.id, &owner_left);
        if (owner_left) bump_owner_bucket(old->second.id);
        clients_.release(client_key(old->second.id.node_id, old->second.id.client_id));
        old->second = Rel{id, opts, sub_id, owner_id};              // HashMap::insert replaces (router.rs:447)
    }
    const uint32_t client_idx = clients_.acquire(client_key(id.node_id, id.client_id));
    auto ni = node_idx_.find(id.node_id);
    if (ni == node_idx_.end()) {
        if (nodes_.size() >= 0xFFFF) return Result<bool>::Err("more than 65535 distinct node ids");
        ni = node_idx_.emplace(id.node_id, uint16_t(nodes_.size())).first;
        nodes_.push_back(id.node_id);
    }
    slab_[sub_id] = Slot{&it->second->filter, &old->second, it->second};
    if (rgr_group_subscribe_ex(g_, topic_filter.data(), uint32_t(topic_filter.size()), sub_id, opts.qos, flags_of(opts), ni->second, owner_id,
                               client_idx) != RGR_OK)
        return Result<bool>::Err(rgr_last_error());
    dirty_ = true;
    return Result<bool>::Ok(true);
}

// rmqtt-cluster-raft/src/router.rs:549-566
Result<bool> GpuRouter::restore(const raft::Snapshot& snap) {
    if (!g_) return Result<bool>::Err(create_error_);
    for (auto& r : snap.relations)
        if (!valid_topic(r.topic_filter)) return Result<bool>::Err("invalid topic filter `" + r.topic_filter + "`");   // router.rs:559
    if (snap.relations.size() >= RGR_ID_NONE) return Result<bool>::Err("snapshot holds more relations than sub ids");
    std::unique_lock<TableMutex> g(mu_);
    // relations.clear() (router.rs:557): a fresh device table takes the place of the old one
    rgr_group* fresh = nullptr;
    rgr_config cfg{};
    if (rgr_group_create(&cfg, devices_.data(), uint32_t(devices_.size()), &fresh) != RGR_OK) return Result<bool>::Err(rgr_last_error());
    std::unordered_map<TopicFilter, std::shared_ptr<FilterEntry>> relations;
    for (auto& r : snap.relations) {                                                                        // HashMap::insert: the later entry wins
        auto& e = relations[r.topic_filter];
        if (!e) e = std::make_shared<FilterEntry>(FilterEntry{r.topic_filter, {}});
        e->rels[r.client_id] = Rel{r.id, r.opts, 0, 0};
    }
    std::vector<Slot> slab;
    OwnerIndex owners;
    Dense clients;
    std::string blob;
    std::vector<uint64_t> offs{0};
    std::vector<uint32_t> sub_ids, owner_ids, client_idx;
    std::vector<uint8_t> qos, flags;
    for (auto& kv : relations)
        for (auto& rel : kv.second->rels) {
            Rel& x = rel.second;
            x.sub_id = uint32_t(slab.size());
            x.owner_id = owners.acquire(x.id);
            slab.push_back(Slot{&kv.second->filter, &x, kv.second});
            blob += kv.first;
            offs.push_back(blob.size());
            sub_ids.push_back(x.sub_id);
            owner_ids.push_back(x.owner_id);
            client_idx.push_back(clients.acquire(client_key(x.id.node_id, x.id.client_id)));
            qos.push_back(x.opts.qos);
            flags.push_back(flags_of(x.opts));
        }
    uint64_t rejected = 0;
    const uint64_t n = sub_ids.size();
    int32_t rc = rgr_group_subscribe_bulk(fresh, reinterpret_cast<const uint8_t*>(blob.data()), offs.data(), n, sub_ids.data(), qos.data(), flags.data(),
                                          &rejected);
    if (rc == RGR_OK && rejected != 0) { rgr_group_destroy(fresh); return Result<bool>::Err(std::to_string(rejected) + " filters of the snapshot were rejected"); }
    if (rc == RGR_OK) rc = rgr_group_sub_attrs_bulk(fresh, sub_ids.data(), owner_ids.data(), client_idx.data(), n);
    if (rc != RGR_OK) { std::string e = rgr_last_error(); rgr_group_destroy(fresh); return Result<bool>::Err(e); }
    rgr_group_destroy(g_);
    g_ = fresh;
    relations_ = std::move(relations);      // node-based map: the slab's pointers into it stay valid
    slab_ = std::move(slab);
    free_sub_ids_.clear();
    limbo_[0].clear(); limbo_[1].clear();
    restore_epoch_++;                            // every id was renumbered: delivery passes in flight are stale
    owners_ = std::move(owners);
    for (auto& e : owner_bucket_epoch_) e.fetch_add(1, std::memory_order_acq_rel);      // (a restore replaces every owner id)
    clients_ = std::move(clients);
    bulk_loaded_ = true;
    shared_rels_ = 0;
    for (auto& kv : relations_) for (auto& rel : kv.second->rels) shared_rels_ += rel.second.opts.shared_group ? 1 : 0;
    topics_count_ = Counter{snap.topics_count.count, snap.topics_count.max};             // router.rs:555
    relations_count_ = Counter{snap.relations_count.count, snap.relations_count.max};   // router.rs:568
    dirty_ = true;
    return Result<bool>::Ok(true);
}

// router.rs:456-496
Result<bool> GpuRouter::remove(const std::string& topic_filter, const Id& id) {
    if (!g_) return Result<bool>::Err(create_error_);
    std::unique_lock<TableMutex> g(mu_);
    auto it = relations_.find(topic_filter);
    if (it == relations_.end()) return Result<bool>::Ok(false);
    auto& rels = it->second->rels;
    auto r = rels.find(id.client_id);
    if (r == rels.end() || r->second.id != id) return Result<bool>::Ok(false);   // router.rs:460-467
    const uint32_t sub_id = r->second.sub_id;
    const bool last = rels.size() == 1;                              // router.rs:484-490: the filter leaves the trie with its last relation
    if (rgr_group_unsubscribe(g_, topic_filter.data(), uint32_t(topic_filter.size()), sub_id, last ? 1 : 0) != RGR_OK)
        return Result<bool>::Err(rgr_last_error());
    if (r->second.opts.shared_group) shared_rels_--;
    slab_[sub_id].rel = nullptr;                      // (the slot keeps its filter's entry until it is handed out again: gpu_router.hpp Slot)
    limbo_[pass_generation_ & 1u].push_back(sub_id);  // reusable when the device table has dropped it and no delivery pass that may hold it lives (limbo_)
    bool owner_left = false;
    owners_.release(r->second.id, &owner_left);
    if (owner_left) bump_owner_bucket(r->second.id);
    clients_.release(client_key(r->second.id.node_id, r->second.id.client_id));
    rels.erase(r);
    relations_count_.dec();
    if (last) {
        relations_.erase(it);
        topics_count_.dec();
    }
    dirty_ = true;
    return Result<bool>::Ok(true);
}

namespace {
// SubscriptioRelationsCollector (types.rs:503-541) of one node
struct Collector {
    SubRelations v3;
    SubRelations v5;                                       // rows in first-hit order
    std::unordered_map<ClientId, size_t> v5_index;         // the HashMap<ClientId, ...> of types.rs:506
    // returns whether the hit created the client's v5 entry (true) or only contributed its identifier (false); v3: true
    bool add(const TopicFilter& filter, const ClientId& client, const SubscriptionOptions& opts, std::optional<SharedGroupType> group) {
        if (opts.is_v3()) { v3.push_back(SubRelation{filter, client, opts, std::nullopt, std::move(group)}); return true; }   // types.rs:519-521
        auto it = v5_index.find(client);
        if (it != v5_index.end()) {                                                                          // types.rs:526-534
            if (opts.subscription_identifier) {
                auto& ids = v5[it->second].sub_ids;
                if (ids) ids->push_back(opts.subscription_identifier); else ids = std::vector<uint32_t>{opts.subscription_identifier};
            }
            return false;
        }
        SubRelation r{filter, client, opts, std::nullopt, std::move(group)};                                 // types.rs:535-538
        if (opts.subscription_identifier) r.sub_ids = std::vector<uint32_t>{opts.subscription_identifier};
        v5_index.emplace(client, v5.size());
        v5.push_back(std::move(r));
        return true;
    }
};
}  // namespace

// ---- Filters path: device walk -> one sub id per matched filter -> host expansion from relations_ (router.rs:194-231)
Result<bool> GpuRouter::filters_pass(const std::vector<TopicName>& topics, FilterPass& pass) {
    std::string blob;
    std::vector<uint64_t> offs(topics.size() + 1, 0);
    for (size_t i = 0; i < topics.size(); ++i) { blob += topics[i]; offs[i + 1] = blob.size(); }
    return filters_pass(blob, offs, pass);
}

Result<bool> GpuRouter::filters_pass(const std::string& blob, const std::vector<uint64_t>& offs, FilterPass& pass) {
    if (!g_) return Result<bool>::Err(create_error_);
    // The device pass runs under the SHARED lock: several passes (the Batcher's drivers) walk the same epoch at once, add / remove wait
    // for them — a fraction of a millisecond, as they wait for the trie's write lock in the reference, router.rs:438.  Pending
    // changes are committed first, under the exclusive lock; a writer slipping in between the two locks sends us round again, and
    // after a few rounds the pass simply runs under the exclusive lock (no live-lock under subscribe churn: r3c).
    for (int attempt = 0; attempt < 3; ++attempt) {
        {
            std::shared_lock<TableMutex> sh(mu_);
            if (!dirty_) return device_pass(blob, offs, pass);
        }
        std::unique_lock<TableMutex> x(mu_);
        if (commit_if_dirty() != RGR_OK) return Result<bool>::Err(rgr_last_error());
    }
    std::unique_lock<TableMutex> x(mu_);
    return filters_pass_locked(blob, offs, pass);
}

// caller holds mu_ (shared or exclusive) and the table is committed
Result<bool> GpuRouter::device_pass(const std::string& blob, const std::vector<uint64_t>& offs, FilterPass& pass) {
    pass.epoch = restore_epoch_.load(std::memory_order_acquire);
    if (!pass.lease_of) { pass.lease_of = this; pass.lease_parity = unsigned(pass_generation_ & 1u); live_passes_[pass.lease_parity].fetch_add(1, std::memory_order_acq_rel); }
    rgr_filters_result_free(&pass.res);
    if (rgr_group_match_filter_subs(g_, reinterpret_cast<const uint8_t*>(blob.data()), offs.data(), uint32_t(offs.size() - 1), &pass.res) != RGR_OK)
        return Result<bool>::Err(rgr_last_error());
    return Result<bool>::Ok(true);
}

// caller holds mu_ exclusively
Result<bool> GpuRouter::filters_pass_locked(const std::string& blob, const std::vector<uint64_t>& offs, FilterPass& pass) {
    if (commit_if_dirty() != RGR_OK) return Result<bool>::Err(rgr_last_error());
    return device_pass(blob, offs, pass);
}

// caller holds mu_ (shared) and the pass is current
std::optional<SubRelationsMap> GpuRouter::expand_locked(const rgr_filters_result& res, size_t t, const Id& id, const TopicName& topic, uint64_t* hits) {
    if (res.status[t] != RGR_TOPIC_OK) return std::nullopt;                  // Topic::from_str Err (router.rs:177)
    std::map<NodeId, Collector> collector_map;                                 // router.rs:176
    uint64_t n_hits = 0;
    for (uint64_t k = res.pair_offsets[t]; k < res.pair_offsets[t + 1]; ++k) {  // matched filters in TopicTree::matches order (router.rs:181)
        const uint32_t rep = res.filter_ids[k];
        if (rep == RGR_ID_NONE || rep >= slab_.size() || !slab_[rep].entry) continue;     // (a filter without relations: router.rs:184 `None`)
        const Slot& s = slab_[rep];
        std::map<std::string, std::vector<SharedCandidate>> groups;           // router.rs:183-192
        for (const auto& kv : s.entry->rels) {                                  // router.rs:194: iteration order unspecified there too
            const Rel& rel = kv.second;
            ++n_hits;
            if (rel.opts.v5 && rel.opts.no_local && rel.id == id) continue;     // router.rs:196-201
            const NodeId node = rel.id.node_id;
            if (rel.opts.shared_group) {                                        // router.rs:204-213
                groups[*rel.opts.shared_group].push_back(SharedCandidate{node, rel.id.client_id, rel.opts, is_online(node, rel.id.client_id)});
                continue;
            }
            collector_map[node].add(*s.filter, rel.id.client_id, rel.opts, std::nullopt);       // router.rs:214-229
        }
        for (auto& gk : groups) {                                               // router.rs:236-255, once per matched filter
            std::vector<ClientId> cids;
            for (auto& c : gk.second) cids.push_back(c.client_id);
            const auto pick = shared_ ? shared_->choice(gk.first, id, topic, gk.second) : std::nullopt;
            if (!pick || pick->first >= gk.second.size()) continue;
            const SharedCandidate& c = gk.second[pick->first];
            collector_map[c.node_id].add(*s.filter, c.client_id, c.opts, SharedGroupType{gk.first, pick->second, cids});
        }
    }
    if (hits) *hits = n_hits;
    SubRelationsMap m;                                                          // router.rs:258-261 + types.rs:488-497
    for (auto& kv : collector_map) {
        auto& dst = m[kv.first];
        dst = std::move(kv.second.v3);
        for (auto& r : kv.second.v5) dst.push_back(std::move(r));
    }
    return m;
}

Result<SubRelationsMap> GpuRouter::expand(const FilterPass& pass, size_t t, const Id& id, const TopicName& topic) {
    {
        std::shared_lock<TableMutex> g(mu_);
        if (pass.epoch == restore_epoch_.load(std::memory_order_acquire)) {
            uint64_t hits = 0;
            auto m = expand_locked(pass.res, t, id, topic, &hits);
            mean_hits_ = 0.9 * std::min(mean_hits_.load(), 1e8) + 0.1 * double(hits);
            if (!m) return Result<SubRelationsMap>::Err("invalid topic `" + topic + "`");
            return Result<SubRelationsMap>::Ok(std::move(*m));
        }
    }
    return rematch(id, topic);
}

// The table lost a relation since the pass ran (its sub ids may have been freed): match this publish again, alone, with the table
// held still from the commit to the end of the expansion — always current, no retry loop.  A device failure is reported as what it
// is, not as an invalid topic (round-3 advisor).
Result<SubRelationsMap> GpuRouter::rematch(const Id& id, const TopicName& topic) {
    stale_expansions_++;
    FilterPass fresh;
    const std::vector<uint64_t> offs{0, topic.size()};
    std::unique_lock<TableMutex> x(mu_);
    auto r = filters_pass_locked(topic, offs, fresh);
    if (!r.ok()) return Result<SubRelationsMap>::Err(r.error);
    auto m = expand_locked(fresh.res, 0, id, topic, nullptr);
    if (!m) return Result<SubRelationsMap>::Err("invalid topic `" + topic + "`");
    return Result<SubRelationsMap>::Ok(std::move(*m));
}

// A run of publishes of ONE pass expanded under one acquisition of the shared lock (the Batcher's workers: a shared_mutex taken
// once per publish by dozens of threads is itself a hot cache line).
void GpuRouter::expand_chunk(const FilterPass& pass, const size_t* index, const Id* const* ids, const TopicName* const* topics, size_t n,
                             std::vector<Result<SubRelationsMap>>& out, bool* stale) {
    out.clear();
    out.reserve(n);
    bool current;
    {
        std::shared_lock<TableMutex> g(mu_);
        current = pass.epoch == restore_epoch_.load(std::memory_order_acquire);
        if (current) {
            uint64_t hits = 0, h = 0;
            for (size_t i = 0; i < n; ++i) {
                auto m = expand_locked(pass.res, index[i], *ids[i], *topics[i], &h);
                hits += h;
                out.push_back(m ? Result<SubRelationsMap>::Ok(std::move(*m)) : Result<SubRelationsMap>::Err("invalid topic `" + *topics[i] + "`"));
            }
            if (n) mean_hits_ = 0.9 * std::min(mean_hits_.load(), 1e8) + 0.1 * double(hits) / double(n);
        }
    }
    if (!current && stale) { *stale = true; return; }            // (the caller sends the run through another batch)
    if (!current) for (size_t i = 0; i < n; ++i) out.push_back(rematch(*ids[i], *topics[i]));
}

Result<bool> GpuRouter::matches_batch(const std::vector<Id>& ids, const std::vector<TopicName>& topics,
                                      std::vector<std::optional<SubRelationsMap>>& out) {
    if (!g_) return Result<bool>::Err(create_error_);
    if (ids.size() != topics.size()) return Result<bool>::Err("matches_batch: ids/topics size mismatch");
    if (mode_ == MatchMode::Deliver || (mode_ == MatchMode::Auto && mean_hits_ < kAutoDeliverBelow)) return matches_batch_deliver(ids, topics, out);
    FilterPass pass;
    auto r = filters_pass(topics, pass);
    if (!r.ok()) return r;
    out.assign(topics.size(), std::nullopt);
    for (size_t t = 0; t < topics.size(); ++t) {
        auto m = expand(pass, t, ids[t], topics[t]);
        if (m.ok()) out[t] = std::move(*m.value);
        else if (m.error.rfind("invalid topic", 0) != 0) return Result<bool>::Err(m.error);      // a device failure fails the call
    }
    return Result<bool>::Ok(true);
}

// ---- Deliver path: 12-byte tuples with the device's delivery words
Result<bool> GpuRouter::matches_batch_deliver(const std::vector<Id>& ids, const std::vector<TopicName>& topics,
                                              std::vector<std::optional<SubRelationsMap>>& out) {
    std::unique_lock<TableMutex> g(mu_);
    if (commit_if_dirty() != RGR_OK) return Result<bool>::Err(rgr_last_error());
    std::string blob;
    std::vector<uint64_t> offs(topics.size() + 1, 0);
    for (size_t i = 0; i < topics.size(); ++i) { blob += topics[i]; offs[i + 1] = blob.size(); }
    // publish attributes: who publishes (for No Local); qos 2 / retain 0 leave the subscription's own qos in the word
    std::vector<rgr_publish_attr> attrs(topics.size());
    for (size_t i = 0; i < topics.size(); ++i) attrs[i] = rgr_publish_attr{owners_.find(ids[i]), 2u};
    rgr_result res{};
    // Without $share members the grouping by node (router.rs:258-261: one collector per node) is done on the device: every
    // topic's tuples arrive partitioned by node index, one slice per collector.  ($share picks are added where their filter's
    // hits end, across nodes — router.rs:236-255 — so tables that hold shared members keep the ungrouped order.)
    if (shared_rels_ == 0 && !bulk_loaded_) {
        rgr_node_groups ng{};
        if (rgr_group_match_batch_deliver_grouped(g_, reinterpret_cast<const uint8_t*>(blob.data()), offs.data(), uint32_t(topics.size()), attrs.data(), &res, &ng) != RGR_OK)
            return Result<bool>::Err(rgr_last_error());
        out.assign(topics.size(), std::nullopt);
        for (size_t t = 0; t < topics.size(); ++t) {
            if (res.status[t] != RGR_TOPIC_OK) continue;
            SubRelationsMap m;
            for (uint64_t gi = ng.group_offsets[t]; gi < ng.group_offsets[t + 1]; ++gi) {
                Collector col;                                           // the node's SubscriptioRelationsCollector
                const uint64_t e = gi + 1 < ng.group_offsets[t + 1] ? ng.group_begin[gi + 1] : res.hit_offsets[t + 1];
                for (uint64_t k = ng.group_begin[gi]; k < e; ++k) {
                    const uint32_t w = res.tuples[k].qos_flags;
                    if (w & RGR_HIT_NO_LOCAL) continue;                  // router.rs:196-201, decided on the device
                    const Slot& sl = slab_[res.tuples[k].sub_id];
                    const Rel& rel = *sl.rel;
                    const bool created = col.add(*sl.filter, rel.id.client_id, rel.opts, std::nullopt);
                    if (!rel.opts.is_v3() && created == ((w & RGR_HIT_V5_DUP) != 0)) flag_mismatches_++;
                }
                if (col.v3.empty() && col.v5.empty()) continue;          // (every hit of the node was the publisher's own No Local subscription)
                auto& dst = m[nodes_[ng.group_node[gi]]];
                dst = std::move(col.v3);
                for (auto& r : col.v5) dst.push_back(std::move(r));
            }
            out[t] = std::move(m);
        }
        if (!topics.empty()) mean_hits_ = 0.9 * std::min(mean_hits_.load(), 1e8) + 0.1 * double(res.n_hits) / double(topics.size());
        rgr_result_free(&res);
        return Result<bool>::Ok(true);
    }
    if (rgr_group_match_batch_deliver(g_, reinterpret_cast<const uint8_t*>(blob.data()), offs.data(), uint32_t(topics.size()), attrs.data(), &res) != RGR_OK)
        return Result<bool>::Err(rgr_last_error());
    out.assign(topics.size(), std::nullopt);
    for (size_t t = 0; t < topics.size(); ++t) {
        if (res.status[t] != RGR_TOPIC_OK) continue;                 // Topic::from_str Err (router.rs:177)
        std::map<NodeId, Collector> collector_map;                   // router.rs:176
        // members of $share groups of the filter whose hits are going by (router.rs:183-192), keyed by group
        std::map<std::string, std::vector<std::pair<SharedCandidate, const Rel*>>> groups;
        const std::string* cur_filter = nullptr;
        bool shared_chosen = false;
        auto flush_groups = [&]() {                                  // router.rs:236-255, once per matched filter
            for (auto& gk : groups) {
                std::vector<SharedCandidate> ncs;
                std::vector<ClientId> cids;
                for (auto& c : gk.second) { ncs.push_back(c.first); cids.push_back(c.first.client_id); }
                const auto pick = shared_ ? shared_->choice(gk.first, ids[t], topics[t], ncs) : std::nullopt;
                if (!pick || pick->first >= ncs.size()) continue;
                const SharedCandidate& c = ncs[pick->first];
                collector_map[c.node_id].add(*cur_filter, c.client_id, c.opts, SharedGroupType{gk.first, pick->second, cids});
                shared_chosen = true;
            }
            groups.clear();
        };
        for (uint64_t k = res.hit_offsets[t]; k < res.hit_offsets[t + 1]; ++k) {
            const uint32_t w = res.tuples[k].qos_flags;
            const Slot& s = slab_[res.tuples[k].sub_id];
            if (s.filter != cur_filter) { flush_groups(); cur_filter = s.filter; }
            if (w & RGR_HIT_NO_LOCAL) continue;                      // router.rs:196-201, decided on the device
            const Rel& rel = *s.rel;
            const NodeId node = bulk_loaded_ ? rel.id.node_id : nodes_[w >> 16];
            if (rel.opts.shared_group) {                              // router.rs:204-213
                groups[*rel.opts.shared_group].push_back({SharedCandidate{node, rel.id.client_id, rel.opts, is_online(node, rel.id.client_id)}, &rel});
                continue;
            }
            const bool created = collector_map[node].add(*s.filter, rel.id.client_id, rel.opts, std::nullopt);   // router.rs:214-229
            // the device's verdict on the same hit (types.rs:524-539 as a min-position-per-client problem)
            if (!rel.opts.is_v3() && !shared_chosen && created == ((w & RGR_HIT_V5_DUP) != 0)) flag_mismatches_++;
        }
        flush_groups();
        SubRelationsMap m;                                           // router.rs:258-261 + types.rs:488-497: v3 rows, then the v5 map's
        for (auto& kv : collector_map) {
            auto& dst = m[kv.first];
            dst = std::move(kv.second.v3);
            for (auto& r : kv.second.v5) dst.push_back(std::move(r));
        }
        out[t] = std::move(m);
    }
    if (!topics.empty()) mean_hits_ = 0.9 * std::min(mean_hits_.load(), 1e8) + 0.1 * double(res.n_hits) / double(topics.size());
    rgr_result_free(&res);
    return Result<bool>::Ok(true);
}

// The delivery stage as a pass of its own: see gpu_router.hpp.  The pass runs under the table's SHARED lock (several passes at once, like
// filters_pass); a dirty table is committed first under the exclusive one.
Result<bool> GpuRouter::deliver_pass(const std::string& blob, const std::vector<uint64_t>& offs, const Id* const* ids, const uint8_t* qos_retain, DeliverPass& pass,
                                     const OwnerHint* hints) {
    if (!g_) return Result<bool>::Err(create_error_);
    const uint32_t n = uint32_t(offs.size() - 1);
    auto run = [&]() -> Result<bool> {
        std::vector<rgr_publish_attr> attrs(n);
        pass.epoch = restore_epoch_.load(std::memory_order_acquire);
        if (!pass.lease_of) { pass.lease_of = this; pass.lease_parity = unsigned(pass_generation_ & 1u); live_passes_[pass.lease_parity].fetch_add(1, std::memory_order_acq_rel); }
        for (uint32_t i = 0; i < n; ++i)
            attrs[i] = rgr_publish_attr{hints && owner_hint_current(hints[i]) ? hints[i].owner : owners_.find(*ids[i]), uint32_t(qos_retain[i] & 7u)};
        if (rgr_group_match_batch_deliver(g_, reinterpret_cast<const uint8_t*>(blob.data()), offs.data(), n, attrs.data(), &pass.res) != RGR_OK)
            return Result<bool>::Err(rgr_last_error());
        return Result<bool>::Ok(true);
    };
    // As filters_pass: the pass itself runs under the SHARED lock; pending changes are committed first under the exclusive one, and a writer slipping in
    // between the two sends us round again (a few times: then the pass runs under the exclusive lock).  (Until r7z a dirty table meant the whole pass
    // under the exclusive lock — under subscribe churn that kept the completion threads out for the length of every other pass.)
    for (int attempt = 0; attempt < 3; ++attempt) {
        {
            std::shared_lock<TableMutex> g(mu_);
            if (!dirty_.load(std::memory_order_acquire)) return run();
        }
        std::unique_lock<TableMutex> g(mu_);
        if (commit_if_dirty() != RGR_OK) return Result<bool>::Err(rgr_last_error());
    }
    std::unique_lock<TableMutex> g(mu_);
    if (commit_if_dirty() != RGR_OK) return Result<bool>::Err(rgr_last_error());
    return run();
}

// router.rs:499-501
Result<SubRelationsMap> GpuRouter::matches(const Id& id, const TopicName& topic) {
    std::vector<std::optional<SubRelationsMap>> out;
    auto r = matches_batch({id}, {topic}, out);
    if (!r.ok()) return Result<SubRelationsMap>::Err(r.error);
    if (!out[0]) return Result<SubRelationsMap>::Err("invalid topic `" + topic + "`");
    return Result<SubRelationsMap>::Ok(std::move(*out[0]));
}

// router.rs:157-170: the matched filters, unique, in iteration order.  Every filter in the trie has at least one
// relation (it leaves with its last one, router.rs:484-490), so the filters of the hits ARE the matched filters.
Result<std::vector<Route>> GpuRouter::get(const std::string& topic) {
    if (!g_) return Result<std::vector<Route>>::Err(create_error_);
    std::unique_lock<TableMutex> g(mu_);
    if (commit_if_dirty() != RGR_OK) return Result<std::vector<Route>>::Err(rgr_last_error());
    const uint64_t offs[2] = {0, topic.size()};
    rgr_result res{};
    if (rgr_group_match_batch(g_, reinterpret_cast<const uint8_t*>(topic.data()), offs, 1, &res) != RGR_OK)
        return Result<std::vector<Route>>::Err(rgr_last_error());
    std::vector<Route> routes;
    const bool bad = res.status[0] != RGR_TOPIC_OK;
    const std::string* last = nullptr;
    for (uint64_t k = 0; !bad && k < res.n_hits; ++k) {
        const std::string* f = slab_[res.tuples[k].sub_id].filter;
        if (f == last) continue;
        last = f;
        if (std::none_of(routes.begin(), routes.end(), [&](const Route& r) { return r.topic == *f; }))   // .unique()
            routes.push_back(Route{this_node_, *f});
    }
    rgr_result_free(&res);
    if (bad) return Result<std::vector<Route>>::Err("invalid topic `" + topic + "`");
    return Result<std::vector<Route>>::Ok(std::move(routes));
}

Result<bool> GpuRouter::has_matches(const std::string& topic) {
    auto r = get(topic);
    if (!r.ok()) return Result<bool>::Err(r.error);
    return Result<bool>::Ok(!r.value->empty());
}

std::vector<Route> GpuRouter::gets(size_t limit) {   // router.rs:514-541: unique (node, filter) pairs
    std::shared_lock<TableMutex> g(mu_);
    std::vector<Route> out;
    for (auto& kv : relations_) {
        std::vector<NodeId> seen;
        for (auto& r : kv.second->rels) {
            if (out.size() >= limit) return out;
            if (std::find(seen.begin(), seen.end(), r.second.id.node_id) != seen.end()) continue;
            seen.push_back(r.second.id.node_id);
            out.push_back(Route{r.second.id.node_id, kv.first});
        }
    }
    return out;
}

size_t GpuRouter::topics_tree() {
    std::shared_lock<TableMutex> g(mu_);
    return relations_.size();      // one trie value per distinct filter (router.rs:571-574)
}

std::vector<std::string> GpuRouter::list_topics(size_t top) {
    std::shared_lock<TableMutex> g(mu_);
    std::vector<std::string> v;
    for (auto& kv : relations_) { if (v.size() >= top) break; v.push_back(kv.first); }
    return v;
}

}  // namespace rmqtt

namespace rmqtt {

// ---------------------------------------------------------------------------------------------- Batcher
namespace {
size_t shard_of_this_thread(size_t n) {
    static std::atomic<size_t> next{0};
    thread_local size_t mine = next.fetch_add(1);
    return mine % n;
}
}  // namespace

Batcher::Batcher(GpuRouter& router, size_t max_batch, std::chrono::microseconds max_delay, unsigned passes_in_flight, unsigned workers)
    : router_(router), max_batch_(max_batch ? max_batch : 1), max_delay_(max_delay) {
    for (unsigned i = 0; i < std::max(1u, passes_in_flight); ++i) drivers_.emplace_back([this] { run(); });
    for (unsigned i = 0; i < workers; ++i) workers_.emplace_back([this] { work(); });
}

Batcher::~Batcher() {
    { std::lock_guard<std::mutex> g(mu_); stop_.store(true); }
    cv_req_.notify_all();
    for (auto& d : drivers_) d.join();
    { std::lock_guard<std::mutex> g(task_mu_); task_stop_ = true; }
    task_cv_.notify_all();
    for (auto& w : workers_) w.join();
    for (Shard& sh : shards_) for (Req* r : sh.free) delete r;
}

void Batcher::enqueue(Req* req) {
    Shard& sh = shards_[(req->cb || req->dcb) ? req->shard : shard_of_this_thread(kShards)];
    { std::lock_guard<std::mutex> lk(sh.m); sh.q.push_back(req); ++sh.requests; }      // (counted per shard, under the lock already held: one shared atomic less per publish)
    const size_t before = pending_.fetch_add(1, std::memory_order_seq_cst);
    // wake a driver for the first request of a batch and when the batch is full; everything in between rides on its deadline.  Only
    // when a driver is actually asleep: under load the drivers find work without sleeping and a submit stays a queue push (r4b: a
    // futex wake per 0 -> 1 transition, with batches draining as fast as they formed, held 8 submitters at 1 M publishes/s).
    if ((before == 0 || before + 1 == max_batch_) && sleepers_.load(std::memory_order_seq_cst) > 0) { { std::lock_guard<std::mutex> g(mu_); } cv_req_.notify_one(); }
}

Result<SubRelationsMap> Batcher::matches(const Id& id, const TopicName& topic) {
    Req req;
    req.id = id; req.topic = topic;
    if (stop_.load(std::memory_order_acquire)) return Result<SubRelationsMap>::Err("batcher stopped");
    enqueue(&req);
    {
        std::unique_lock<std::mutex> lk(req.m);
        req.cv.wait(lk, [&] { return req.done; });
    }
    if (!req.err.empty()) return Result<SubRelationsMap>::Err(req.err);
    // the expansion (router.rs:194-261) runs HERE, on the caller's thread: N callers expand in parallel, like N tokio workers
    return router_.expand(*req.pass, req.index, id, topic);
}

void Batcher::submit(const Id& id, std::string_view topic, Callback cb, void* user, uint64_t tag) {
    if (stop_.load(std::memory_order_acquire)) { cb(user, tag, Result<SubRelationsMap>::Err("batcher stopped")); return; }
    const uint32_t shard = uint32_t(shard_of_this_thread(kShards));
    Shard& sh = shards_[shard];
    Req* req = nullptr;
    { std::lock_guard<std::mutex> lk(sh.m); if (!sh.free.empty()) { req = sh.free.back(); sh.free.pop_back(); } }
    if (!req) req = new Req;
    req->id = id; req->topic.assign(topic.data(), topic.size());      // (recycled objects: the strings' capacity is reused)
    req->cb = cb; req->user = user; req->tag = tag; req->shard = shard;
    req->pass.reset(); req->err.clear(); req->done = false; req->tries = 0;
    enqueue(req);
}

void Batcher::submit_deliver(const Id& from, std::string_view topic, uint8_t qos_retain, DeliverCallback cb, void* user, uint64_t tag, const GpuRouter::OwnerHint* hint) {
    if (stop_.load(std::memory_order_acquire)) { const std::string e = "batcher stopped"; cb(user, tag, nullptr, 0, from, &e); return; }
    const uint32_t shard = uint32_t(shard_of_this_thread(kShards));
    Shard& sh = shards_[shard];
    Req* req = nullptr;
    { std::lock_guard<std::mutex> lk(sh.m); if (!sh.free.empty()) { req = sh.free.back(); sh.free.pop_back(); } }
    if (!req) req = new Req;
    req->id = from; req->topic.assign(topic.data(), topic.size());
    req->owner = hint && router_.owner_hint_current(*hint) ? *hint : router_.owner_hint(from);
    req->cb = nullptr; req->dcb = cb; req->qos_retain = qos_retain; req->user = user; req->tag = tag; req->shard = shard;
    req->pass.reset(); req->err.clear(); req->done = false; req->tries = 0;
    enqueue(req);
}

// finished asynchronous requests of one task go back to their shards' free lists, one lock per shard touched
void Batcher::recycle(std::vector<Req*>& reqs) {
    for (size_t i = 0; i < reqs.size();) {
        const uint32_t shard = reqs[i]->shard;
        std::lock_guard<std::mutex> lk(shards_[shard].m);
        auto& fl = shards_[shard].free;
        for (; i < reqs.size() && reqs[i]->shard == shard; ++i) { if (fl.size() < kFreeMax) fl.push_back(reqs[i]); else delete reqs[i]; }
    }
    reqs.clear();
}

void Batcher::run() {
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(mu_);
            sleepers_.fetch_add(1, std::memory_order_seq_cst);
            cv_req_.wait(lk, [&] { return stop_.load() || pending_.load(std::memory_order_seq_cst) > 0; });
            if (pending_.load(std::memory_order_acquire) == 0) { sleepers_.fetch_sub(1); if (stop_.load()) return; continue; }
            // deadline counted from (about) the first request of the batch
            const auto deadline = std::chrono::steady_clock::now() + max_delay_;
            cv_req_.wait_until(lk, deadline, [&] { return stop_.load() || pending_.load(std::memory_order_seq_cst) >= max_batch_; });
            sleepers_.fetch_sub(1, std::memory_order_seq_cst);
        }
        const auto t_collect = std::chrono::steady_clock::now();
        std::vector<Req*> reqs;
        // the shards are visited round-robin from a rotating start and each gives an equal share first, so that a batch that cannot
        // take everything does not starve the submitters on the later shards (r4c: always starting at shard 0 left a 45-90 ms p99)
        const size_t first = next_shard_.fetch_add(1, std::memory_order_relaxed);
        for (int round = 0; round < 2 && reqs.size() < max_batch_; ++round)
            for (size_t k = 0; k < kShards && reqs.size() < max_batch_; ++k) {
                Shard& sh = shards_[(first + k) % kShards];
                std::lock_guard<std::mutex> lk(sh.m);
                const size_t share = round == 0 ? std::max<size_t>(1, max_batch_ / kShards) : max_batch_;
                const size_t take = std::min({sh.q.size(), max_batch_ - reqs.size(), share});
                reqs.insert(reqs.end(), sh.q.begin(), sh.q.begin() + take);
                sh.q.erase(sh.q.begin(), sh.q.begin() + take);
            }
        if (reqs.empty()) continue;                               // another driver took them
        if (pending_.fetch_sub(reqs.size(), std::memory_order_acq_rel) > reqs.size() && sleepers_.load() > 0) cv_req_.notify_one();     // more are waiting: next driver
        std::string blob;
        std::vector<uint64_t> offs(reqs.size() + 1, 0);
        {
            size_t bytes = 0;
            for (Req* r : reqs) bytes += r->topic.size();
            blob.reserve(bytes);
            for (size_t i = 0; i < reqs.size(); ++i) { blob += reqs[i]->topic; offs[i + 1] = blob.size(); }
        }
        auto pass = std::make_shared<GpuRouter::FilterPass>();
        std::shared_ptr<GpuRouter::DeliverPass> dpass;
        const bool deliver = reqs[0]->dcb != nullptr;            // (a batcher serves one kind of request)
        const auto t_pass = std::chrono::steady_clock::now();
        Result<bool> res = Result<bool>::Ok(true);
        if (deliver) {
            dpass = std::make_shared<GpuRouter::DeliverPass>();
            std::vector<const Id*> ids(reqs.size());
            std::vector<uint8_t> qr(reqs.size());
            std::vector<GpuRouter::OwnerHint> hints(reqs.size());
            for (size_t i = 0; i < reqs.size(); ++i) { ids[i] = &reqs[i]->id; qr[i] = reqs[i]->qos_retain; hints[i] = reqs[i]->owner; }
            res = router_.deliver_pass(blob, offs, ids.data(), qr.data(), *dpass, hints.data());
        } else res = router_.filters_pass(blob, offs, *pass);
        const auto t_done = std::chrono::steady_clock::now();
        passes_.fetch_add(1, std::memory_order_relaxed);
        collect_ns_.fetch_add(uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(t_pass - t_collect).count()), std::memory_order_relaxed);
        pass_ns_.fetch_add(uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(t_done - t_pass).count()), std::memory_order_relaxed);
        // asynchronous requests go to the workers in runs of kTaskRun (one shared-lock acquisition per run); blocking callers are woken
        Task task;
        std::vector<Req*> failed;
        auto flush = [&] {
            if (task.reqs.empty()) return;
            task.pass = pass; task.dpass = dpass;
            if (workers_.empty()) run_task(task);                 // no pool configured: complete on the driver
            else { { std::lock_guard<std::mutex> g(task_mu_); tasks_.push_back(std::move(task)); } task_cv_.notify_one(); }
            task = Task{};
        };
        for (size_t i = 0; i < reqs.size(); ++i) {
            Req* r = reqs[i];
            if (r->dcb) {
                if (!res.ok()) { r->dcb(r->user, r->tag, nullptr, 0, r->id, &res.error); failed.push_back(r); continue; }
                r->index = i;
                task.reqs.push_back(r);
                if (task.reqs.size() >= kTaskRun) flush();
                continue;
            }
            if (r->cb) {
                if (!res.ok()) { r->cb(r->user, r->tag, Result<SubRelationsMap>::Err(res.error)); failed.push_back(r); continue; }
                r->index = i;
                task.reqs.push_back(r);
                if (task.reqs.size() >= kTaskRun) flush();
                continue;
            }
            std::lock_guard<std::mutex> g(r->m);
            if (!res.ok()) r->err = res.error; else { r->pass = pass; r->index = i; }
            r->done = true;
            r->cv.notify_one();             // under r->m: the caller cannot destroy the request before this returns
        }
        flush();
        if (!failed.empty()) recycle(failed);
        dispatch_ns_.fetch_add(uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_done).count()), std::memory_order_relaxed);
    }
}

void Batcher::run_task(Task& t) {
    const auto t0 = std::chrono::steady_clock::now();
    struct Timed { Batcher* b; std::chrono::steady_clock::time_point t0; ~Timed() {
        b->task_ns_.fetch_add(uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count()), std::memory_order_relaxed);
        b->tasks_run_.fetch_add(1, std::memory_order_relaxed); } } timed{this, t0};
    const size_t n = t.reqs.size();
    if (t.dpass) {                                                // Shared::forwards requests: the completion consumes the delivery words itself
        GpuRouter::SharedHold hold(router_);                      // (one shared-lock acquisition per run, as for the expansions below)
        for (size_t i = 0; i < n; ++i) t.reqs[i]->dcb(t.reqs[i]->user, t.reqs[i]->tag, t.dpass, t.reqs[i]->index, t.reqs[i]->id, nullptr);
        recycle(t.reqs);
        return;
    }
    std::vector<size_t> index(n);
    std::vector<const Id*> ids(n);
    std::vector<const TopicName*> topics(n);
    for (size_t i = 0; i < n; ++i) { index[i] = t.reqs[i]->index; ids[i] = &t.reqs[i]->id; topics[i] = &t.reqs[i]->topic; }
    std::vector<Result<SubRelationsMap>> out;
    bool stale = false;
    router_.expand_chunk(*t.pass, index.data(), ids.data(), topics.data(), n, out, &stale);
    if (stale) {
        // a removal overtook the pass: the run joins another batch (up to kMaxRequeues times per publish, then the one-publish re-match) — re-matching
        // each publish on its own, a device pass of one, is what made the rate collapse under unsubscribe churn (profiles/r07x_*)
        std::vector<Req*> done;
        for (Req* r : t.reqs) {
            if (r->tries < kMaxRequeues && !stop_.load(std::memory_order_acquire)) { r->tries++; r->pass.reset(); requeued_.fetch_add(1, std::memory_order_relaxed); enqueue(r); }
            else { r->cb(r->user, r->tag, router_.rematch_public(r->id, r->topic)); done.push_back(r); }
        }
        if (!done.empty()) recycle(done);
        t.reqs.clear();
        return;
    }
    for (size_t i = 0; i < n; ++i) t.reqs[i]->cb(t.reqs[i]->user, t.reqs[i]->tag, std::move(out[i]));
    recycle(t.reqs);
}

void Batcher::work() {
    for (;;) {
        Task t;
        {
            std::unique_lock<std::mutex> lk(task_mu_);
            task_cv_.wait(lk, [&] { return task_stop_ || !tasks_.empty(); });
            if (tasks_.empty()) return;                           // (stop: the drivers are gone, nothing more can arrive)
            if (tasks_.size() > max_task_queue_) max_task_queue_ = tasks_.size();
            t = std::move(tasks_.front());
            tasks_.pop_front();
        }
        run_task(t);
    }
}

}  // namespace rmqtt
