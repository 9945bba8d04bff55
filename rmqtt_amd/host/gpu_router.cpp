// GpuRouter — see gpu_router.hpp.  Uses nothing but the C ABI of rmqtt_gpu_router.h.
#include "gpu_router.hpp"

#include <algorithm>

namespace rmqtt {

namespace {
// types.rs:503-541
struct Collector {
    SubRelations v3_rels;
    std::vector<ClientId> v5_order;
    std::unordered_map<ClientId, SubRelation> v5_rels;
    void add(const TopicFilter& filter, const ClientId& client, const SubscriptionOptions& opts) {
        if (opts.is_v3()) { v3_rels.push_back(SubRelation{filter, client, opts, std::nullopt}); return; }
        auto it = v5_rels.find(client);
        if (it != v5_rels.end()) {                                  // types.rs:526-534
            if (opts.subscription_identifier) {
                if (it->second.sub_ids) it->second.sub_ids->push_back(opts.subscription_identifier);
                else it->second.sub_ids = std::vector<uint32_t>{opts.subscription_identifier};
            }
        } else {                                                    // types.rs:535-538
            SubRelation r{filter, client, opts, std::nullopt};
            if (opts.subscription_identifier) r.sub_ids = std::vector<uint32_t>{opts.subscription_identifier};
            v5_rels.emplace(client, std::move(r));
            v5_order.push_back(client);
        }
    }
};
uint8_t flags_of(const SubscriptionOptions& o) { return uint8_t((o.v5 ? RGR_SUB_V5 : 0) | (o.no_local ? RGR_SUB_NO_LOCAL : 0)); }
}  // namespace

GpuRouter::GpuRouter(NodeId this_node, int device) : this_node_(this_node) {
    rgr_config cfg{};
    cfg.device = device;
    if (rgr_create(&cfg, &h_) != RGR_OK) { h_ = nullptr; create_error_ = rgr_last_error(); }
}

GpuRouter::~GpuRouter() { if (h_) rgr_destroy(h_); }

int32_t GpuRouter::commit_if_dirty() {
    if (!dirty_) return RGR_OK;
    int32_t rc = rgr_commit(h_);
    if (rc == RGR_OK) dirty_ = false;
    return rc;
}

// router.rs:434-453
Result<bool> GpuRouter::add(const std::string& topic_filter, const Id& id, const SubscriptionOptions& opts) {
    if (!h_) return Result<bool>::Err(create_error_);
    std::lock_guard<std::mutex> g(mu_);
    uint32_t fid = 0;
    int32_t rc = rgr_filter_add(h_, topic_filter.data(), uint32_t(topic_filter.size()), &fid);   // Topic::from_str + trie insert
    if (rc != RGR_OK) return Result<bool>::Err(std::string("invalid topic filter: ") + rgr_last_error());
    auto it = relations_.find(topic_filter);
    if (it == relations_.end()) {
        topics_count_.inc();
        it = relations_.emplace(topic_filter, FilterEntry{fid, {}}).first;
        filter_names_[fid] = &it->first;
    }
    auto& rels = it->second.rels;
    auto old = rels.find(id.client_id);
    uint32_t sub_id;
    if (old == rels.end()) {
        relations_count_.inc();
        if (!free_sub_ids_.empty()) { sub_id = free_sub_ids_.back(); free_sub_ids_.pop_back(); }
        else { sub_id = uint32_t(slab_.size()); slab_.emplace_back(); }
        old = rels.emplace(id.client_id, Rel{id, opts, sub_id}).first;
    } else {
        sub_id = old->second.sub_id;
        old->second = Rel{id, opts, sub_id};                        // HashMap::insert replaces (router.rs:447)
    }
    slab_[sub_id] = Slot{&it->first, &old->second};
    rc = rgr_sub_add(h_, fid, sub_id, opts.qos, flags_of(opts));
    if (rc != RGR_OK) return Result<bool>::Err(rgr_last_error());
    dirty_ = true;
    return Result<bool>::Ok(true);
}

// router.rs:456-496
Result<bool> GpuRouter::remove(const std::string& topic_filter, const Id& id) {
    if (!h_) return Result<bool>::Err(create_error_);
    std::lock_guard<std::mutex> g(mu_);
    auto it = relations_.find(topic_filter);
    if (it == relations_.end()) return Result<bool>::Ok(false);
    auto& rels = it->second.rels;
    auto r = rels.find(id.client_id);
    if (r == rels.end() || r->second.id != id) return Result<bool>::Ok(false);   // router.rs:460-467
    const uint32_t sub_id = r->second.sub_id, fid = it->second.filter_id;
    if (rgr_sub_remove(h_, fid, sub_id) != RGR_OK) return Result<bool>::Err(rgr_last_error());
    slab_[sub_id] = Slot{};
    free_sub_ids_.push_back(sub_id);
    rels.erase(r);
    relations_count_.dec();
    if (rels.empty()) {                                              // router.rs:484-490
        if (rgr_filter_remove(h_, fid) != RGR_OK) return Result<bool>::Err(rgr_last_error());
        filter_names_.erase(fid);
        relations_.erase(it);
        topics_count_.dec();
    }
    dirty_ = true;
    return Result<bool>::Ok(true);
}

Result<bool> GpuRouter::matches_batch(const std::vector<Id>& ids, const std::vector<TopicName>& topics,
                                      std::vector<std::optional<SubRelationsMap>>& out) {
    if (!h_) return Result<bool>::Err(create_error_);
    if (ids.size() != topics.size()) return Result<bool>::Err("matches_batch: ids/topics size mismatch");
    std::lock_guard<std::mutex> g(mu_);
    if (commit_if_dirty() != RGR_OK) return Result<bool>::Err(rgr_last_error());
    std::string blob;
    std::vector<uint64_t> offs(topics.size() + 1, 0);
    for (size_t i = 0; i < topics.size(); ++i) { blob += topics[i]; offs[i + 1] = blob.size(); }
    rgr_result res{};
    if (rgr_match_batch(h_, reinterpret_cast<const uint8_t*>(blob.data()), offs.data(), uint32_t(topics.size()), &res) != RGR_OK)
        return Result<bool>::Err(rgr_last_error());
    out.assign(topics.size(), std::nullopt);
    for (size_t t = 0; t < topics.size(); ++t) {
        if (res.status[t] != RGR_TOPIC_OK) continue;                 // Topic::from_str Err (router.rs:177)
        std::map<NodeId, Collector> collector_map;
        for (uint64_t k = res.hit_offsets[t]; k < res.hit_offsets[t + 1]; ++k) {
            const Slot& s = slab_[res.tuples[k].sub_id];
            const Rel& rel = *s.rel;
            auto nl = rel.opts.opt_no_local();
            if (nl && *nl && ids[t] == rel.id) continue;             // router.rs:196-201
            collector_map[rel.id.node_id].add(*s.filter, rel.id.client_id, rel.opts);
        }
        SubRelationsMap m;
        for (auto& kv : collector_map) {                             // router.rs:258-261 + types.rs:488-497
            auto& dst = m[kv.first];
            dst = std::move(kv.second.v3_rels);
            for (auto& c : kv.second.v5_order) dst.push_back(std::move(kv.second.v5_rels[c]));
        }
        out[t] = std::move(m);
    }
    rgr_result_free(&res);
    return Result<bool>::Ok(true);
}

// router.rs:499-501
Result<SubRelationsMap> GpuRouter::matches(const Id& id, const TopicName& topic) {
    std::vector<std::optional<SubRelationsMap>> out;
    auto r = matches_batch({id}, {topic}, out);
    if (!r.ok()) return Result<SubRelationsMap>::Err(r.error);
    if (!out[0]) return Result<SubRelationsMap>::Err("invalid topic `" + topic + "`");
    return Result<SubRelationsMap>::Ok(std::move(*out[0]));
}

// router.rs:157-170
Result<std::vector<Route>> GpuRouter::get(const std::string& topic) {
    if (!h_) return Result<std::vector<Route>>::Err(create_error_);
    std::lock_guard<std::mutex> g(mu_);
    if (commit_if_dirty() != RGR_OK) return Result<std::vector<Route>>::Err(rgr_last_error());
    const uint64_t offs[2] = {0, topic.size()};
    rgr_filters_result res{};
    if (rgr_match_filters(h_, reinterpret_cast<const uint8_t*>(topic.data()), offs, 1, &res) != RGR_OK)
        return Result<std::vector<Route>>::Err(rgr_last_error());
    std::vector<Route> routes;
    const bool bad = res.status[0] != RGR_TOPIC_OK;
    for (uint64_t k = 0; !bad && k < res.n_pairs; ++k) {
        const std::string& f = *filter_names_.at(res.filter_ids[k]);
        if (std::none_of(routes.begin(), routes.end(), [&](const Route& r) { return r.topic == f; }))   // .unique()
            routes.push_back(Route{this_node_, f});
    }
    rgr_filters_result_free(&res);
    if (bad) return Result<std::vector<Route>>::Err("invalid topic `" + topic + "`");
    return Result<std::vector<Route>>::Ok(std::move(routes));
}

Result<bool> GpuRouter::has_matches(const std::string& topic) {
    auto r = get(topic);
    if (!r.ok()) return Result<bool>::Err(r.error);
    return Result<bool>::Ok(!r.value->empty());
}

std::vector<Route> GpuRouter::gets(size_t limit) {   // router.rs:514-541: unique (node, filter) pairs
    std::lock_guard<std::mutex> g(mu_);
    std::vector<Route> out;
    for (auto& kv : relations_) {
        std::vector<NodeId> seen;
        for (auto& r : kv.second.rels) {
            if (out.size() >= limit) return out;
            if (std::find(seen.begin(), seen.end(), r.second.id.node_id) != seen.end()) continue;
            seen.push_back(r.second.id.node_id);
            out.push_back(Route{r.second.id.node_id, kv.first});
        }
    }
    return out;
}

size_t GpuRouter::topics_tree() {
    std::lock_guard<std::mutex> g(mu_);
    return relations_.size();      // one trie value per distinct filter (router.rs:571-574)
}

std::vector<std::string> GpuRouter::list_topics(size_t top) {
    std::lock_guard<std::mutex> g(mu_);
    std::vector<std::string> v;
    for (auto& kv : relations_) { if (v.size() >= top) break; v.push_back(kv.first); }
    return v;
}

}  // namespace rmqtt
