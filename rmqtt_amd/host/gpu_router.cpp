// GpuRouter — see gpu_router.hpp.  Uses nothing but the C ABI of rmqtt_gpu_router.h.
#include "gpu_router.hpp"

#include <algorithm>

namespace rmqtt {

namespace {
uint8_t flags_of(const SubscriptionOptions& o) {
    return uint8_t((o.v5 ? RGR_SUB_V5 : 0) | (o.v5 && o.no_local ? RGR_SUB_NO_LOCAL : 0) | (o.v5 && o.retain_as_published ? RGR_SUB_RAP : 0));
}
// Every field Id equality looks at (types.rs:1841-1851), unambiguously joined.
std::string id_key(const Id& id) {
    std::string k = std::to_string(id.node_id) + '|' + std::to_string(id.lid) + '|' + std::to_string(id.create_time);
    for (const std::string* f : {&id.local_addr, &id.remote_addr, &id.client_id, &id.username}) { k += '|'; k += std::to_string(f->size()); k += ':'; k += *f; }
    return k;
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
    const uint32_t owner_id = owners_.acquire(id_key(id));
    if (old == rels.end()) {
        relations_count_.inc();
        if (!free_sub_ids_.empty()) { sub_id = free_sub_ids_.back(); free_sub_ids_.pop_back(); }
        else { sub_id = uint32_t(slab_.size()); slab_.emplace_back(); }
        old = rels.emplace(id.client_id, Rel{id, opts, sub_id, owner_id}).first;
    } else {
        sub_id = old->second.sub_id;
        owners_.release(id_key(old->second.id));
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
    slab_[sub_id] = Slot{&it->first, &old->second};
    rc = rgr_sub_add_ex(h_, fid, sub_id, opts.qos, flags_of(opts), ni->second, owner_id, client_idx);
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
    owners_.release(id_key(r->second.id));
    clients_.release(client_key(r->second.id.node_id, r->second.id.client_id));
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
    // publish attributes: who publishes (for No Local); qos 2 / retain 0 leave the subscription's own qos in the word
    std::vector<rgr_publish_attr> attrs(topics.size());
    for (size_t i = 0; i < topics.size(); ++i) attrs[i] = rgr_publish_attr{owners_.find(id_key(ids[i])), 2u};
    rgr_result res{};
    if (rgr_match_batch_deliver(h_, reinterpret_cast<const uint8_t*>(blob.data()), offs.data(), uint32_t(topics.size()), attrs.data(), &res) != RGR_OK)
        return Result<bool>::Err(rgr_last_error());
    out.assign(topics.size(), std::nullopt);
    for (size_t t = 0; t < topics.size(); ++t) {
        if (res.status[t] != RGR_TOPIC_OK) continue;                 // Topic::from_str Err (router.rs:177)
        SubRelationsMap m;                                           // collector_map (router.rs:176) + router.rs:258-261
        std::map<NodeId, SubRelations> v5;                           // types.rs:488-497: v3 rows first, then the v5 map's rows
        for (uint64_t k = res.hit_offsets[t]; k < res.hit_offsets[t + 1]; ++k) {
            const uint32_t w = res.tuples[k].qos_flags;
            if (w & RGR_HIT_NO_LOCAL) continue;                      // router.rs:196-201, decided on the device
            const Slot& s = slab_[res.tuples[k].sub_id];
            const Rel& rel = *s.rel;
            const NodeId node = nodes_[w >> 16];
            if (rel.opts.is_v3()) { m[node].push_back(SubRelation{*s.filter, rel.id.client_id, rel.opts, std::nullopt}); continue; }
            auto& rows = v5[node];
            m[node];                                                  // the node has a collector even if only v5 rows follow
            if (w & RGR_HIT_V5_DUP) {                                // types.rs:526-534: only the subscription identifier is kept
                if (!rel.opts.subscription_identifier) continue;
                for (auto& r : rows) {
                    if (r.client_id != rel.id.client_id) continue;
                    if (r.sub_ids) r.sub_ids->push_back(rel.opts.subscription_identifier);
                    else r.sub_ids = std::vector<uint32_t>{rel.opts.subscription_identifier};
                    break;
                }
            } else {                                                 // types.rs:535-538
                SubRelation r{*s.filter, rel.id.client_id, rel.opts, std::nullopt};
                if (rel.opts.subscription_identifier) r.sub_ids = std::vector<uint32_t>{rel.opts.subscription_identifier};
                rows.push_back(std::move(r));
            }
        }
        for (auto& kv : v5) { auto& dst = m[kv.first]; for (auto& r : kv.second) dst.push_back(std::move(r)); }
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
