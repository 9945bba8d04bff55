// GpuRetainStorage — see gpu_retain.hpp.  Uses nothing but the C ABI.
#include "gpu_retain.hpp"

#include <algorithm>

namespace rmqtt {

namespace {
// The dense answer of one filter (rgr_retain_match_ranges): every live topic id of its ranges, in hit order.  What came over PCIe
// is 16 bytes per RANGE; the ids are read from the library's host mirror of the value arrays.
template <class Fn> void for_each_hit(const rgr_retain_ranges& res, uint32_t filter, Fn&& fn) {
    for (uint64_t k = res.range_offsets[filter]; k < res.range_offsets[filter + 1]; ++k) {
        const rgr_id_range r = res.ranges[k];
        const rgr_retain_val* v = res.vals[r.len >> 31] + r.begin;
        for (uint32_t i = 0, n = r.len & 0x7FFFFFFFu; i < n; ++i)
            if (!(v[i].flags & RGR_RETAIN_HIT_DEAD)) fn(v[i].topic_id);
    }
}
}  // namespace

GpuRetainStorage::GpuRetainStorage(int device, uint32_t retain_delta_max) {
    rgr_config cfg{};
    cfg.device = device;
    cfg.retain_delta_max = retain_delta_max;
    if (rgr_create(&cfg, &h_) != RGR_OK) h_ = nullptr;
}
GpuRetainStorage::~GpuRetainStorage() { if (h_) rgr_destroy(h_); }

// retain.rs:229-247: remove the old value; a non-empty payload stores the new one.
Result<bool> GpuRetainStorage::set(const TopicName& topic, const Retain& retain, int64_t expiry_ms, int64_t now_ms) {
    if (!h_) return Result<bool>::Err("no device");
    std::lock_guard<std::mutex> g(mu_);
    auto it = ids_.find(topic);
    const bool had = it != ids_.end();
    if (retain.payload.empty()) {
        if (had) {
            if (rgr_retain_topic_remove(h_, topic.data(), uint32_t(topic.size())) != RGR_OK) return Result<bool>::Err(rgr_last_error());
            slab_[it->second] = Entry{};
            free_.push_back(it->second);
            ids_.erase(it);
            retaineds_.dec();
            dirty_ = true;
        } else {
            // still has to be a valid topic name (Topic::from_str, retain.rs:235)
            uint32_t probe_id = 0xFFFFFFF0u;
            if (rgr_retain_topic_add(h_, topic.data(), uint32_t(topic.size()), probe_id) != RGR_OK) return Result<bool>::Err("invalid topic");
            rgr_retain_topic_remove(h_, topic.data(), uint32_t(topic.size()));
        }
        return Result<bool>::Ok(true);
    }
    uint32_t id;
    if (had) id = it->second;
    else if (!free_.empty()) { id = free_.back(); free_.pop_back(); }
    else { id = uint32_t(slab_.size()); slab_.emplace_back(); }
    if (rgr_retain_topic_add(h_, topic.data(), uint32_t(topic.size()), id) != RGR_OK) {
        if (!had) free_.push_back(id);
        return Result<bool>::Err(std::string("invalid topic: ") + rgr_last_error());
    }
    slab_[id] = Entry{topic, retain, expiry_ms ? now_ms + expiry_ms : 0, true};
    if (!had) { ids_.emplace(topic, id); retaineds_.inc(); }
    dirty_ = true;
    return Result<bool>::Ok(true);
}

// retain.rs:250-267
Result<std::vector<std::pair<TopicName, Retain>>> GpuRetainStorage::get(const TopicFilter& f, int64_t now_ms) {
    using R = Result<std::vector<std::pair<TopicName, Retain>>>;
    if (!h_) return R::Err("no device");
    std::lock_guard<std::mutex> g(mu_);
    if (dirty_) { if (rgr_retain_commit(h_) != RGR_OK) return R::Err(rgr_last_error()); dirty_ = false; }
    const uint64_t offs[2] = {0, f.size()};
    rgr_retain_ranges res{};
    if (rgr_retain_match_ranges(h_, reinterpret_cast<const uint8_t*>(f.data()), offs, 1, &res) != RGR_OK) return R::Err(rgr_last_error());
    std::vector<std::pair<TopicName, Retain>> out;
    const bool bad = res.status[0] != RGR_TOPIC_OK;
    if (!bad) for_each_hit(res, 0, [&](uint32_t id) {
        const Entry& e = slab_[id];
        if (!e.live || (e.expire_at && now_ms >= e.expire_at)) return;        // TimedValue::is_expired
        out.emplace_back(e.topic, e.retain);
    });
    rgr_retain_ranges_free(&res);
    if (bad) return R::Err("invalid topic filter `" + f + "`");
    return R::Ok(std::move(out));
}

// retain.rs:216-226 — RetainTree::retain(usize::MAX, |tv| !tv.is_expired())
size_t GpuRetainStorage::remove_expired_messages(int64_t now_ms) {
    if (!h_) return 0;
    std::lock_guard<std::mutex> g(mu_);
    size_t removed = 0;
    for (uint32_t id = 0; id < slab_.size(); ++id) {
        Entry& e = slab_[id];
        if (!e.live || !e.expire_at || now_ms < e.expire_at) continue;
        rgr_retain_topic_remove(h_, e.topic.data(), uint32_t(e.topic.size()));
        ids_.erase(e.topic);
        e = Entry{};
        free_.push_back(id);
        retaineds_.dec();
        ++removed;
        dirty_ = true;
    }
    return removed;
}

// ---- GpuMessageIndex (rmqtt-message-storage/src/ram.rs) ---------------------------------------
GpuMessageIndex::GpuMessageIndex(int device, uint32_t retain_delta_max) {
    rgr_config cfg{};
    cfg.device = device;
    cfg.retain_delta_max = retain_delta_max;
    if (rgr_create(&cfg, &h_) != RGR_OK) h_ = nullptr;
}
GpuMessageIndex::~GpuMessageIndex() { if (h_) rgr_destroy(h_); }

namespace {
// Topic::push(Level::Normal(msg_id.to_string())) in string form: one more '/'-separated level.
std::string with_msg_level(const TopicName& topic, MsgID id) { return topic + "/" + std::to_string(id); }
}  // namespace

// ram.rs:333-334 + :351/:361
Result<bool> GpuMessageIndex::set(const TopicName& topic, MsgID msg_id) {
    if (!h_) return Result<bool>::Err("no device");
    std::lock_guard<std::mutex> g(mu_);
    const std::string key = with_msg_level(topic, msg_id);
    auto it = ids_.find(msg_id);
    uint32_t id;
    const bool had = it != ids_.end();
    if (had) id = it->second;
    else if (!free_.empty()) { id = free_.back(); free_.pop_back(); }
    else { id = uint32_t(slab_.size()); slab_.push_back(0); }
    if (rgr_retain_topic_add(h_, key.data(), uint32_t(key.size()), id) != RGR_OK) {
        if (!had) free_.push_back(id);
        return Result<bool>::Err(std::string("invalid topic: ") + rgr_last_error());
    }
    slab_[id] = msg_id;
    if (!had) { ids_.emplace(msg_id, id); ++live_; }
    dirty_ = true;
    return Result<bool>::Ok(true);
}

// ram.rs:226-233
Result<bool> GpuMessageIndex::remove(const TopicName& topic, MsgID msg_id) {
    if (!h_) return Result<bool>::Err("no device");
    std::lock_guard<std::mutex> g(mu_);
    const std::string key = with_msg_level(topic, msg_id);
    const int32_t rc = rgr_retain_topic_remove(h_, key.data(), uint32_t(key.size()));
    if (rc == RGR_ENOENT) return Result<bool>::Ok(false);
    if (rc != RGR_OK) return Result<bool>::Err(rgr_last_error());
    auto it = ids_.find(msg_id);
    if (it != ids_.end()) { free_.push_back(it->second); ids_.erase(it); --live_; }
    dirty_ = true;
    return Result<bool>::Ok(true);
}

// ram.rs:381-394
Result<std::vector<MsgID>> GpuMessageIndex::get(const TopicFilter& f) {
    using R = Result<std::vector<MsgID>>;
    if (!h_) return R::Err("no device");
    std::lock_guard<std::mutex> g(mu_);
    if (dirty_) { if (rgr_retain_commit(h_) != RGR_OK) return R::Err(rgr_last_error()); dirty_ = false; }
    // last level `#`  <=>  the string is "#" or ends in "/#" (levels are the '/'-separated pieces)
    const bool multi = f == "#" || (f.size() >= 2 && f.compare(f.size() - 2, 2, "/#") == 0);
    const std::string q = multi ? f : f + "/+";
    const uint64_t offs[2] = {0, q.size()};
    rgr_retain_ranges res{};
    if (rgr_retain_match_ranges(h_, reinterpret_cast<const uint8_t*>(q.data()), offs, 1, &res) != RGR_OK) return R::Err(rgr_last_error());
    const bool bad = res.status[0] != RGR_TOPIC_OK;
    std::vector<MsgID> out;
    if (!bad) for_each_hit(res, 0, [&](uint32_t id) { out.push_back(slab_[id]); });
    rgr_retain_ranges_free(&res);
    if (bad) return R::Err("invalid topic filter `" + f + "`");
    std::sort(out.begin(), out.end());
    return R::Ok(std::move(out));
}

}  // namespace rmqtt
