// C++ host-side mirror of rmqtt's `RetainStorage` surface over the C ABI (retain twin).
//
// Mirrors DefaultRetainStorage (rmqtt/src/retain.rs:198-301) / the retainer plugin's storage
// (rmqtt-plugins/rmqtt-retainer/src/storage.rs:576-641): the messages stay on the host keyed by a
// dense topic_id; only the in-memory topic index (`RetainTree<TimedValue<..>>`) is replaced by
// the device table.  Same method names / semantics as the trait (retain.rs:100-186):
//   set(topic, retain, expiry)   empty payload deletes (retain.rs:229-247)
//   get(topic_filter)            wildcard match + drop expired (retain.rs:250-267)
//   count() / max()
//   remove_expired_messages()    RetainTree::retain(|tv| !tv.is_expired()) (retain.rs:216-226)
#pragma once
#include <cstdint>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "gpu_router.hpp"

namespace rmqtt {

struct Retain {                // types.rs `Retain`: the fields the index path carries
    std::string payload;
    uint8_t qos = 0;
};

class GpuRetainStorage {
   public:
    // retain_delta_max > 0: two-tier retained set (rgr_config.retain_delta_max)
    explicit GpuRetainStorage(int device = 0, uint32_t retain_delta_max = 0);
    ~GpuRetainStorage();
    bool usable() const { return h_ != nullptr; }
    // `now_ms` / `expiry_ms` make TimedValue::is_expired (types.rs:2308-2338) testable: 0 = never expires.
    Result<bool> set(const TopicName& topic, const Retain& retain, int64_t expiry_ms = 0, int64_t now_ms = 0);
    Result<std::vector<std::pair<TopicName, Retain>>> get(const TopicFilter& topic_filter, int64_t now_ms = 0);
    size_t remove_expired_messages(int64_t now_ms);
    Counter count_max() const { return retaineds_; }

   private:
    struct Entry { TopicName topic; Retain retain; int64_t expire_at = 0; bool live = false; };
    rgr_handle* h_ = nullptr;
    std::mutex mu_;
    std::unordered_map<TopicName, uint32_t> ids_;
    std::vector<Entry> slab_;
    std::vector<uint32_t> free_;
    Counter retaineds_;
    bool dirty_ = false;
};

// the message-storage topic index.  rmqtt-message-storage keeps
// `topic_tree: RwLock<RetainTree<MsgID>>` (rmqtt-plugins/rmqtt-message-storage/src/ram.rs:158):
//   _set  (ram.rs:333-334,351,361): Topic::from_str(publish.topic) + push(Level::Normal(msg_id.to_string()))
//                                   -> topic_tree.insert(&topic, msg_id)
//   _get  (ram.rs:381-394): Topic::from_str(filter); push(Level::SingleWildcard) unless the last
//                           level is `#`; topic_tree.matches(&topic) -> msg ids
//   remove (ram.rs:232): topic_tree.remove(&topic) with the same msg-id level appended
// (One divergence, unreachable from the broker: a PUBLISH topic ending in `#` would be indexed by
// the reference because it validates before pushing the id level; here it is an Err.  PUBLISH
// topic names with wildcards are rejected before they get this far.)
// Same structure and query shape as the retained path, so it binds the same rgr_retain_* calls;
// the stored messages, expiries heap and forwardeds map stay host-side as in the reference.
using MsgID = uint64_t;

// One new leaf per stored message (ram.rs:333-394): the index runs in two-tier mode by default — additions recompile
// a small delta table, removals set dead bits, the tiers merge when the delta passes this many topics.
constexpr uint32_t kDefaultDeltaMax = 65536;
class GpuMessageIndex {
   public:
    // the index gains one leaf per stored message: this is the table the two-tier mode is for
    explicit GpuMessageIndex(int device = 0, uint32_t retain_delta_max = kDefaultDeltaMax);
    ~GpuMessageIndex();
    bool usable() const { return h_ != nullptr; }
    Result<bool> set(const TopicName& topic, MsgID msg_id);
    Result<bool> remove(const TopicName& topic, MsgID msg_id);        // Ok(false): nothing stored there
    Result<std::vector<MsgID>> get(const TopicFilter& topic_filter);  // ids in ascending order
    size_t values_size() const { return live_; }

   private:
    rgr_handle* h_ = nullptr;
    std::mutex mu_;
    std::vector<MsgID> slab_;            // dense topic_id -> MsgID
    std::vector<uint32_t> free_;
    std::unordered_map<MsgID, uint32_t> ids_;
    size_t live_ = 0;
    bool dirty_ = false;
};

}  // namespace rmqtt
