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
// node (rgr_sub_add_ex), and `matches` only reads the RGR_HIT_* flags.  Shared-group choice
// (router.rs:236-255) needs live session state and is not modelled (flagged RGR_SUB_SHARED for
// the Rust glue).
#pragma once
#include <cstdint>
#include <map>
#include <mutex>
#include <optional>
#include <string>
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

// types.rs:607-827 (fields the matching path carries).
struct SubscriptionOptions {
    bool v5 = false;
    uint8_t qos = 0;
    bool no_local = false, retain_as_published = false;
    uint8_t retain_handling = 0;
    uint32_t subscription_identifier = 0;   // 0 = None
    bool is_v3() const { return !v5; }
    std::optional<bool> opt_no_local() const { return v5 ? std::optional<bool>(no_local) : std::nullopt; }
};

struct SubRelation {   // types.rs:478-484 (shared-group member omitted)
    TopicFilter topic_filter;
    ClientId client_id;
    SubscriptionOptions opts;
    std::optional<std::vector<uint32_t>> sub_ids;
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
    ~GpuRouter() override;
    GpuRouter(const GpuRouter&) = delete;
    GpuRouter& operator=(const GpuRouter&) = delete;
    bool usable() const { return h_ != nullptr; }
    const std::string& create_error() const { return create_error_; }

    Result<bool> add(const std::string& topic_filter, const Id& id, const SubscriptionOptions& opts) override;
    Result<bool> remove(const std::string& topic_filter, const Id& id) override;
    Result<SubRelationsMap> matches(const Id& id, const TopicName& topic) override;
    // Batched form for a micro-batcher in front of the trait: one device pass for many publishes.
    // out[i] is nullopt where the reference would return Err (invalid topic name).
    Result<bool> matches_batch(const std::vector<Id>& ids, const std::vector<TopicName>& topics,
                               std::vector<std::optional<SubRelationsMap>>& out);
    bool is_online(NodeId, const std::string&) override { return true; }   // session state lives in the broker
    std::vector<Route> gets(size_t limit) override;
    Result<std::vector<Route>> get(const std::string& topic) override;       // router.rs:157-170 (.unique())
    Result<bool> has_matches(const std::string& topic);                     // router.rs:151-154
    size_t topics_tree() override;
    Counter topics() override { return topics_count_; }
    Counter routes() override { return relations_count_; }
    std::vector<std::string> list_topics(size_t top) override;

   private:
    struct Rel { Id id; SubscriptionOptions opts; uint32_t sub_id; uint32_t owner_id; };
    struct Dense {                       // string key -> dense u32 id with reference counts
        std::unordered_map<std::string, std::pair<uint32_t, uint32_t>> ids;   // key -> (id, refs)
        std::vector<uint32_t> free;
        uint32_t next = 0;
        uint32_t acquire(const std::string& k);
        void release(const std::string& k);
        uint32_t find(const std::string& k) const;                             // RGR_ID_NONE if absent
    };
    struct FilterEntry { uint32_t filter_id; std::unordered_map<ClientId, Rel> rels; };
    struct Slot { const std::string* filter = nullptr; const Rel* rel = nullptr; };

    rgr_handle* h_ = nullptr;
    std::string create_error_;
    NodeId this_node_;
    std::mutex mu_;   // the reference uses DashMap + a trie RwLock; one mutex is enough for the mirror
    std::unordered_map<TopicFilter, FilterEntry> relations_;   // AllRelationsMap
    std::vector<Slot> slab_;           // sub_id -> relation
    std::vector<uint32_t> free_sub_ids_;
    std::unordered_map<uint32_t, const std::string*> filter_names_;   // filter_id -> filter string
    Dense owners_, clients_;             // Id -> owner_id, (node, ClientId) -> client_idx
    std::vector<NodeId> nodes_;          // node_idx -> NodeId
    std::unordered_map<NodeId, uint16_t> node_idx_;
    Counter topics_count_, relations_count_;
    bool dirty_ = false;

    int32_t commit_if_dirty();
};

}  // namespace rmqtt
