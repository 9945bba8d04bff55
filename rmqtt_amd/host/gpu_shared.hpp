// C++ host-side mirror of the part of rmqtt's `Shared` trait that sits right behind the matcher:
//
//   Shared::forwards      rmqtt/src/shared.rs:735-820   router.matches(from.id, &publish.topic) -> this node's relations -> forwards_to
//   Shared::forwards_to   rmqtt/src/shared.rs:876-963   per relation: retain' (Retain As Published), qos' = min, subscription ids, the
//                                                      session's tx (self.tx(&client_id), shared.rs:910), tx.unbounded_send(Message::Forward)
//
// `DefaultShared` restates the reference: it builds the SubRelationsMap through whatever `Router` it is given and walks this node's relations.
// `GpuShared` is what the plugin installs through `extends.shared_mut()` (rmqtt/src/extend.rs:123; rmqtt-cluster-raft/src/lib.rs and
// rmqtt-cluster-broadcast/src/lib.rs replace the same slot): publishes are micro-batched (Batcher, delivery kind), ONE device pass per batch runs
// the delivery stage (rgr_group_match_batch_deliver with the publishes' real qos / retain), and every publish's hits go from delivery words
// straight into the sessions' channels — no SubRelationsMap, no per-hit clones of filter strings and options.  What stays on the host per
// recipient is what cannot be anywhere else: the session lookup and the channel send.  Publishes the device cannot finish ($share members among
// the hits — SharedSubscription::choice is the broker's; `target_clientid`; a pass older than the last removal) take `inner`'s path unchanged.
// Rust twin: rust/rmqtt-gpu-router/src/shared.rs.
#pragma once
#include <atomic>
#include <memory>
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "gpu_router.hpp"

namespace rmqtt {

struct From {                                               // types.rs From: the publisher's Id (+ kind)
    Id id;
    // the publisher's owner id in the device table, kept by whoever keeps the From (a session keeps its own): owner << 32 | low 32 bits of the epoch of
    // the Id's bucket of the owner index it was read at (0 = never looked up: bucket epochs start at 1), and the bucket itself (the hash of an Id never
    // changes).  GpuShared refreshes the hint when the bucket's epoch has moved.
    mutable std::atomic<uint64_t> owner_hint{0};
    mutable std::atomic<uint32_t> owner_bucket{0xFFFFFFFFu};
    From() = default;
    explicit From(Id i) : id(std::move(i)) {}
    From(const From& o) : id(o.id), owner_hint(o.owner_hint.load(std::memory_order_relaxed)), owner_bucket(o.owner_bucket.load(std::memory_order_relaxed)) {}
    From& operator=(const From& o) {
        id = o.id;
        owner_hint.store(o.owner_hint.load(std::memory_order_relaxed), std::memory_order_relaxed);
        owner_bucket.store(o.owner_bucket.load(std::memory_order_relaxed), std::memory_order_relaxed);
        return *this;
    }
};
struct Publish {                                            // types.rs Publish: the fields forwards / forwards_to read or rewrite
    std::shared_ptr<const TopicName> topic;                 // ByteString: shared, a clone per recipient (shared.rs:899) bumps a count, copies nothing
    uint8_t qos = 0;
    bool retain = false, dup = false;
    std::optional<uint16_t> packet_id;
    std::vector<uint32_t> subscription_ids;                 // properties.subscription_ids (v5)
    std::shared_ptr<const std::string> payload;             // Bytes: shared, never copied
    std::optional<ClientId> target_clientid;
};
using ForwardedCount = size_t;

// A session's sending half (`Tx`, shared.rs:910-930).  unbounded_send(Message::Forward(from, p)): false = "Connection Tx is closed".
struct Tx {
    virtual ~Tx() = default;
    virtual bool unbounded_send(const From& from, Publish&& p) = 0;
};
struct Undelivered { ClientId to; std::string reason; };    // (To, From, Publish, Reason) of shared.rs:918-943, the part tests look at

class Shared {                                              // the slice of rmqtt/src/shared.rs `trait Shared` this path touches
   public:
    virtual ~Shared() = default;
    virtual Tx* tx(const ClientId& client_id) = 0;          // shared.rs: self.tx(&client_id) — peers map lookup
    // Ok: recipients reached; errs (optional out): the relations that could not be sent to
    virtual Result<ForwardedCount> forwards(const From& from, const Publish& publish, std::vector<Undelivered>* errs = nullptr) = 0;
};

// Registry of connected sessions (DefaultShared::peers, shared.rs:672): client id -> tx.
class Sessions {
   public:
    void connect(const ClientId& c, std::shared_ptr<Tx> tx) { std::unique_lock<std::shared_mutex> g(mu_); peers_[c] = std::move(tx); }
    void disconnect(const ClientId& c) { std::unique_lock<std::shared_mutex> g(mu_); peers_.erase(c); }
    Tx* find(const ClientId& c) const { std::shared_lock<std::shared_mutex> g(mu_); auto it = peers_.find(c); return it == peers_.end() ? nullptr : it->second.get(); }

   private:
    mutable std::shared_mutex mu_;
    std::unordered_map<ClientId, std::shared_ptr<Tx>> peers_;
};

// The reference's own path over any Router (shared.rs:735-820, 876-963).
class DefaultShared : public Shared {
   public:
    DefaultShared(Router& router, Sessions& sessions, NodeId this_node) : router_(router), sessions_(sessions), this_node_(this_node) {}
    Tx* tx(const ClientId& c) override { return sessions_.find(c); }
    Result<ForwardedCount> forwards(const From& from, const Publish& publish, std::vector<Undelivered>* errs = nullptr) override;
    ForwardedCount forwards_to(const From& from, const Publish& publish, SubRelations& relations, std::vector<Undelivered>* errs);
    uint64_t remote_relations() const { return remote_; }   // relations of other nodes met ("Received message from remote node", shared.rs:809-815)

   private:
    Router& router_;
    Sessions& sessions_;
    NodeId this_node_;
    std::atomic<uint64_t> remote_{0};
};

class GpuShared : public Shared {
   public:
    // workers = 0: completions run on the batcher's driver threads
    GpuShared(GpuRouter& router, Shared& inner, size_t max_batch = 4096, std::chrono::microseconds max_delay = std::chrono::microseconds(200),
              unsigned passes_in_flight = 3, unsigned workers = 0);
    Tx* tx(const ClientId& c) override { return inner_.tx(c); }
    // blocking form (a tokio task awaiting `forwards`)
    Result<ForwardedCount> forwards(const From& from, const Publish& publish, std::vector<Undelivered>* errs = nullptr) override;
    // asynchronous form: `done(user, tag, count, error-or-null)` runs on a batcher thread; from / publish must stay alive until then
    using Done = void (*)(void* user, uint64_t tag, ForwardedCount count, const std::string* err);
    void submit(const From* from, const Publish* publish, Done done, void* user, uint64_t tag);
    // One publish of a finished pass: delivery words -> sessions (forwards_to without the map).  count: recipients reached.
    GpuRouter::DeliverOutcome deliver(const GpuRouter::DeliverPass& pass, size_t index, const From& from, const Publish& publish, ForwardedCount& count,
                                      std::vector<Undelivered>* errs);
    GpuRouter::OwnerHint owner_hint_of(const From& from);       // From::owner_hint, refreshed against the owner index's epoch
    struct Counters { uint64_t device_path, host_path, deliveries, remote, passes, resubmitted; };
    Counters counters() const { return Counters{device_path_, host_path_, deliveries_, remote_, batcher_.passes(), resubmitted_}; }
    Batcher::Timing timing() const { return batcher_.timing(); }        // where the batcher's threads spent their time (ns summed over threads)

   private:
    struct Pending { GpuShared* self; const From* from; const Publish* publish; Done done; void* user; unsigned tries = 0; };
    static constexpr unsigned kMaxResubmits = 3;        // a publish whose pass a removal overtook joins another batch; after that many, the host path
    std::atomic<uint64_t> resubmitted_{0};
    static void on_pass(void* user, uint64_t tag, const std::shared_ptr<GpuRouter::DeliverPass>& pass, size_t index, const Id& from, const std::string* err);
    GpuRouter& router_;
    Shared& inner_;
    Batcher batcher_;
    std::atomic<uint64_t> device_path_{0}, host_path_{0}, deliveries_{0}, remote_{0};
};

}  // namespace rmqtt
