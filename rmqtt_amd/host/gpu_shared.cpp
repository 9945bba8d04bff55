// DefaultShared / GpuShared — see gpu_shared.hpp.
#include "gpu_shared.hpp"

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <string_view>

namespace rmqtt {

// ---------------------------------------------------------------------------------------------- the reference's path (shared.rs:735-820)
Result<ForwardedCount> DefaultShared::forwards(const From& from, const Publish& publish, std::vector<Undelivered>* errs) {
    if (publish.target_clientid) {                                     // shared.rs:744-770
        SubscriptionOptions opts;
        opts.qos = publish.qos;
        SubRelations relations{SubRelation{*publish.topic, *publish.target_clientid, opts, std::nullopt, std::nullopt}};
        forwards_to(from, publish, relations, errs);
        return Result<ForwardedCount>::Ok(from.id.node_id == this_node_ ? 1 : 0);
    }
    auto m = router_.matches(from.id, *publish.topic);                  // shared.rs:772; an Err is logged and an empty map is used (:774-777)
    SubRelationsMap map = m.ok() ? std::move(*m.value) : SubRelationsMap{};
    ForwardedCount count = 0;
    auto it = map.find(this_node_);                                    // shared.rs:781
    if (it != map.end()) {
        count = forwards_to(from, publish, it->second, errs);
        map.erase(it);
    }
    for (auto& kv : map) remote_ += kv.second.size();                  // shared.rs:809-815: a single-node Shared only warns about them
    return Result<ForwardedCount>::Ok(count);
}

// shared.rs:876-963
ForwardedCount DefaultShared::forwards_to(const From& from, const Publish& publish, SubRelations& relations, std::vector<Undelivered>* errs) {
    ForwardedCount ok = 0;
    for (SubRelation& r : relations) {
        const bool retain = r.opts.v5 ? (r.opts.retain_as_published && publish.retain) : false;      // :886-897 (retain_as_published() is Some for v5 only)
        Publish p = publish;                                                                            // :899
        p.dup = false; p.retain = retain; p.qos = std::min(p.qos, r.opts.qos); p.packet_id.reset();     // :900-903
        if (r.sub_ids) p.subscription_ids = *r.sub_ids;                                                 // :904-908
        Tx* tx = this->tx(r.client_id);                                                                 // :910
        if (!tx) { if (errs) errs->push_back(Undelivered{r.client_id, "the client has disconnected"}); continue; }
        if (!tx->unbounded_send(from, std::move(p))) { if (errs) errs->push_back(Undelivered{r.client_id, "Connection Tx is closed"}); continue; }
        ok += r.group ? r.group->group_cids.size() : 1;                                                 // :945-956: one entry per member of a chosen group
    }
    relations.clear();                                                                                  // relations.drain(..)
    return ok;
}

// ---------------------------------------------------------------------------------------------- the device's path
GpuShared::GpuShared(GpuRouter& router, Shared& inner, size_t max_batch, std::chrono::microseconds max_delay, unsigned passes_in_flight, unsigned workers)
    : router_(router), inner_(inner), batcher_(router, max_batch, max_delay, passes_in_flight, workers) {}

// forwards_to over delivery words.  v3 hits are sent as they go by; a v5 client's FIRST hit carries the subscription identifiers of its later
// hits as well (types.rs:526-534), so v5 rows are held until the publish's hits have all been seen — v3 rows, then the v5 collector's, is the
// order the reference sends in too (types.rs:488-497).
GpuRouter::DeliverOutcome GpuShared::deliver(const GpuRouter::DeliverPass& pass, size_t index, const From& from, const Publish& publish, ForwardedCount& count,
                                             std::vector<Undelivered>* errs) {
    struct Row5 { const ClientId* client; uint8_t qos; bool retain; std::vector<uint32_t> ids; };
    std::vector<Row5> rows5;
    std::unordered_map<std::string_view, size_t> index5;               // built only when a later hit has an identifier to add
    bool indexed = false;
    uint64_t remote = 0, sent = 0;
    count = 0;
    const NodeId here = router_.this_node();
    auto send = [&](const ClientId& client, uint8_t qos, bool retain, std::vector<uint32_t>* ids) {
        Tx* tx = inner_.tx(client);                                                                     // shared.rs:910
        if (!tx) { if (errs) errs->push_back(Undelivered{client, "the client has disconnected"}); return; }
        Publish p = publish;                                                                            // shared.rs:899-908, with the device's qos' / retain'
        p.dup = false; p.retain = retain; p.qos = qos; p.packet_id.reset();
        if (ids && !ids->empty()) p.subscription_ids = std::move(*ids);
        if (!tx->unbounded_send(from, std::move(p))) { if (errs) errs->push_back(Undelivered{client, "Connection Tx is closed"}); return; }
        ++count; ++sent;
    };
    const auto outcome = router_.visit_deliveries(pass, index, [&](const GpuRouter::Delivery& d) {
        if (d.node_id != here) { ++remote; return; }                                                    // shared.rs:809-815
        if (d.v5_duplicate) {                                                                           // types.rs:526-534: only its identifier counts
            if (!d.subscription_identifier) return;
            if (!indexed) { for (size_t i = 0; i < rows5.size(); ++i) index5.emplace(std::string_view(*rows5[i].client), i); indexed = true; }
            auto it = index5.find(std::string_view(*d.client_id));
            if (it != index5.end()) rows5[it->second].ids.push_back(d.subscription_identifier);
            return;
        }
        // v3 rows (never duplicated, no identifiers) go out now; a v5 row may still collect identifiers of later duplicate hits: it waits
        if (d.is_v5) {
            if (indexed) index5.emplace(std::string_view(*d.client_id), rows5.size());
            rows5.push_back(Row5{d.client_id, d.qos, d.retain, d.subscription_identifier ? std::vector<uint32_t>{d.subscription_identifier} : std::vector<uint32_t>{}});
        } else send(*d.client_id, d.qos, d.retain, nullptr);
    });
    if (outcome != GpuRouter::DeliverOutcome::Done) { count = 0; return outcome; }      // (nothing was sent: NeedsHostPath is decided before the first hit is visited)
    for (Row5& r : rows5) send(*r.client, r.qos, r.retain, &r.ids);
    deliveries_ += sent; remote_ += remote;
    return outcome;
}

void GpuShared::on_pass(void* user, uint64_t tag, const std::shared_ptr<GpuRouter::DeliverPass>& pass, size_t index, const Id&, const std::string* err) {
    std::unique_ptr<Pending> p(static_cast<Pending*>(user));
    GpuShared* self = p->self;
    if (err) { p->done(p->user, tag, 0, err); return; }
    ForwardedCount count = 0;
    const auto outcome = self->deliver(*pass, index, *p->from, *p->publish, count, nullptr);
    if (outcome == GpuRouter::DeliverOutcome::Stale && p->tries < kMaxResubmits) {
        // a removal overtook the pass: the publish joins another batch (r7y: handing each of them to the host path — a device pass of ONE publish
        // each — took the throughput from 4.1 M to 1.5 M publishes/s under two removals a second, profiles/r07x_*)
        p->tries++;
        self->resubmitted_++;
        const From* from = p->from; const Publish* publish = p->publish;
        const GpuRouter::OwnerHint hint = self->owner_hint_of(*from);
        self->batcher_.submit_deliver(from->id, *publish->topic, uint8_t((publish->qos & 3u) | (publish->retain ? 4u : 0u)), &GpuShared::on_pass, p.release(), tag, &hint);
        return;
    }
    if (outcome == GpuRouter::DeliverOutcome::NeedsHostPath || outcome == GpuRouter::DeliverOutcome::Stale) {
        self->host_path_++;
        GpuRouter::SharedPause pause(self->router_);          // (the worker's run holds the table's shared lock; the host path takes it itself)
        auto r = self->inner_.forwards(*p->from, *p->publish, nullptr);
        if (r.ok()) p->done(p->user, tag, *r.value, nullptr); else p->done(p->user, tag, 0, &r.error);
        return;
    }
    self->device_path_++;
    p->done(p->user, tag, count, nullptr);              // InvalidTopic: nobody matched (the reference logs the Err and forwards to nobody)
}

// the publisher's cached owner id (From::owner_hint), refreshed when the owner index has changed since it was read
GpuRouter::OwnerHint GpuShared::owner_hint_of(const From& from) {
    uint32_t bucket = from.owner_bucket.load(std::memory_order_relaxed);
    if (bucket == 0xFFFFFFFFu) { bucket = GpuRouter::owner_bucket_of(from.id); from.owner_bucket.store(bucket, std::memory_order_relaxed); }
    const uint64_t oe = router_.owners_epoch(bucket);
    const uint64_t packed = from.owner_hint.load(std::memory_order_relaxed);
    if (packed != 0 && uint32_t(packed) == uint32_t(oe)) return GpuRouter::OwnerHint{uint32_t(packed >> 32), bucket, oe};
    const GpuRouter::OwnerHint hint = router_.owner_hint(from.id);
    // (0 means "never looked up": an epoch whose low word is 0 is simply not cached)
    if (uint32_t(hint.epoch) != 0) from.owner_hint.store(uint64_t(hint.owner) << 32 | uint32_t(hint.epoch), std::memory_order_relaxed);
    return hint;
}

void GpuShared::submit(const From* from, const Publish* publish, Done done, void* user, uint64_t tag) {
    if (publish->target_clientid) {                     // shared.rs:744: no matching involved
        host_path_++;
        auto r = inner_.forwards(*from, *publish, nullptr);
        if (r.ok()) done(user, tag, *r.value, nullptr); else done(user, tag, 0, &r.error);
        return;
    }
    auto* p = new Pending{this, from, publish, done, user};
    const GpuRouter::OwnerHint hint = owner_hint_of(*from);
    batcher_.submit_deliver(from->id, *publish->topic, uint8_t((publish->qos & 3u) | (publish->retain ? 4u : 0u)), &GpuShared::on_pass, p, tag, &hint);
}

Result<ForwardedCount> GpuShared::forwards(const From& from, const Publish& publish, std::vector<Undelivered>* errs) {
    if (publish.target_clientid) { host_path_++; return inner_.forwards(from, publish, errs); }
    // the blocking caller consumes its publish itself (so that `errs` can be filled): a pass of its own request through the batcher — again when a
    // removal overtook the pass, the host path after kMaxResubmits of those
    struct Wait { std::mutex m; std::condition_variable cv; bool done = false; std::shared_ptr<GpuRouter::DeliverPass> pass; size_t index = 0; std::string err; };
    ForwardedCount count = 0;
    for (unsigned tries = 0;; ++tries) {
        Wait w;
        const GpuRouter::OwnerHint hint = owner_hint_of(from);
        batcher_.submit_deliver(from.id, *publish.topic, uint8_t((publish.qos & 3u) | (publish.retain ? 4u : 0u)),
                                [](void* user, uint64_t, const std::shared_ptr<GpuRouter::DeliverPass>& pass, size_t index, const Id&, const std::string* err) {
                                    auto* w = static_cast<Wait*>(user);
                                    std::lock_guard<std::mutex> g(w->m);
                                    if (err) w->err = *err; else { w->pass = pass; w->index = index; }
                                    w->done = true;
                                    w->cv.notify_one();
                                }, &w, 0, &hint);
        { std::unique_lock<std::mutex> lk(w.m); w.cv.wait(lk, [&] { return w.done; }); }
        if (!w.err.empty()) return Result<ForwardedCount>::Err(w.err);
        const auto outcome = deliver(*w.pass, w.index, from, publish, count, errs);
        if (outcome == GpuRouter::DeliverOutcome::Stale && tries < kMaxResubmits) { resubmitted_++; continue; }
        if (outcome == GpuRouter::DeliverOutcome::NeedsHostPath || outcome == GpuRouter::DeliverOutcome::Stale) { host_path_++; return inner_.forwards(from, publish, errs); }
        break;
    }
    device_path_++;
    return Result<ForwardedCount>::Ok(count);
}

}  // namespace rmqtt
