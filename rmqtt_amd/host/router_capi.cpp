// ctypes-facing shim over rmqtt::GpuRouter so that the Python parity tests can drive the C++
// Router mirror and compare its SubRelationsMap with the oracle's DefaultRouter, in the
// canonical text form of SURVEY.md App. A.5 (see oracle.cpp: orc_router_matches).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <memory>

#include "gpu_retain.hpp"
#include "gpu_router.hpp"
#include "gpu_shared.hpp"
#include "raft_snapshot.hpp"

using namespace rmqtt;

namespace {
char* dup_str(const std::string& s) {
    char* p = static_cast<char*>(std::malloc(s.size() + 1));
    std::memcpy(p, s.data(), s.size());
    p[s.size()] = 0;
    return p;
}
struct hr_id { uint64_t node_id; const char* client_id; uint32_t client_len; int64_t create_time; uint16_t lid; };
struct hr_opts { uint8_t v5, qos, no_local, retain_as_published, retain_handling; uint32_t sub_ident; const char* shared_group; uint32_t shared_group_len; };
Id mk_id(const hr_id* i) { Id id; id.node_id = i->node_id; id.lid = i->lid; id.create_time = i->create_time; id.client_id.assign(i->client_id, i->client_len); return id; }
SubscriptionOptions mk_opts(const hr_opts* o) {
    SubscriptionOptions s; s.v5 = o->v5; s.qos = o->qos; s.no_local = o->no_local; s.retain_as_published = o->retain_as_published;
    s.retain_handling = o->retain_handling; s.subscription_identifier = o->sub_ident;
    if (o->shared_group && o->shared_group_len) s.shared_group = std::string(o->shared_group, o->shared_group_len);
    return s;
}
// "\t$<group>:<online>:<sorted member client ids,>" — same text as the oracle's dump (oracle.cpp group_text)
std::string group_text(const SubRelation& s) {
    if (!s.group) return "";
    auto c = s.group->group_cids;
    std::sort(c.begin(), c.end());
    std::string r = "\t$" + s.group->group + ":" + std::to_string(int(s.group->is_online)) + ":";
    for (size_t i = 0; i < c.size(); ++i) { if (i) r.push_back(','); r += c[i]; }
    return r;
}
// order-independent test policy shared with the oracle (orc_router_set_shared_policy 1): smallest client id
struct SmallestClient final : SharedSubscription {
    bool is_supported() const override { return true; }
    std::optional<std::pair<size_t, bool>> choice(const std::string&, const Id&, const TopicName&, const std::vector<SharedCandidate>& ncs) override {
        if (ncs.empty()) return std::nullopt;
        size_t best = 0;
        for (size_t i = 1; i < ncs.size(); ++i) if (ncs[i].client_id < ncs[best].client_id) best = i;
        return std::make_pair(best, ncs[best].is_online);
    }
};
std::string dump(const SubRelationsMap& m) {
    std::string out;
    for (auto& kv : m) {
        out += "N " + std::to_string(kv.first) + "\n";
        std::vector<std::string> v3, v5;
        for (auto& s : kv.second) {
            if (s.opts.is_v3()) v3.push_back("3 " + s.topic_filter + "\t" + s.client_id + "\t" + std::to_string(s.opts.qos) + group_text(s) + "\n");
            else {
                std::string ids = "-";
                if (s.sub_ids) {
                    auto v = *s.sub_ids; std::sort(v.begin(), v.end()); ids.clear();
                    for (size_t i = 0; i < v.size(); ++i) { if (i) ids.push_back(','); ids += std::to_string(v[i]); }
                }
                v5.push_back("5 " + s.client_id + "\t" + s.topic_filter + "\t" + std::to_string(s.opts.qos) + "\t" + std::to_string(int(s.opts.no_local)) + "\t" + ids + group_text(s) + "\n");
            }
        }
        std::sort(v3.begin(), v3.end()); std::sort(v5.begin(), v5.end());
        for (auto& s : v3) out += s;
        for (auto& s : v5) out += s;
    }
    return out;
}
}  // namespace

extern "C" {
void* hr_new(uint64_t node_id, int device) {
    auto* r = new GpuRouter(node_id, device);
    if (!r->usable()) { delete r; return nullptr; }
    return r;
}
// one shard per entry of devs (repeated ordinals: several shards on one GPU)
void* hr_new_sharded(uint64_t node_id, const int* devs, uint32_t n) {
    auto* r = new GpuRouter(node_id, std::vector<int>(devs, devs + n));
    if (!r->usable()) { delete r; return nullptr; }
    return r;
}
uint32_t hr_shards(void* r) { return static_cast<GpuRouter*>(r)->shards(); }
// 0 = the reference's default SharedSubscription (selects nobody), 1 = smallest client id of the group
void hr_set_shared_policy(void* r, int policy) {
    static_cast<GpuRouter*>(r)->set_shared_subscription(policy == 1 ? std::make_shared<SmallestClient>() : nullptr);
}
uint64_t hr_flag_mismatches(void* r) { return static_cast<GpuRouter*>(r)->flag_mismatches(); }
// 0 auto, 1 filters (rgr_group_match_filter_subs + host expansion), 2 deliver (tuples with delivery words)
void hr_set_match_mode(void* r, int mode) {
    static_cast<GpuRouter*>(r)->set_match_mode(mode == 1 ? GpuRouter::MatchMode::Filters : mode == 2 ? GpuRouter::MatchMode::Deliver : GpuRouter::MatchMode::Auto);
}
uint64_t hr_stale_expansions(void* r) { return static_cast<GpuRouter*>(r)->stale_expansions(); }
// Bulk restore straight from (filter, client) arrays — the bench's way to load 10 M relations without 10 M trait calls
// (same path as ClusterRouter::restore: GpuRouter::restore over a decoded snapshot).
int hr_restore_bulk(void* r, const uint8_t* blob, const uint64_t* offs, const uint32_t* client, const uint8_t* qos, uint64_t n) {
    raft::Snapshot snap;
    snap.relations.reserve(n);
    for (uint64_t i = 0; i < n; ++i) {
        raft::Relation rel;
        rel.topic_filter.assign(reinterpret_cast<const char*>(blob) + offs[i], offs[i + 1] - offs[i]);
        rel.client_id = "c" + std::to_string(client[i]);
        rel.id.node_id = 1; rel.id.client_id = rel.client_id;
        rel.opts.qos = qos ? qos[i] : 0;
        snap.relations.push_back(std::move(rel));
    }
    auto res = static_cast<GpuRouter*>(r)->restore(snap);
    return res.ok() ? 0 : -1;
}
// What a broker sees: n_threads host threads call Router::matches through the Batcher, one publish per call, for
// `seconds` (or until every topic was published `rounds` times).  out[0] = publishes, out[1] = hits (rows of the returned
// maps), out[2] = device passes; lat_us (optional, [n_lat]) receives per-call latencies of thread 0 in microseconds.
int hr_e2e_run(void* r, const uint8_t* blob, const uint64_t* offs, uint32_t n, uint32_t n_threads, uint32_t max_batch, uint32_t max_delay_us,
               double seconds, uint64_t* out, double* wall_s, float* lat_us, uint32_t n_lat, uint32_t* n_lat_out) {
    auto* router = static_cast<GpuRouter*>(r);
    std::atomic<uint64_t> pubs{0}, hits{0};
    std::atomic<bool> stop{false};
    uint32_t lat_n = 0;
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t passes = 0;
    {
        Batcher b(*router, max_batch, std::chrono::microseconds(max_delay_us));
        std::vector<std::thread> th;
        for (uint32_t k = 0; k < n_threads; ++k)
            th.emplace_back([&, k] {
                Id id; id.node_id = 1; id.client_id = "publisher" + std::to_string(k);
                uint64_t my_p = 0, my_h = 0;
                for (uint64_t i = k; !stop.load(std::memory_order_relaxed); i += n_threads) {
                    const uint32_t t = uint32_t(i % n);
                    const auto a = std::chrono::steady_clock::now();
                    auto res = b.matches(id, std::string(reinterpret_cast<const char*>(blob) + offs[t], offs[t + 1] - offs[t]));
                    if (k == 0 && lat_us && lat_n < n_lat) lat_us[lat_n++] = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - a).count();
                    ++my_p;
                    if (res.ok()) for (auto& kv : *res.value) my_h += kv.second.size();
                }
                pubs += my_p; hits += my_h;
            });
        std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
        stop = true;
        for (auto& t : th) t.join();
        passes = b.passes();
    }
    if (wall_s) *wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    out[0] = pubs; out[1] = hits; out[2] = passes;
    if (n_lat_out) *n_lat_out = lat_n;
    return 0;
}
// The boundary at its design point: what tokio's task-level concurrency gives the reference's callers (shared.rs:772 is awaited by
// one task per PUBLISH, tens of thousands of them in flight on a few worker threads).  `n_submitters` threads keep `outstanding`
// publishes in flight in total through Batcher::submit; the completions (each publish's SubRelationsMap) run on `workers` pool
// threads.  out[0] = publishes completed, out[1] = rows of the returned maps, out[2] = device passes, out[3] = errors; lat_us:
// submit -> completion latencies of submitter 0's publishes, in microseconds.
int hr_e2e_run_async(void* r, const uint8_t* blob, const uint64_t* offs, uint32_t n, uint32_t n_submitters, uint32_t outstanding, uint32_t workers,
                     uint32_t passes_in_flight, uint32_t max_batch, uint32_t max_delay_us, double seconds, uint64_t* out, double* wall_s,
                     float* lat_us, uint32_t n_lat, uint32_t* n_lat_out) {
    auto* router = static_cast<GpuRouter*>(r);
    // Per submitter: what its completions count, spread over per-worker cache lines.  (r4: `pubs`, `rows` and `inflight` used to be one
    // atomic each per submitter — every completion of every worker did three read-modify-writes on lines all 64 workers and the
    // submitter shared; the harness's own bookkeeping, not the batcher, was what a publish cost.  Now a completion adds to the line of
    // ITS worker thread; the submitter sums the lines only when its own count of submissions says it may be at its cap.)
    constexpr uint32_t kLanes = 128;                                  // worker threads hash onto lanes (two workers on one lane: still correct, atomics)
    struct alignas(64) Lane { std::atomic<uint64_t> pubs{0}, rows{0}, errs{0}; char pad[40]; };
    struct alignas(64) Ctx { Lane lane[kLanes]; uint64_t submitted = 0, seen_done = 0; float* lat = nullptr; uint32_t n_lat = 0; std::atomic<uint32_t> lat_n{0};
                             std::chrono::steady_clock::time_point t0;
                             uint64_t done() const { uint64_t d = 0; for (const Lane& l : lane) d += l.pubs.load(std::memory_order_acquire); return d; } };
    std::vector<std::unique_ptr<Ctx>> ctx;
    for (uint32_t k = 0; k < n_submitters; ++k) ctx.push_back(std::make_unique<Ctx>());
    const uint64_t cap = std::max<uint64_t>(1, outstanding / std::max(1u, n_submitters));
    std::atomic<bool> stop{false};
    uint64_t passes = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (auto& c : ctx) c->t0 = t0;
    ctx[0]->lat = lat_us; ctx[0]->n_lat = lat_us ? n_lat : 0;
    double wall = 0;
    // completion: tag = submit time in ns since t0
    const Batcher::Callback done = [](void* user, uint64_t tag, Result<SubRelationsMap>&& res) {
        static std::atomic<uint32_t> next_lane{0};
        thread_local const uint32_t my_lane = next_lane.fetch_add(1, std::memory_order_relaxed) % kLanes;
        Ctx& c = *static_cast<Ctx*>(user);
        Lane& l = c.lane[my_lane];
        uint64_t rows = 0;
        if (res.ok()) for (auto& kv : *res.value) rows += kv.second.size();
        else if (res.error.rfind("invalid topic", 0) != 0) l.errs.fetch_add(1, std::memory_order_relaxed);
        if (rows) l.rows.fetch_add(rows, std::memory_order_relaxed);
        if (c.lat) {
            const uint32_t j = c.lat_n.fetch_add(1, std::memory_order_relaxed);
            if (j < c.n_lat) c.lat[j] = float(double(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - c.t0).count() - int64_t(tag)) / 1e3);
        }
        l.pubs.fetch_add(1, std::memory_order_release);               // last: the publish is complete
    };
    {
        Batcher b(*router, max_batch, std::chrono::microseconds(max_delay_us), passes_in_flight, workers);
        std::vector<std::thread> th;
        for (uint32_t k = 0; k < n_submitters; ++k)
            th.emplace_back([&, k] {
                Ctx& c = *ctx[k];
                Id id; id.node_id = 1; id.client_id = "publisher" + std::to_string(k);
                for (uint64_t i = k; !stop.load(std::memory_order_relaxed); i += n_submitters) {
                    if (c.submitted - c.seen_done >= cap) {                      // maybe at the cap: look at what has completed since
                        c.seen_done = c.done();
                        if (c.submitted - c.seen_done >= cap) { std::this_thread::sleep_for(std::chrono::microseconds(50)); i -= n_submitters; continue; }
                    }
                    const uint32_t t = uint32_t(i % n);
                    ++c.submitted;
                    const uint64_t now_ns = uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
                    b.submit(id, std::string_view(reinterpret_cast<const char*>(blob) + offs[t], offs[t + 1] - offs[t]), done, &c, now_ns);
                }
            });
        std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
        stop = true;
        for (auto& t : th) t.join();
        for (auto& c : ctx) while (c->done() < c->submitted) std::this_thread::sleep_for(std::chrono::microseconds(100));
        wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        passes = b.passes();
        const Batcher::Timing tm = b.timing();      // out[4..10]: where the batcher's threads spent their time (ns summed over threads); publishes that joined another batch (stale filter pass)
        out[4] = tm.collect_ns; out[5] = tm.pass_ns; out[6] = tm.dispatch_ns; out[7] = tm.task_ns; out[8] = tm.tasks; out[9] = tm.max_task_queue; out[10] = tm.requeued;
    }
    if (wall_s) *wall_s = wall;
    out[0] = out[1] = out[3] = 0;
    for (auto& c : ctx) for (const auto& l : c->lane) { out[0] += l.pubs.load(); out[1] += l.rows.load(); out[3] += l.errs.load(); }
    out[2] = passes;
    if (n_lat_out) *n_lat_out = std::min(ctx[0]->lat_n.load(), ctx[0]->n_lat);
    return 0;
}
// n publishes submitted asynchronously (Batcher::submit) from n_threads threads, completions on `workers` pool threads: dumps joined by
// '\x1e' as in hr_batcher_run.  The parity form of the asynchronous path.
char* hr_batcher_run_async(void* r, const hr_id* ids, const char* const* topics, const uint32_t* lens, uint32_t n, uint32_t n_threads,
                           uint32_t max_batch, uint32_t max_delay_us, uint32_t passes_in_flight, uint32_t workers, uint64_t* passes) {
    auto* router = static_cast<GpuRouter*>(r);
    struct Sink { std::vector<std::string> outs; std::atomic<uint32_t> done{0}; } sink;
    sink.outs.resize(n);
    std::vector<std::string>& outs = sink.outs;
    std::atomic<uint32_t>& done = sink.done;
    const Batcher::Callback cb = [](void* user, uint64_t tag, Result<SubRelationsMap>&& res) {
        Sink& s = *static_cast<Sink*>(user);
        s.outs[tag] = res.ok() ? dump(*res.value) : std::string("!ERR");
        s.done.fetch_add(1, std::memory_order_release);
    };
    {
        Batcher b(*router, max_batch, std::chrono::microseconds(max_delay_us), passes_in_flight, workers);
        std::vector<std::thread> th;
        for (uint32_t k = 0; k < n_threads; ++k)
            th.emplace_back([&, k] {
                for (uint32_t i = k; i < n; i += n_threads) b.submit(mk_id(&ids[i]), std::string_view(topics[i], lens[i]), cb, &sink, i);
            });
        for (auto& t : th) t.join();
        while (done.load(std::memory_order_acquire) < n) std::this_thread::sleep_for(std::chrono::microseconds(100));
        if (passes) *passes = b.passes();
    }
    std::string all;
    for (uint32_t i = 0; i < n; ++i) { if (i) all.push_back('\x1e'); all += outs[i]; }
    return dup_str(all);
}
// n publishes issued from n_threads threads through a Batcher (max_batch / max_delay_us): dumps joined by '\x1e'
// ("!ERR" where matches returned Err); *passes = device passes the batcher needed.
char* hr_batcher_run(void* r, const hr_id* ids, const char* const* topics, const uint32_t* lens, uint32_t n, uint32_t n_threads,
                     uint32_t max_batch, uint32_t max_delay_us, uint64_t* passes) {
    auto* router = static_cast<GpuRouter*>(r);
    std::vector<std::string> outs(n);
    {
        Batcher b(*router, max_batch, std::chrono::microseconds(max_delay_us));
        std::vector<std::thread> th;
        for (uint32_t k = 0; k < n_threads; ++k)
            th.emplace_back([&, k] {
                for (uint32_t i = k; i < n; i += n_threads) {
                    auto res = b.matches(mk_id(&ids[i]), std::string(topics[i], lens[i]));
                    outs[i] = res.ok() ? dump(*res.value) : std::string("!ERR");
                }
            });
        for (auto& t : th) t.join();
        if (passes) *passes = b.passes();
    }
    std::string all;
    for (uint32_t i = 0; i < n; ++i) { if (i) all.push_back('\x1e'); all += outs[i]; }
    return dup_str(all);
}
void hr_free(void* r) { delete static_cast<GpuRouter*>(r); }
void hr_free_str(char* p) { std::free(p); }
int hr_add(void* r, const char* f, uint32_t len, const hr_id* id, const hr_opts* o) {
    return static_cast<GpuRouter*>(r)->add(std::string(f, len), mk_id(id), mk_opts(o)).ok() ? 0 : -1;
}
// 0 removed, 1 not removed, -1 error
int hr_remove(void* r, const char* f, uint32_t len, const hr_id* id) {
    auto res = static_cast<GpuRouter*>(r)->remove(std::string(f, len), mk_id(id));
    return !res.ok() ? -1 : (*res.value ? 0 : 1);
}
char* hr_matches(void* r, const hr_id* id, const char* topic, uint32_t len) {
    auto res = static_cast<GpuRouter*>(r)->matches(mk_id(id), std::string(topic, len));
    return res.ok() ? dup_str(dump(*res.value)) : nullptr;
}
// routes joined by '\n'; NULL on Err
char* hr_get(void* r, const char* topic, uint32_t len) {
    auto res = static_cast<GpuRouter*>(r)->get(std::string(topic, len));
    if (!res.ok()) return nullptr;
    std::string s;
    for (auto& rt : *res.value) { s += rt.topic; s.push_back('\n'); }
    return dup_str(s);
}
// ---- GpuRetainStorage shim -------------------------------------------------------------------
void* hs_new(int device) {
    auto* s = new GpuRetainStorage(device);
    if (!s->usable()) { delete s; return nullptr; }
    return s;
}
void hs_free(void* s) { delete static_cast<GpuRetainStorage*>(s); }
int hs_set(void* s, const char* t, uint32_t tl, const char* payload, uint32_t pl, int64_t expiry_ms, int64_t now_ms) {
    return static_cast<GpuRetainStorage*>(s)->set(std::string(t, tl), Retain{std::string(payload, pl), 0}, expiry_ms, now_ms).ok() ? 0 : -1;
}
// "topic\tpayload\n" rows sorted by topic; NULL on Err
char* hs_get(void* s, const char* f, uint32_t fl, int64_t now_ms) {
    auto r = static_cast<GpuRetainStorage*>(s)->get(std::string(f, fl), now_ms);
    if (!r.ok()) return nullptr;
    std::vector<std::string> rows;
    for (auto& kv : *r.value) rows.push_back(kv.first + "\t" + kv.second.payload + "\n");
    std::sort(rows.begin(), rows.end());
    std::string out;
    for (auto& x : rows) out += x;
    return dup_str(out);
}
uint64_t hs_remove_expired(void* s, int64_t now_ms) { return static_cast<GpuRetainStorage*>(s)->remove_expired_messages(now_ms); }
int64_t hs_count(void* s) { return static_cast<GpuRetainStorage*>(s)->count_max().count; }
int64_t hs_max(void* s) { return static_cast<GpuRetainStorage*>(s)->count_max().max; }

// ---- GpuMessageIndex shim ---------------------------------------------------------------------
void* hm_new(int device) {
    auto* s = new GpuMessageIndex(device);
    if (!s->usable()) { delete s; return nullptr; }
    return s;
}
void hm_free(void* s) { delete static_cast<GpuMessageIndex*>(s); }
int hm_set(void* s, const char* t, uint32_t tl, uint64_t msg_id) {
    return static_cast<GpuMessageIndex*>(s)->set(std::string(t, tl), msg_id).ok() ? 0 : -1;
}
// 0 removed, 1 nothing there, -1 error
int hm_remove(void* s, const char* t, uint32_t tl, uint64_t msg_id) {
    auto r = static_cast<GpuMessageIndex*>(s)->remove(std::string(t, tl), msg_id);
    return !r.ok() ? -1 : (*r.value ? 0 : 1);
}
// decimal ids joined by ','; NULL on Err
char* hm_get(void* s, const char* f, uint32_t fl) {
    auto r = static_cast<GpuMessageIndex*>(s)->get(std::string(f, fl));
    if (!r.ok()) return nullptr;
    std::string out;
    for (auto id : *r.value) { out += std::to_string(id); out.push_back(','); }
    return dup_str(out);
}
uint64_t hm_values_size(void* s) { return static_cast<GpuMessageIndex*>(s)->values_size(); }

// ---- Raft snapshot reader shim (raft_snapshot.hpp) --------------------------------------------
// features: bit 0 = shared-subscription, bit 1 = limit-subscription.  The error of the last failed call of
// this thread is kept for rs_last_error.
static thread_local std::string g_rs_error;
static raft::Features rs_features(uint32_t bits) { raft::Features f; f.shared_subscription = bits & 1; f.limit_subscription = bits & 2; return f; }
static std::string hex(const std::string& s) {
    static const char* d = "0123456789abcdef";
    std::string o;
    for (unsigned char c : s) { o.push_back(d[c >> 4]); o.push_back(d[c & 15]); }
    return o.empty() ? "-" : o;
}
static std::string id_text(const Id& id) {
    return std::to_string(id.node_id) + "\t" + std::to_string(id.lid) + "\t" + (id.local_addr.empty() ? "-" : id.local_addr) + "\t" +
           (id.remote_addr.empty() ? "-" : id.remote_addr) + "\t" + hex(id.client_id) + "\t" + hex(id.username) + "\t" + std::to_string(id.create_time);
}
const char* rs_last_error() { return g_rs_error.c_str(); }
// plain bytes of one compressed section (malloc'ed, *out_len bytes); NULL on error
uint8_t* rs_uncompress(const uint8_t* p, uint64_t n, int compression, uint64_t* out_len) {
    auto r = raft::uncompress(raft::Compression(compression), p, size_t(n));
    if (!r.ok()) { g_rs_error = r.error; return nullptr; }
    *out_len = r.value->size();
    auto* o = static_cast<uint8_t*>(std::malloc(r.value->size() + 1));
    if (!r.value->empty()) std::memcpy(o, r.value->data(), r.value->size());
    return o;
}
// canonical text of a decoded snapshot, rows in wire order (strings as hex, "-" = empty / None):
//   F <n_filters>
//   R <filter> <key> <id...> <3|5> <qos> <group> <limit_subs> <no_local> <rap> <rh> <sub_ident>
//   C <key> <id...> <online> <handshaking> <handshak_duration>
//   T <count> <max> <merge_mode>            (topics_count)      N ... (relations_count)
char* rs_decode_dump(const uint8_t* p, uint64_t n, int compression, uint32_t features) {
    auto r = raft::decode_snapshot(p, size_t(n), raft::Compression(compression), rs_features(features));
    if (!r.ok()) { g_rs_error = r.error; return nullptr; }
    const raft::Snapshot& s = *r.value;
    std::string o = "F\t" + std::to_string(s.n_filters) + "\n";
    for (auto& x : s.relations) {
        o += "R\t" + hex(x.topic_filter) + "\t" + hex(x.client_id) + "\t" + id_text(x.id) + "\t" + (x.opts.v5 ? "5" : "3") + "\t" + std::to_string(x.opts.qos) +
             "\t" + (x.opts.shared_group ? hex(*x.opts.shared_group) + "." : std::string("-")) + "\t" + (x.limit_subs ? std::to_string(*x.limit_subs) : std::string("-")) +
             "\t" + std::to_string(int(x.opts.no_local)) + "\t" + std::to_string(int(x.opts.retain_as_published)) + "\t" + std::to_string(x.opts.retain_handling) +
             "\t" + (x.opts.subscription_identifier ? std::to_string(x.opts.subscription_identifier) : std::string("-")) + "\n";
    }
    for (auto& c : s.client_states)
        o += "C\t" + hex(c.client_id) + "\t" + id_text(c.id) + "\t" + std::to_string(int(c.online)) + "\t" + std::to_string(int(c.handshaking)) + "\t" +
             std::to_string(c.handshak_duration) + "\n";
    o += "T\t" + std::to_string(s.topics_count.count) + "\t" + std::to_string(s.topics_count.max) + "\t" + std::to_string(s.topics_count.merge_mode) + "\n";
    o += "N\t" + std::to_string(s.relations_count.count) + "\t" + std::to_string(s.relations_count.max) + "\t" + std::to_string(s.relations_count.merge_mode) + "\n";
    return dup_str(o);
}
// ClusterRouter::restore on the mirror: 0 ok, -1 error (rs_last_error)
int hr_restore_raft(void* r, const uint8_t* p, uint64_t n, int compression, uint32_t features) {
    auto s = raft::decode_snapshot(p, size_t(n), raft::Compression(compression), rs_features(features));
    if (!s.ok()) { g_rs_error = s.error; return -1; }
    auto res = static_cast<GpuRouter*>(r)->restore(*s.value);
    if (!res.ok()) { g_rs_error = res.error; return -1; }
    return 0;
}

int64_t hr_topics(void* r) { return static_cast<GpuRouter*>(r)->topics().count; }
int64_t hr_routes(void* r) { return static_cast<GpuRouter*>(r)->routes().count; }
uint64_t hr_topics_tree(void* r) { return static_cast<GpuRouter*>(r)->topics_tree(); }
}

// ---------------------------------------------------------------------------------------------- Shared::forwards (gpu_shared.hpp)
namespace {
// a session whose channel writes into its Shared's log: "<client>\t<qos'>\t<retain'>\t<subscription ids in arrival order,|->"
struct LogShared;
struct LogTx final : Tx {
    LogShared* owner; ClientId client; bool closed = false;
    bool unbounded_send(const From&, Publish&& p) override;
};
struct LogShared {
    Sessions sessions;
    std::unique_ptr<DefaultShared> inner;
    std::unique_ptr<GpuShared> gpu;
    std::mutex m;
    std::vector<std::string> log;
    // one From per publisher for as long as the Shared lives, as a session keeps its own: the owner id cached in it (From::owner_hint) has to survive
    // the publisher's own subscribes / unsubscribes between two publishes (tests/test_host_router.py)
    std::mutex from_m;
    std::map<std::string, std::unique_ptr<From>> froms;
    const From& from_of(const Id& id) {
        std::string key = std::to_string(id.node_id) + '|' + std::to_string(id.lid) + '|' + std::to_string(id.create_time) + '|' + id.client_id + '|' + id.username + '|' + id.local_addr + '|' + id.remote_addr;
        std::lock_guard<std::mutex> g(from_m);
        auto& f = froms[key];
        if (!f) f = std::make_unique<From>(id);
        return *f;
    }
};
bool LogTx::unbounded_send(const From&, Publish&& p) {
    if (closed) return false;
    std::string ids = "-";
    if (!p.subscription_ids.empty()) { ids.clear(); for (size_t i = 0; i < p.subscription_ids.size(); ++i) { if (i) ids.push_back(','); ids += std::to_string(p.subscription_ids[i]); } }
    std::lock_guard<std::mutex> g(owner->m);
    owner->log.push_back(client + "\t" + std::to_string(p.qos) + "\t" + std::to_string(int(p.retain)) + "\t" + ids + "\n");
    return true;
}
// every recipient is reachable and only counted (the bench: what the oracle's forwards_shaped ends at is the row, not a channel)
struct CountingTx final : Tx {
    static constexpr uint32_t kLanes = 128;
    struct alignas(64) Lane { std::atomic<uint64_t> n{0}, qos_sum{0}; char pad[48]; };
    Lane lane[kLanes];
    bool unbounded_send(const From&, Publish&& p) override {
        static std::atomic<uint32_t> next{0};
        thread_local const uint32_t mine = next.fetch_add(1, std::memory_order_relaxed) % kLanes;
        lane[mine].n.fetch_add(1, std::memory_order_relaxed);
        lane[mine].qos_sum.fetch_add(uint64_t(p.qos) + (p.retain ? 4u : 0u), std::memory_order_relaxed);
        return true;
    }
    uint64_t total() const { uint64_t t = 0; for (const Lane& l : lane) t += l.n.load(); return t; }
    uint64_t checksum() const { uint64_t t = 0; for (const Lane& l : lane) t += l.qos_sum.load(); return t; }
};
struct CountingShared final : Shared {
    CountingTx sink;
    Router& router;
    explicit CountingShared(Router& r) : router(r) {}
    Tx* tx(const ClientId&) override { return &sink; }
    Result<ForwardedCount> forwards(const From& from, const Publish& publish, std::vector<Undelivered>*) override {      // the host path of a publish the device handed back
        auto m = router.matches(from.id, *publish.topic);
        ForwardedCount n = 0;
        if (m.ok()) for (auto& kv : *m.value) for (auto& r : kv.second) { Publish p = publish; p.qos = std::min(p.qos, r.opts.qos); sink.unbounded_send(from, std::move(p)); ++n; }
        return Result<ForwardedCount>::Ok(n);
    }
};
}  // namespace

extern "C" {
// A Shared pair over router `r`: `inner` = the reference's forwards over Router::matches, `gpu` = GpuShared in front of it.
void* hr_shared_new(void* r, uint64_t this_node, uint32_t max_batch, uint32_t max_delay_us) {
    auto* router = static_cast<GpuRouter*>(r);
    auto* s = new LogShared;
    s->inner = std::make_unique<DefaultShared>(*router, s->sessions, this_node);
    s->gpu = std::make_unique<GpuShared>(*router, *s->inner, max_batch ? max_batch : 64, std::chrono::microseconds(max_delay_us), 2, 0);
    return s;
}
void hr_shared_free(void* sh) { delete static_cast<LogShared*>(sh); }
// closed != 0: the session exists but its channel is closed ("Connection Tx is closed")
void hr_shared_connect(void* sh, const char* client, uint32_t len, int closed) {
    auto* s = static_cast<LogShared*>(sh);
    auto tx = std::make_shared<LogTx>();
    tx->owner = s; tx->client.assign(client, len); tx->closed = closed != 0;
    s->sessions.connect(tx->client, tx);
}
void hr_shared_disconnect(void* sh, const char* client, uint32_t len) { static_cast<LogShared*>(sh)->sessions.disconnect(std::string(client, len)); }
// One publish through Shared::forwards: use_gpu 0 = the reference path (DefaultShared over Router::matches), 1 = GpuShared.  Returns the sorted log of what the
// sessions were sent, then "= <count>\n", then one "! <client>\t<reason>\n" per undelivered relation (sorted); NULL on Err.
char* hr_shared_forwards(void* sh, int use_gpu, const hr_id* from_id, const char* topic, uint64_t len, uint8_t qos, uint8_t retain, const char* target, uint32_t target_len) {
    auto* s = static_cast<LogShared*>(sh);
    const From& from = s->from_of(mk_id(from_id));
    Publish p;
    p.topic = std::make_shared<const TopicName>(topic, len);
    p.qos = qos; p.retain = retain != 0; p.dup = true; p.packet_id = 77;
    if (target) p.target_clientid = std::string(target, target_len);
    { std::lock_guard<std::mutex> g(s->m); s->log.clear(); }
    std::vector<Undelivered> errs;
    Shared& which = use_gpu ? static_cast<Shared&>(*s->gpu) : static_cast<Shared&>(*s->inner);
    auto r = which.forwards(from, p, &errs);
    if (!r.ok()) return nullptr;
    std::vector<std::string> log;
    { std::lock_guard<std::mutex> g(s->m); log.swap(s->log); }
    std::sort(log.begin(), log.end());
    std::string out;
    for (auto& l : log) out += l;
    out += "= " + std::to_string(*r.value) + "\n";
    std::vector<std::string> es;
    for (auto& e : errs) es.push_back("! " + e.to) ;
    for (size_t i = 0; i < errs.size(); ++i) es[i] += "\t" + errs[i].reason + "\n";
    std::sort(es.begin(), es.end());
    for (auto& e : es) out += e;
    return dup_str(out);
}
void hr_shared_counters(void* sh, uint64_t* out /* [5] device_path, host_path, deliveries, remote, passes (resubmissions: hr_forwards_run_async out[12]) */) {
    const auto c = static_cast<LogShared*>(sh)->gpu->counters();
    out[0] = c.device_path; out[1] = c.host_path; out[2] = c.deliveries; out[3] = c.remote; out[4] = c.passes;
}

// hr_restore_bulk with the delivery stage's flags per relation (RGR_SUB_V5 | RGR_SUB_NO_LOCAL | RGR_SUB_RAP bits, bench.py deliver_flags)
int hr_restore_bulk_ex(void* r, const uint8_t* blob, const uint64_t* offs, const uint32_t* client, const uint8_t* qos, const uint8_t* flags, uint64_t n) {
    raft::Snapshot snap;
    snap.relations.reserve(n);
    for (uint64_t i = 0; i < n; ++i) {
        raft::Relation rel;
        rel.topic_filter.assign(reinterpret_cast<const char*>(blob) + offs[i], offs[i + 1] - offs[i]);
        rel.client_id = "c" + std::to_string(client[i]);
        rel.id.node_id = 1; rel.id.client_id = rel.client_id;
        rel.opts.qos = qos ? qos[i] : 0;
        if (flags) { rel.opts.v5 = (flags[i] & RGR_SUB_V5) != 0; rel.opts.no_local = rel.opts.v5 && (flags[i] & RGR_SUB_NO_LOCAL); rel.opts.retain_as_published = rel.opts.v5 && (flags[i] & RGR_SUB_RAP); }
        snap.relations.push_back(std::move(rel));
    }
    return static_cast<GpuRouter*>(r)->restore(snap).ok() ? 0 : -1;
}

// Shared::forwards at its design point (the asynchronous shape of hr_e2e_run_async): `n_submitters` threads keep `outstanding` publishes in flight through
// GpuShared::submit; every publish goes through ONE delivery pass of its batch and from delivery words straight to the (counting) sessions on `workers` pool
// threads.  Publisher of topic i = client "c<from_client[i]>" of node 1, publish qos / retain = qos_retain[i] (bench.py's delivery workload).
// out[0] = publishes completed, out[1] = recipients reached, out[2] = device passes, out[3] = errors, out[4] = publishes that took the host path,
// out[5] = checksum of the delivered (qos', retain') pairs.
int hr_forwards_run_async(void* r, const uint8_t* blob, const uint64_t* offs, uint32_t n, const uint32_t* from_client, const uint8_t* qos_retain, uint32_t n_submitters,
                          uint32_t outstanding, uint32_t workers, uint32_t passes_in_flight, uint32_t max_batch, uint32_t max_delay_us, double seconds, uint64_t* out,
                          double* wall_s, float* lat_us, uint32_t n_lat, uint32_t* n_lat_out) {
    auto* router = static_cast<GpuRouter*>(r);
    CountingShared inner(*router);
    // the publishes and their publishers, built once (a broker holds them as it decodes them)
    std::vector<Publish> pubs(n);
    std::vector<From> froms(n);
    for (uint32_t i = 0; i < n; ++i) {
        pubs[i].topic = std::make_shared<const TopicName>(reinterpret_cast<const char*>(blob) + offs[i], offs[i + 1] - offs[i]);
        pubs[i].qos = qos_retain[i] & 3u; pubs[i].retain = (qos_retain[i] & 4u) != 0;
        froms[i].id.node_id = from_client[i] == 0xFFFFFFFFu ? 0 : 1; froms[i].id.client_id = "c" + std::to_string(from_client[i]);
    }
    constexpr uint32_t kLanes = 128;
    struct alignas(64) Lane { std::atomic<uint64_t> pubs{0}, rows{0}, errs{0}; char pad[40]; };
    struct alignas(64) Ctx { Lane lane[kLanes]; float* lat = nullptr; uint32_t n_lat = 0; std::atomic<uint32_t> lat_n{0}; std::chrono::steady_clock::time_point t0;
                             uint64_t done() const { uint64_t d = 0; for (const Lane& l : lane) d += l.pubs.load(std::memory_order_acquire); return d; } };
    std::vector<std::unique_ptr<Ctx>> ctx;
    for (uint32_t k = 0; k < n_submitters; ++k) ctx.push_back(std::make_unique<Ctx>());
    const uint64_t cap = std::max<uint64_t>(1, outstanding / std::max(1u, n_submitters));
    std::atomic<bool> stop{false};
    const auto t0 = std::chrono::steady_clock::now();
    for (auto& c : ctx) c->t0 = t0;
    ctx[0]->lat = lat_us; ctx[0]->n_lat = lat_us ? n_lat : 0;
    const GpuShared::Done done = [](void* user, uint64_t tag, ForwardedCount count, const std::string* err) {
        static std::atomic<uint32_t> next_lane{0};
        thread_local const uint32_t my_lane = next_lane.fetch_add(1, std::memory_order_relaxed) % kLanes;
        Ctx& c = *static_cast<Ctx*>(user);
        Lane& l = c.lane[my_lane];
        if (err) l.errs.fetch_add(1, std::memory_order_relaxed);
        if (count) l.rows.fetch_add(count, std::memory_order_relaxed);
        if (c.lat) {
            const uint32_t j = c.lat_n.fetch_add(1, std::memory_order_relaxed);
            if (j < c.n_lat) c.lat[j] = float(double(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - c.t0).count() - int64_t(tag)) / 1e3);
        }
        l.pubs.fetch_add(1, std::memory_order_release);
    };
    double wall = 0;
    GpuShared::Counters cnt{};
    Batcher::Timing tm{};
    {
        GpuShared gs(*router, inner, max_batch, std::chrono::microseconds(max_delay_us), passes_in_flight, workers);
        std::vector<std::thread> th;
        for (uint32_t k = 0; k < n_submitters; ++k)
            th.emplace_back([&, k] {
                Ctx& c = *ctx[k];
                uint64_t submitted = 0, seen = 0;
                for (uint64_t i = k; !stop.load(std::memory_order_relaxed); i += n_submitters) {
                    while (submitted - seen >= cap) {
                        seen = c.done();
                        if (submitted - seen >= cap) { if (stop.load(std::memory_order_relaxed)) break; std::this_thread::yield(); }
                    }
                    const uint32_t t = uint32_t(i % n);
                    const uint64_t tag = uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
                    gs.submit(&froms[t], &pubs[t], done, &c, tag);
                    ++submitted;
                }
                while (c.done() < submitted) std::this_thread::yield();      // drain: every publish completes before the Shared goes away
            });
        std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
        stop = true;
        for (auto& t : th) t.join();
        wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        cnt = gs.counters();
        tm = gs.timing();
    }
    if (wall_s) *wall_s = wall;
    uint64_t p = 0, rws = 0, errs = 0;
    for (auto& c : ctx) for (const auto& l : c->lane) { p += l.pubs.load(); rws += l.rows.load(); errs += l.errs.load(); }
    out[0] = p; out[1] = rws; out[2] = cnt.passes; out[3] = errs; out[4] = cnt.host_path; out[5] = inner.sink.checksum();
    // out[6..11]: the batcher's clocks, as hr_e2e_run_async reports them
    out[6] = tm.collect_ns; out[7] = tm.pass_ns; out[8] = tm.dispatch_ns; out[9] = tm.task_ns; out[10] = tm.tasks; out[11] = tm.max_task_queue;
    out[12] = cnt.resubmitted;      // publishes that joined another batch because a removal overtook their pass
    if (n_lat_out) *n_lat_out = std::min<uint32_t>(ctx[0]->lat_n.load(), ctx[0]->n_lat);
    return 0;
}
}  // extern "C"
