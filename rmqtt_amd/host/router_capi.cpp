// ctypes-facing shim over rmqtt::GpuRouter so that the Python parity tests can drive the C++
// Router mirror and compare its SubRelationsMap with the oracle's DefaultRouter, in the
// canonical text form of SURVEY.md App. A.5 (see oracle.cpp: orc_router_matches).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>

#include "gpu_retain.hpp"
#include "gpu_router.hpp"

using namespace rmqtt;

namespace {
char* dup_str(const std::string& s) {
    char* p = static_cast<char*>(std::malloc(s.size() + 1));
    std::memcpy(p, s.data(), s.size());
    p[s.size()] = 0;
    return p;
}
struct hr_id { uint64_t node_id; const char* client_id; uint32_t client_len; int64_t create_time; uint16_t lid; };
struct hr_opts { uint8_t v5, qos, no_local, retain_as_published, retain_handling; uint32_t sub_ident; };
Id mk_id(const hr_id* i) { Id id; id.node_id = i->node_id; id.lid = i->lid; id.create_time = i->create_time; id.client_id.assign(i->client_id, i->client_len); return id; }
SubscriptionOptions mk_opts(const hr_opts* o) {
    SubscriptionOptions s; s.v5 = o->v5; s.qos = o->qos; s.no_local = o->no_local; s.retain_as_published = o->retain_as_published;
    s.retain_handling = o->retain_handling; s.subscription_identifier = o->sub_ident; return s;
}
std::string dump(const SubRelationsMap& m) {
    std::string out;
    for (auto& kv : m) {
        out += "N " + std::to_string(kv.first) + "\n";
        std::vector<std::string> v3, v5;
        for (auto& s : kv.second) {
            if (s.opts.is_v3()) v3.push_back("3 " + s.topic_filter + "\t" + s.client_id + "\t" + std::to_string(s.opts.qos) + "\n");
            else {
                std::string ids = "-";
                if (s.sub_ids) {
                    auto v = *s.sub_ids; std::sort(v.begin(), v.end()); ids.clear();
                    for (size_t i = 0; i < v.size(); ++i) { if (i) ids.push_back(','); ids += std::to_string(v[i]); }
                }
                v5.push_back("5 " + s.client_id + "\t" + s.topic_filter + "\t" + std::to_string(s.opts.qos) + "\t" + std::to_string(int(s.opts.no_local)) + "\t" + ids + "\n");
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

int64_t hr_topics(void* r) { return static_cast<GpuRouter*>(r)->topics().count; }
int64_t hr_routes(void* r) { return static_cast<GpuRouter*>(r)->routes().count; }
uint64_t hr_topics_tree(void* r) { return static_cast<GpuRouter*>(r)->topics_tree(); }
}
