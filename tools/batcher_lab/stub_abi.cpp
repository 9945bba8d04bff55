// Batcher lab (host-side microbenchmark, not product, not test infrastructure of the parity suite): a STUB of the C ABI so that the
// C++ twin's Batcher / GpuRouter host machinery can be profiled on a box without a GPU.  rgr_group_match_filter_subs sleeps for
// RGR_STUB_PASS_US microseconds (default 450: what a 1-4 k topic pass costs on an MI355X, profiles/r04d_router_e2e_*) and answers
// "no matched filters".  Everything else is a no-op that succeeds.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "rmqtt_gpu_router.h"

namespace {
struct Own { std::vector<int32_t> status; std::vector<uint64_t> offs; };
int pass_us() { static const int v = [] { const char* e = std::getenv("RGR_STUB_PASS_US"); return e ? std::atoi(e) : 450; }(); return v; }
}
extern "C" {
const char* rgr_last_error(void) { return "stub"; }
int32_t rgr_create(const rgr_config*, rgr_handle** out) { *out = reinterpret_cast<rgr_handle*>(new int(1)); return RGR_OK; }
void rgr_destroy(rgr_handle* h) { delete reinterpret_cast<int*>(h); }
int32_t rgr_group_create(const rgr_config*, const int32_t*, uint32_t, rgr_group** out) { *out = reinterpret_cast<rgr_group*>(new int(1)); return RGR_OK; }
void rgr_group_destroy(rgr_group* g) { delete reinterpret_cast<int*>(g); }
uint32_t rgr_group_size(const rgr_group*) { return 1; }
int32_t rgr_group_subscribe_bulk(rgr_group*, const uint8_t*, const uint64_t*, uint64_t, const uint32_t*, const uint8_t*, const uint8_t*, uint64_t* rej) { if (rej) *rej = 0; return RGR_OK; }
int32_t rgr_group_sub_attrs_bulk(rgr_group*, const uint32_t*, const uint32_t*, const uint32_t*, uint64_t) { return RGR_OK; }
int32_t rgr_group_subscribe_ex(rgr_group*, const char*, uint32_t, uint32_t, uint8_t, uint8_t, uint16_t, uint32_t, uint32_t) { return RGR_OK; }
int32_t rgr_group_unsubscribe(rgr_group*, const char*, uint32_t, uint32_t, int32_t) { return RGR_OK; }
int32_t rgr_group_commit(rgr_group*) { return RGR_OK; }
int32_t rgr_group_match_filter_subs(rgr_group*, const uint8_t*, const uint64_t*, uint32_t n, rgr_filters_result* out) {
    std::this_thread::sleep_for(std::chrono::microseconds(pass_us()));
    auto* o = new Own;
    o->status.assign(n, RGR_TOPIC_OK);
    o->offs.assign(size_t(n) + 1, 0);
    std::memset(out, 0, sizeof *out);
    out->n_topics = n; out->status = o->status.data(); out->pair_offsets = o->offs.data(); out->_owner = o;
    return RGR_OK;
}
void rgr_filters_result_free(rgr_filters_result* r) { if (r && r->_owner) { delete static_cast<Own*>(r->_owner); std::memset(r, 0, sizeof *r); } }
int32_t rgr_group_match_batch(rgr_group*, const uint8_t*, const uint64_t*, uint32_t, rgr_result*) { return RGR_EDEVICE; }
int32_t rgr_group_match_batch_deliver(rgr_group*, const uint8_t*, const uint64_t*, uint32_t, const rgr_publish_attr*, rgr_result*) { return RGR_EDEVICE; }
int32_t rgr_group_match_batch_deliver_grouped(rgr_group*, const uint8_t*, const uint64_t*, uint32_t, const rgr_publish_attr*, rgr_result*, rgr_node_groups*) { return RGR_EDEVICE; }
void rgr_result_free(rgr_result*) {}
int32_t rgr_retain_topic_add(rgr_handle*, const char*, uint32_t, uint32_t) { return RGR_OK; }
int32_t rgr_retain_topic_remove(rgr_handle*, const char*, uint32_t) { return RGR_OK; }
int32_t rgr_retain_commit(rgr_handle*) { return RGR_OK; }
int32_t rgr_retain_match_ranges(rgr_handle*, const uint8_t*, const uint64_t*, uint32_t, rgr_retain_ranges*) { return RGR_EDEVICE; }
void rgr_retain_ranges_free(rgr_retain_ranges*) {}
}
