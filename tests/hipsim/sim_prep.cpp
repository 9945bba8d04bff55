// TEST INFRASTRUCTURE ONLY — count_batched_kernel / compact_batched_kernel of rmqtt_amd/csrc/prep_batched.inc on the host (hipsim.hpp)
// against the per-topic functions the emulator and the product kernels share (count_topic / compact_topic, match_core.hpp).
#include "hipsim.hpp"

#include "kernels.hpp"
#include "match_core.hpp"

namespace rgr {
namespace {
constexpr int kCompactStage = 512;
constexpr int kCompactWave = 64;
#include "prep_batched.inc"
}  // namespace
}  // namespace rgr

using namespace rgr;

extern "C" {

// One chunk: count (batched kernel), the two scans on the host, compact (batched kernel); the same with count_topic / compact_topic.
// Topics with more than kBigPairs matched filters belong to the *_big kernels: counted here by count_topic, their pair ranges are
// left out of the comparison.  Returns the number of differing words (0 = equal), -2 on divergence.
int64_t sim_prep(uint32_t n, uint32_t slot_cap, const uint32_t* slots, const uint32_t* pair_cnt, const uint64_t* ovf_base, const uint32_t* arena, uint64_t arena_cap,
                 const FilterDesc* filt, uint32_t topic_base, const PublishAttr* pub, uint64_t* n_pairs_out, uint64_t* n_hits_out) {
    TrieView tv{};
    tv.filt = filt;
    std::vector<uint32_t> hit_cnt(n, 0xDEADBEEF), pair_live(n, 0xDEADBEEF), big_list(n), hit_cnt_ref(n), pair_live_ref(n);
    std::vector<uint64_t> hit_off(size_t(n) + 1), pair_base(size_t(n) + 1);
    uint32_t big_count = 0, err = 0, err_ref = 0;
    ChunkArrays c{};
    c.n = n; c.slot_cap = slot_cap; c.slots = slots; c.pair_cnt = pair_cnt; c.hit_cnt = hit_cnt.data(); c.pair_live = pair_live.data();
    c.hit_off = hit_off.data(); c.pair_base = pair_base.data(); c.ovf_base = ovf_base; c.ovf_arena = arena; c.ovf_arena_cap = arena_cap;
    c.error_flag = &err; c.big_list = big_list.data(); c.big_count = &big_count; c.pub = pub;
    bool ok = hipsim::run((n + 255) / 256, 256, [&] { count_batched_kernel(tv, c); });
    for (uint32_t b = 0; b < big_count; ++b) count_topic(tv, c, big_list[b]);                 // (count_big_kernel's job)
    ChunkArrays r = c;
    r.hit_cnt = hit_cnt_ref.data(); r.pair_live = pair_live_ref.data(); r.error_flag = &err_ref;
    for (uint32_t t = 0; t < n; ++t) count_topic(tv, r, t);
    int64_t diff = err != err_ref;
    for (uint32_t t = 0; t < n; ++t) diff += (hit_cnt[t] != hit_cnt_ref[t]) + (pair_live[t] != pair_live_ref[t]);
    hit_off[0] = pair_base[0] = 0;
    for (uint32_t t = 0; t < n; ++t) { hit_off[t + 1] = hit_off[t] + hit_cnt_ref[t]; pair_base[t + 1] = pair_base[t] + pair_live_ref[t]; }
    const uint64_t P = pair_base[n];
    std::vector<uint32_t> src(P + 1, 0xAAAAAAAA), topic(P + 1, 0xAAAAAAAA), src_ref(P + 1, 0xAAAAAAAA), topic_ref(P + 1, 0xAAAAAAAA);
    std::vector<uint64_t> off(P + 2, ~0ull), off_ref(P + 2, ~0ull);
    std::vector<uint8_t> qr(P + 1, 0xEE), qr_ref(P + 1, 0xEE);
    c.hit_cnt = hit_cnt_ref.data(); c.pair_live = pair_live_ref.data();
    c.pair_src = src.data(); c.pair_topic = topic.data(); c.pair_off = off.data(); c.pair_qr = pub ? qr.data() : nullptr;
    ok &= hipsim::run((n + kCompactWave - 1) / kCompactWave, kCompactWave, [&] { compact_batched_kernel(tv, c, topic_base); });
    r.hit_cnt = hit_cnt_ref.data(); r.pair_live = pair_live_ref.data();
    r.pair_src = src_ref.data(); r.pair_topic = topic_ref.data(); r.pair_off = off_ref.data(); r.pair_qr = pub ? qr_ref.data() : nullptr;
    for (uint32_t t = 0; t < n; ++t) compact_topic(tv, r, topic_base, t);
    for (uint32_t t = 0; t < n; ++t) {
        if (pair_cnt[t] > kBigPairs) continue;
        for (uint64_t p = pair_base[t]; p < pair_base[t + 1]; ++p)
            diff += (src[p] != src_ref[p]) + (topic[p] != topic_ref[p]) + (off[p] != off_ref[p]) + (pub && qr[p] != qr_ref[p]);
    }
    diff += off[P] != off_ref[P];                                                            // the sentinel: total hits of the chunk
    if (n_pairs_out) *n_pairs_out = P;
    if (n_hits_out) *n_hits_out = hit_off[n];
    return ok ? diff : -2;
}

}  // extern "C"
