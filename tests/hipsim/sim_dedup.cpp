// TEST INFRASTRUCTURE ONLY — the v5 per-client dedup kernels of rmqtt_amd/csrc/dedup.inc run on the host (hipsim.hpp): the tile pass,
// (with the classification in the same launch) and the topic pass over one window's candidate lists, as
// launch_dedup chains them.  tests/test_hipsim_dedup.py compares the flags with a first-position map built in numpy.
#include "hipsim.hpp"

#include "kernels.hpp"
#include "match_core.hpp"

namespace rgr {
namespace {
constexpr int kTile = 2048;
#include "dedup.inc"
}  // namespace
}  // namespace rgr

using namespace rgr;

extern "C" {

// variant 3: dedup_topic_kernel<3> (the product: double hashing, 16-byte clears), 0: dedup_topic_kernel<0> (RGR_DEDUP_PROBE=0).  grid_topic:
// blocks of the topic pass (the product launches 1024; fewer blocks make every block walk several items).  Returns 0, -2 when threads diverged around a barrier.
int32_t sim_dedup(int32_t variant, uint32_t grid_topic, uint32_t max_slots, const Cand* cand, const uint32_t* tile_ncand, const uint32_t* tile_trange,
                  uint32_t ntiles, Tuple* tuples, uint32_t nt, const uint64_t* hit_off, uint64_t hit_lo, uint32_t* n_items_out) {
    std::vector<DedupItem> items(size_t(nt) + (hit_off[nt] - hit_lo) / kDedupTopicCap + 2);
    uint32_t item_counts[2] = {0xDEADu, 0};      // window parity 1: [1] is this window's counter (zero when the pass begins), [0] the next window's (zeroed by block 0)
    unsigned long long stat[2048] = {0};
    bool ok = true;
    const uint32_t tile_blocks = ntiles < 64 ? ntiles : 64;
    // tile pass + classification: one launch (blocks behind the tile pass's classify 256 window topics each)
    ok &= hipsim::run(tile_blocks + (nt + 255) / 256, 256, [&] { dedup_tile_kernel(cand, tile_ncand, tile_trange, ntiles, hit_off, hit_lo, nt, tuple_words(tuples), stat, tile_blocks, items.data(), item_counts, 1u, max_slots, 2u); });
    if (item_counts[0] != 0) return -4;          // the next window's counter was not zeroed
    uint32_t& item_count = item_counts[1];
    if (variant != 0 && variant != 3 && variant != 7) return -3;        // variant = the topic pass's probe_mode (bit 0: double hashing, bit 1: 16-byte clears)
    if (variant == 7) ok &= hipsim::run(grid_topic, kDedupTopicThreads, [&] { dedup_topic_kernel<7>(cand, tile_ncand, items.data(), &item_count, tuple_words(tuples)); });
    else if (variant == 3) ok &= hipsim::run(grid_topic, kDedupTopicThreads, [&] { dedup_topic_kernel<3>(cand, tile_ncand, items.data(), &item_count, tuple_words(tuples)); });
    else ok &= hipsim::run(grid_topic, kDedupTopicThreads, [&] { dedup_topic_kernel<0>(cand, tile_ncand, items.data(), &item_count, tuple_words(tuples)); });
    if (n_items_out) *n_items_out = item_count;
    return ok ? 0 : -2;
}

}  // extern "C"
