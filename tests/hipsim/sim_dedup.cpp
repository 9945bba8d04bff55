// TEST INFRASTRUCTURE ONLY — the v5 per-client dedup kernels of rmqtt_amd/csrc/dedup.inc run on the host (hipsim.hpp): the tile pass,
// the classification and the topic pass over one window's candidate lists, as
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

// variant 0: dedup_topic_kernel (the only one left: the pipelined and the batched variants were measured slower and dropped).  grid_topic:
// blocks of the topic pass (the product launches 1024; fewer blocks make every block walk several items).  Returns 0, -2 when threads diverged around a barrier.
int32_t sim_dedup(int32_t variant, uint32_t grid_topic, uint32_t max_slots, const Cand* cand, const uint32_t* tile_ncand, const uint32_t* tile_trange,
                  uint32_t ntiles, Tuple* tuples, uint32_t nt, const uint64_t* hit_off, uint64_t hit_lo, uint32_t* n_items_out) {
    std::vector<DedupItem> items(size_t(nt) + (hit_off[nt] - hit_lo) / kDedupTopicCap + 2);
    uint32_t item_count = 0;
    unsigned long long stat = 0;
    bool ok = true;
    ok &= hipsim::run(ntiles < 64 ? ntiles : 64, 256, [&] { dedup_tile_kernel(cand, tile_ncand, tile_trange, ntiles, hit_off, hit_lo, nt, tuples, &stat); });
    ok &= hipsim::run((nt + 255) / 256, 256, [&] { dedup_classify_kernel(tile_ncand, nt, hit_off, hit_lo, items.data(), &item_count); });
    if (variant != 0) return -3;
    ok &= hipsim::run(grid_topic, kDedupTopicThreads, [&] { dedup_topic_kernel(cand, tile_ncand, hit_off, hit_lo, items.data(), &item_count, tuples, max_slots); });
    if (n_items_out) *n_items_out = item_count;
    return ok ? 0 : -2;
}

}  // extern "C"
