// TEST INFRASTRUCTURE ONLY — the tuple expansions of rmqtt_amd/csrc/expand_tuple.inc run on the host (hipsim.hpp): the plain kernel,
// its delivery variant (delivery words + v5 dedup candidates per tile) and the delivery variant with its loads issued early
// (RGR_DELIVER_EARLY).  tests/test_hipsim_expand_tuple.py compares with a numpy restatement of the per-hit rules.
#include "hipsim.hpp"

#include "kernels.hpp"
#include "match_core.hpp"

namespace rgr {
namespace {
#define RGR_EXPAND_NT 0
constexpr int kExpandThreads = 1024, kExpandPerThread = 2;
constexpr int kTile = 2048;
#include "expand_tuple.inc"
}  // namespace
}  // namespace rgr

using namespace rgr;

extern "C" {

// variant 0: expand_kernel<false, 1024, 2> (tuples only), 1: expand_kernel<true, 512, 4>, 2: expand_deliver_early_kernel<512, 4>,
// 3: expand_deliver_lean_kernel<512, 4> (r5: v5 hits compacted per wave), 4: expand_deliver_lean_kernel<256, 8>, 5 / 6: the same two with 8-byte hits.
// cand / tile_ncand / tile_trange may be null (an epoch without v5 subscriptions).  Returns 0, -1 unknown variant, -2 divergence.
int32_t sim_expand_tuple(int32_t variant, const SubEntry* subs, const SubAttr* attrs, const PublishAttr* pub, const uint32_t* pair_src,
                         const uint32_t* pair_topic, const uint64_t* pair_off, const uint8_t* pair_qr, uint64_t pair_lo, uint64_t pair_hi, uint32_t topic_lo,
                         Tuple* out, Cand* cand, uint32_t* tile_ncand, uint32_t* tile_trange) {
    ChunkArrays c{};
    c.pair_src = const_cast<uint32_t*>(pair_src);
    c.pair_topic = const_cast<uint32_t*>(pair_topic);
    c.pair_off = const_cast<uint64_t*>(pair_off);
    c.pair_qr = const_cast<uint8_t*>(pair_qr);
    c.pub = pub;
    const uint64_t hit_lo = pair_off[pair_lo], hit_hi = pair_off[pair_hi];
    if (hit_hi <= hit_lo) return 0;
    const uint32_t ntiles = uint32_t((hit_hi - hit_lo + kTile - 1) / kTile);
    std::vector<TileRec> rec(ntiles);
    for (uint64_t p = pair_lo; p < pair_hi; ++p) tiles_pair_rec(c, p, pair_lo, hit_lo, kTile, rec.data());
    const TileRec* tf = rec.data();
    DeliverArgs da{};
    da.pub = pub; da.attrs = attrs; da.cand = cand; da.tile_ncand = tile_ncand; da.tile_trange = tile_trange; da.topic_lo = topic_lo;
    bool ok = true;
    if (variant == 0) ok = hipsim::run(ntiles, 1024, [&] { expand_kernel<false, 1024, 2>(subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tf, ntiles, out, DeliverArgs{}); });
    else if (variant == 1) ok = hipsim::run(ntiles, 512, [&] { expand_kernel<true, 512, 4>(subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tf, ntiles, out, da); });
    else if (variant == 2) ok = hipsim::run(ntiles, 512, [&] { expand_deliver_early_kernel<512, 4>(subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tf, ntiles, out, da); });
    else if (variant == 3) ok = hipsim::run(ntiles, 512, [&] { expand_deliver_lean_kernel<512, 4>(subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tf, ntiles, out, da); });
    else if (variant == 4) ok = hipsim::run(ntiles, 256, [&] { expand_deliver_lean_kernel<256, 8>(subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tf, ntiles, out, da); });
    // 5 / 6: the lean variants writing 8-byte hits {sub_id, delivery word} (RGR_FORMAT_DELIVER8) into the same buffer
    else if (variant == 5) ok = hipsim::run(ntiles, 512, [&] { expand_deliver_lean_kernel<512, 4, true>(subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tf, ntiles, out, da); });
    else if (variant == 6) ok = hipsim::run(ntiles, 256, [&] { expand_deliver_lean_kernel<256, 8, true>(subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tf, ntiles, out, da); });
    else return -1;
    return ok ? 0 : -2;
}

}  // extern "C"
