// TEST INFRASTRUCTURE ONLY — the compact expansions of rmqtt_amd/csrc/expand_compact.inc run on the host (hipsim.hpp).
//
// One window: the pair arrays (source, chunk-local output offset) of its pairs, the tile records tiles_kernel leaves (the same
// tiles_pair_rec, match_core.hpp), then one of the expansion kernels block by block.  tests/test_hipsim_expand.py compares the
// result with a numpy expansion of the same pair list.
#include "hipsim.hpp"

#include "kernels.hpp"
#include "match_core.hpp"

namespace rgr {
namespace {
constexpr int kTile = 2048;
#define RGR_COMPACT_NT 0            // plain stores: a nontemporal vector store has alignment rules of its own on x86
#include "expand_compact.inc"
}  // namespace
}  // namespace rgr

using namespace rgr;

extern "C" {

// variant 0: expand_compact_kernel (one tile per block, or the round-4 pipelined form for tiles_per_block > 1)
// variant 1: expand_compact_lp_kernel (pairs held in lanes); variant 2: expand_ids24_x4_kernel (the same, IDS24 through 16-byte stores)
// fmt: 1 SOA, 2 PACKED, 4 IDS24.  packed may be null (8-byte entry reads; variant 0 only).  Returns 0, -1 for an unknown combination, -2 when threads diverged around a barrier or a cross-lane read.
int32_t sim_expand_compact(int32_t variant, int32_t fmt, int32_t tiles_per_block, const SubEntry* subs, const uint32_t* packed,
                           const uint32_t* pair_src, const uint32_t* pair_topic, const uint64_t* pair_off, uint64_t pair_lo, uint64_t pair_hi,
                           uint32_t* out_ids, uint8_t* out_qos) {
    ChunkArrays c{};
    c.pair_src = const_cast<uint32_t*>(pair_src);
    c.pair_topic = const_cast<uint32_t*>(pair_topic);
    c.pair_off = const_cast<uint64_t*>(pair_off);
    const uint64_t hit_lo = pair_off[pair_lo], hit_hi = pair_off[pair_hi];
    if (hit_hi <= hit_lo) return 0;
    const uint32_t ntiles = uint32_t((hit_hi - hit_lo + kTile - 1) / kTile);
    std::vector<TileRec> rec(ntiles);
    for (uint64_t p = pair_lo; p < pair_hi; ++p) tiles_pair_rec(c, p, pair_lo, hit_lo, kTile, rec.data());
    const TileRec* tf = rec.data();
    const uint32_t T = uint32_t(tiles_per_block), nb = (ntiles + T - 1) / T;
    bool converged = true;
#define SIM_RUN(K, F, TT) converged = hipsim::run(nb, kCompactThreads, [&] { K<F, TT>(subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tf, ntiles, out_ids, out_qos, packed); })
    if (variant == 0) {
        if (fmt == kFmtSoa && T == 1) SIM_RUN(expand_compact_kernel, kFmtSoa, 1);
        else if (fmt == kFmtPacked && T == 1) SIM_RUN(expand_compact_kernel, kFmtPacked, 1);
        else if (fmt == kFmtPacked && T == 4) SIM_RUN(expand_compact_kernel, kFmtPacked, 4);
        else if (fmt == kFmtIds24 && T == 1) SIM_RUN(expand_compact_kernel, kFmtIds24, 1);
        else if (fmt == kFmtIds24 && T == 4) SIM_RUN(expand_compact_kernel, kFmtIds24, 4);
        else return -1;
    } else if (variant == 1) {
        if (!packed) return -1;
        if (fmt == kFmtPacked && T == 1) SIM_RUN(expand_compact_lp_kernel, kFmtPacked, 1);
        else if (fmt == kFmtPacked && T == 2) SIM_RUN(expand_compact_lp_kernel, kFmtPacked, 2);
        else if (fmt == kFmtPacked && T == 4) SIM_RUN(expand_compact_lp_kernel, kFmtPacked, 4);
        else if (fmt == kFmtIds24 && T == 1) SIM_RUN(expand_compact_lp_kernel, kFmtIds24, 1);
        else if (fmt == kFmtIds24 && T == 2) SIM_RUN(expand_compact_lp_kernel, kFmtIds24, 2);
        else if (fmt == kFmtIds24 && T == 4) SIM_RUN(expand_compact_lp_kernel, kFmtIds24, 4);
        else return -1;
    } else if (variant == 2) {                  // IDS24 through 16-byte stores (two lane-held tiles per block)
        if (!packed || fmt != kFmtIds24) return -1;
        converged = hipsim::run((ntiles + 1) / 2, kCompactThreads, [&] { expand_ids24_x4_kernel(subs, c, pair_lo, pair_hi, hit_lo, hit_hi, tf, ntiles, out_ids, out_qos, packed); });
    } else {
        return -1;
    }
#undef SIM_RUN
    return converged ? 0 : -2;         // -2: lanes of a wave (or threads of a block) took different paths around a convergent operation
}

// RGR_TILES_FUSED: expand_ids24_lp_tiles_kernel over the window [pair_lo, pair_mid) while its tail blocks write the tile records of the
// next window [pair_mid, pair_hi).  Returns the number of those records that differ from tiles_pair_rec on the host (0 = equal), -2 on
// divergence; out_ids receives the first window's 3-byte ids.
int64_t sim_expand_ids24_fused(const SubEntry* subs, const uint32_t* packed, const uint32_t* pair_src, const uint32_t* pair_topic, const uint64_t* pair_off,
                               uint64_t pair_lo, uint64_t pair_mid, uint64_t pair_hi, uint32_t* out_ids) {
    ChunkArrays c{};
    c.pair_src = const_cast<uint32_t*>(pair_src);
    c.pair_topic = const_cast<uint32_t*>(pair_topic);
    c.pair_off = const_cast<uint64_t*>(pair_off);
    const uint64_t hit_lo = pair_off[pair_lo], hit_hi = pair_off[pair_mid], nhit_hi = pair_off[pair_hi];
    const uint32_t ntiles = uint32_t((hit_hi - hit_lo + kTile - 1) / kTile), ntiles_next = uint32_t((nhit_hi - hit_hi + kTile - 1) / kTile);
    std::vector<TileRec> rec(ntiles), want(ntiles_next, TileRec{0xDEADBEEF, 0, 0, 0}), got(ntiles_next, TileRec{0xDEADBEEF, 0, 0, 0});
    for (uint64_t p = pair_lo; p < pair_mid; ++p) tiles_pair_rec(c, p, pair_lo, hit_lo, kTile, rec.data());
    for (uint64_t p = pair_mid; p < pair_hi; ++p) tiles_pair_rec(c, p, pair_mid, hit_hi, kTile, want.data());
    const NextTiles nx{pair_mid, pair_hi, hit_hi, got.data()};
    const uint32_t nb_tiles = uint32_t((pair_hi - pair_mid + kCompactThreads - 1) / kCompactThreads);
    const TileRec* tf = rec.data();
    const bool ok = hipsim::run(ntiles + nb_tiles, kCompactThreads, [&] { expand_ids24_lp_tiles_kernel(subs, c, pair_lo, pair_mid, hit_lo, hit_hi, tf, ntiles, out_ids, nullptr, packed, nx); });
    int64_t diff = 0;
    for (uint32_t k = 0; k < ntiles_next; ++k) diff += got[k].first != want[k].first || got[k].src != want[k].src || got[k].topic != want[k].topic || got[k].qr != want[k].qr;
    return ok ? diff : -2;
}

}  // extern "C"
