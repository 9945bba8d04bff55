// TEST INFRASTRUCTURE ONLY — the v5 dedup passes of rmqtt_amd/csrc/dedup.inc on the host under ThreadSanitizer and AddressSanitizer
// (tools/hipsim_sanitizers.sh).  TSAN: the topic pass clears, fills and re-clears ONE table in LDS between block barriers — a clear that overtakes a
// probe, or the overflow flag reset before every thread has read it, is a data race between the OS threads that stand for GPU threads.  ASAN: the
// candidate slices have exactly kTile entries per tile (the list heads are read past a tile's count, never past its slice), the item
// array its exact upper bound.  Also compares every form's flags with a first-position map.
#include "sim_dedup.cpp"

#include <algorithm>
#include <cstdio>
#include <map>
#include <random>

int main() {
    std::mt19937_64 rng(20260930);
    int bad = 0;
    // worlds: (topic sizes, v5 share, clients) — topics spanning many tiles (several parts with small tables), topics inside tiles, a mix
    struct World { std::vector<uint32_t> sizes; double share; uint32_t clients; uint32_t grid, slots; };
    std::vector<World> worlds;
    worlds.push_back({{30000, 5, 2043, 9000, 1, 1, 700, 12000}, 0.3, 900, 3, 4096});
    worlds.push_back({{20000, 18000}, 0.5, 3000, 2, 256});                       // tiny tables: parts overflow, the re-split path runs
    { World w{{}, 0.2, 300, 5, 4096}; for (int i = 0; i < 300; ++i) w.sizes.push_back(1 + uint32_t(rng() % 400)); w.sizes.push_back(7000); worlds.push_back(w); }
    for (size_t wi = 0; wi < worlds.size(); ++wi) {
        const World& w = worlds[wi];
        const uint32_t nt = uint32_t(w.sizes.size());
        const uint64_t hit_lo = (1ull << 34) + 77;
        std::vector<uint64_t> hit_off(nt + 1, hit_lo);
        for (uint32_t t = 0; t < nt; ++t) hit_off[t + 1] = hit_off[t] + w.sizes[t];
        const uint64_t nh = hit_off[nt] - hit_lo;
        const uint32_t ntiles = uint32_t((nh + kTile - 1) / kTile);
        std::vector<Cand> cand(size_t(ntiles) * kTile, Cand{kNone, kNone});
        std::vector<uint32_t> ncand(ntiles, 0), trange(2 * size_t(ntiles), 0);
        std::map<std::pair<uint32_t, uint32_t>, uint32_t> first;                 // (topic, client) -> smallest position
        std::vector<uint32_t> want;
        std::vector<std::pair<uint32_t, uint32_t>> all;                          // (pos, client) of every candidate
        for (uint32_t t = 0; t < nt; ++t)
            for (uint64_t p = hit_off[t] - hit_lo; p < hit_off[t + 1] - hit_lo; ++p)
                if ((rng() % 1000) < uint64_t(w.share * 1000)) {
                    const uint32_t c = uint32_t(rng() % w.clients);
                    all.push_back({uint32_t(p), c});
                    auto it = first.find({t, c});
                    if (it == first.end()) first[{t, c}] = uint32_t(p); else want.push_back(uint32_t(p));     // (positions ascend)
                }
        std::shuffle(all.begin(), all.end(), rng);                               // the expansion's lists are in no particular order
        for (auto& pc : all) { const uint32_t tl = pc.first / kTile; cand[size_t(tl) * kTile + ncand[tl]++] = Cand{pc.first, pc.second}; }
        auto topic_at = [&](uint64_t pos) { return uint32_t(std::upper_bound(hit_off.begin(), hit_off.end(), hit_lo + pos) - hit_off.begin() - 1); };
        for (uint32_t tl = 0; tl < ntiles; ++tl) {
            const uint64_t lo = uint64_t(tl) * kTile, hi = std::min<uint64_t>(nh, lo + kTile);
            const uint32_t tf = topic_at(lo), tlast = topic_at(hi - 1);
            const bool fw = hit_off[tf] - hit_lo >= lo, lw = hit_off[tlast + 1] - hit_lo <= lo + kTile;
            const bool whole = tf == tlast ? (fw && lw) : tlast - tf == 1 ? (fw || lw) : true;
            if (ncand[tl] >= 2 && whole) { ncand[tl] |= 1u << 31; trange[2 * tl] = tf; trange[2 * tl + 1] = tlast; }
        }
        std::sort(want.begin(), want.end());
        for (int variant : {3, 7}) {
            std::vector<Tuple> tuples(nh, Tuple{0, 0, 0});
            uint32_t n_items = 0;
            const int rc = sim_dedup(variant, w.grid, w.slots, cand.data(), ncand.data(), trange.data(), ntiles, tuples.data(), nt, hit_off.data(), hit_lo, &n_items);
            std::vector<uint32_t> got;
            for (uint64_t p = 0; p < nh; ++p) if (tuples[p].qos_flags & 16u) got.push_back(uint32_t(p));
            std::printf("world %zu, topic pass %d: rc %d, %u items, %zu duplicates flagged, %zu expected, %s\n", wi, variant, rc, n_items, got.size(), want.size(),
                        got == want ? "equal" : "DIFFERENT");
            bad += rc != 0 || got != want;
        }
    }
    return bad ? 1 : 0;
}
