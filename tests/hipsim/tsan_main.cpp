// TEST INFRASTRUCTURE ONLY — the hipsim expansion under ThreadSanitizer and AddressSanitizer (tools/hipsim_sanitizers.sh).  TSAN: a
// missing __syncthreads() around the LDS-staged pair views, or a lane mirror overwritten while a slower lane still reads it, is a data
// race between the OS threads that stand for GPU threads.  ASAN: every array has its exact product size (the packed side array with
// its kPackedPad entries, the output without any slack, the last run ending where the pool ends), so a load or store the device would
// do outside an allocation is reported here.  Also compares every output with a scalar expansion.
#include "sim_expand_compact.cpp"

#include <cstdio>
#include <random>

int main() {
    std::mt19937_64 rng(20260921);
    const size_t pool = 1 << 16;
    std::vector<SubEntry> subs(pool);
    std::vector<uint32_t> packed(pool + kPackedPad, 0);
    for (size_t i = 0; i < pool; ++i) { subs[i] = SubEntry{uint32_t(rng() & 0xFFFFFF), uint32_t(rng() % 3) | uint32_t((rng() & 63) << 8)}; packed[i] = subs[i].sub_id | ((subs[i].qos_flags & 3u) << 30); }
    // run lengths: long runs, bursts of singletons (more than 64 pairs in a tile), short runs (straddling groups)
    std::vector<uint32_t> len;
    auto burst = [&](int n, uint32_t lo, uint32_t hi) { for (int i = 0; i < n; ++i) len.push_back(lo + uint32_t(rng() % (hi - lo + 1))); };
    burst(2, 3000, 9000); burst(3000, 1, 6); burst(1, 2048, 2048); burst(70, 1, 1); burst(3, 500, 2500); burst(200, 20, 120); burst(1, 7000, 7000); burst(9, 1, 3);
    burst(12, 300, 2500); burst(5000, 1, 2); burst(2, 9000, 9000);
    const uint64_t first = (1ull << 32) + 77;
    std::vector<uint32_t> src(len.size()), topic(len.size(), 0);
    std::vector<uint64_t> off(len.size() + 1);
    off[0] = first;
    for (size_t p = 0; p < len.size(); ++p) { src[p] = uint32_t(rng() % (pool - len[p])); off[p + 1] = off[p] + len[p]; }
    src.back() = uint32_t(pool - len.back());            // the last run ends where the pool ends: a read past a partial last tile's run leaves the allocation (ASAN)
    const uint64_t hits = off.back() - first;
    std::vector<uint32_t> want(hits);
    { uint64_t k = 0; for (size_t p = 0; p < len.size(); ++p) for (uint32_t j = 0; j < len[p]; ++j) want[k++] = packed[src[p] + j]; }
    int bad = 0;
    struct K { int variant, fmt, tiles; bool pk; } ks[] = {{0, 1, 1, false}, {0, 2, 1, true}, {0, 4, 4, true}, {0, 2, 4, true}, {1, 2, 1, true}, {1, 2, 2, true},
                                                          {1, 2, 4, true}, {1, 4, 1, true}, {1, 4, 2, true}, {1, 4, 4, true}, {2, 4, 2, true}};
    for (const K& k : ks) {
        std::vector<uint8_t> out(hits * (k.fmt == 4 ? 3 : 4), 0), qos(hits, 0);      // exact sizes: a store past the window is an ASAN report
        const int rc = sim_expand_compact(k.variant, k.fmt, k.tiles, subs.data(), k.pk ? packed.data() : nullptr, src.data(), topic.data(), off.data(), 0,
                                          len.size(), reinterpret_cast<uint32_t*>(out.data()), qos.data());
        uint64_t diff = 0;
        for (uint64_t i = 0; i < hits; ++i) {
            uint32_t got;
            if (k.fmt == 4) got = uint32_t(out[3 * i]) | uint32_t(out[3 * i + 1]) << 8 | uint32_t(out[3 * i + 2]) << 16;
            else { std::memcpy(&got, &out[4 * i], 4); }
            const uint32_t w = k.fmt == 2 ? want[i] : (want[i] & 0xFFFFFFu) & (k.fmt == 1 ? 0xFFFFFFFFu : 0xFFFFFFu);
            diff += got != (k.fmt == 1 ? (want[i] & 0x3FFFFFFFu) : w);
        }
        std::printf("variant %d fmt %d tiles %d packed %d: rc %d, %llu of %llu hits differ\n", k.variant, k.fmt, k.tiles, int(k.pk), rc, (unsigned long long)diff,
                    (unsigned long long)hits);
        bad += rc != 0 || diff != 0;
    }
    return bad ? 1 : 0;
}
