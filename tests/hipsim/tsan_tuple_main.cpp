// TEST INFRASTRUCTURE ONLY — the tuple / delivery expansions of rmqtt_amd/csrc/expand_tuple.inc on the host under ThreadSanitizer and
// AddressSanitizer (tools/hipsim_sanitizers.sh).  TSAN: the lean delivery expansion (r5) keeps its v5 hits in wave-private LDS lists
// without block barriers and lets the LAST wave write the tile's count word — an unordered access to the lists, to s_ncand / s_done or to
// the parked words of the whole-topic test is a data race between the OS threads that stand for GPU threads.  ASAN: exact array sizes
// (output without slack, candidate slices of exactly kTile entries per tile, the last run ending where the pool ends).  Also compares the
// lean kernels with expand_kernel<true>: same words, same candidate sets, same count words.
#include "sim_expand_tuple.cpp"

#include <algorithm>
#include <cstdio>
#include <random>

int main() {
    std::mt19937_64 rng(20260922);
    const size_t pool = 1 << 15;
    std::vector<SubEntry> subs(pool);
    std::vector<SubAttr> attrs(pool);
    for (size_t i = 0; i < pool; ++i) {
        const uint32_t v5 = (rng() % 100) < 30, fl = (v5 ? 1u : 0u) | uint32_t((rng() & 7) << 1);
        subs[i] = SubEntry{uint32_t(rng() & 0x3FFFFFFF), uint32_t(rng() % 3) | (fl << 8) | uint32_t((rng() % 40) << 16)};
        const uint32_t owner = uint32_t(rng() % 300);
        attrs[i] = SubAttr{owner, (rng() % 20) == 0 ? kNone : owner};
    }
    // topics: long runs (single-run tiles), bursts of short runs (multi-run tiles, > 64 pairs per tile), a run of exactly one tile, small topics
    std::vector<std::vector<uint32_t>> topics = {{9000, 3, 700}, {2048}, {5, 2043}, {1, 1, 1, 2, 3}, {}, {6000}};
    { std::vector<uint32_t> t; for (int i = 0; i < 400; ++i) t.push_back(1 + uint32_t(rng() % 9)); topics.push_back(t); }
    { std::vector<uint32_t> t; for (int i = 0; i < 30; ++i) t.push_back(20 + uint32_t(rng() % 200)); topics.push_back(t); }
    topics.push_back({777});
    std::vector<uint32_t> src, ptopic;
    std::vector<uint64_t> off;
    std::vector<uint8_t> qr;
    const uint32_t topic_lo = 50;
    std::vector<PublishAttr> pub(topic_lo + topics.size());
    for (auto& p : pub) p = PublishAttr{(rng() % 3) == 0 ? kNone : uint32_t(rng() % 300), uint32_t(rng() % 3) | uint32_t((rng() & 1) << 2)};
    const uint64_t first = (1ull << 33) + 5;
    off.push_back(first);
    for (size_t t = 0; t < topics.size(); ++t)
        for (uint32_t n : topics[t]) { src.push_back(uint32_t(rng() % (pool - n))); ptopic.push_back(topic_lo + uint32_t(t)); qr.push_back(uint8_t(pub[topic_lo + t].qos_retain & 7)); off.push_back(off.back() + n); }
    src.back() = uint32_t(pool - (off.back() - off[off.size() - 2]));        // the last run ends where the pool ends
    const uint64_t np = src.size(), hits = off.back() - first;
    const uint32_t ntiles = uint32_t((hits + kTile - 1) / kTile);
    auto run = [&](int variant, std::vector<Tuple>& out, std::vector<Cand>& cand, std::vector<uint32_t>& ncand, std::vector<uint32_t>& trange) {
        out.assign(hits, Tuple{0, 0, 0}); cand.assign(size_t(ntiles) * kTile, Cand{kNone, kNone}); ncand.assign(ntiles, 0xDEADBEEFu); trange.assign(2 * size_t(ntiles), 0xDEADBEEFu);
        return sim_expand_tuple(variant, subs.data(), attrs.data(), pub.data(), src.data(), ptopic.data(), off.data(), qr.data(), 0, np, topic_lo, out.data(),
                                cand.data(), ncand.data(), trange.data());
    };
    std::vector<Tuple> o1, o; std::vector<Cand> c1, c; std::vector<uint32_t> n1, n, r1, r;
    int bad = run(1, o1, c1, n1, r1) != 0;
    for (int variant : {3, 4, 2}) {
        const int rc = run(variant, o, c, n, r);
        uint64_t dw = 0, dc = 0, dn = 0;
        for (uint64_t i = 0; i < hits; ++i) dw += o[i].topic_idx != o1[i].topic_idx || o[i].sub_id != o1[i].sub_id || o[i].qos_flags != o1[i].qos_flags;
        for (uint32_t t = 0; t < ntiles; ++t) {
            dn += n[t] != n1[t];
            const uint32_t k = n1[t] & 0x7FFFFFFFu;
            auto key = [](const Cand& x) { return (uint64_t(x.pos) << 32) | x.client_idx; };
            std::vector<uint64_t> a, b;
            for (uint32_t i = 0; i < k && i < uint32_t(kTile); ++i) { a.push_back(key(c1[size_t(t) * kTile + i])); b.push_back(key(c[size_t(t) * kTile + i])); }
            std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
            dc += a != b;
            if ((n1[t] >> 31) && (r[2 * t] != r1[2 * t] || r[2 * t + 1] != r1[2 * t + 1])) dn++;
        }
        std::printf("variant %d vs expand_kernel<true>: rc %d, %llu of %llu words differ, %llu of %u tiles' candidate sets differ, %llu count words / ranges differ\n", variant, rc,
                    (unsigned long long)dw, (unsigned long long)hits, (unsigned long long)dc, ntiles, (unsigned long long)dn);
        bad += rc != 0 || dw || dc || dn;
    }
    // (r6) the lean variants writing 8-byte hits {sub_id, delivery word} (RGR_FORMAT_DELIVER8) into a buffer of exactly hits * 8 bytes: the same words
    // and sub ids as the tuples, the same candidate sets and count words
    for (int variant : {5, 6}) {
        std::vector<unsigned long long> o8(hits, 0);
        c.assign(size_t(ntiles) * kTile, Cand{kNone, kNone}); n.assign(ntiles, 0xDEADBEEFu); r.assign(2 * size_t(ntiles), 0xDEADBEEFu);
        const int rc = sim_expand_tuple(variant, subs.data(), attrs.data(), pub.data(), src.data(), ptopic.data(), off.data(), qr.data(), 0, np, topic_lo,
                                        reinterpret_cast<Tuple*>(o8.data()), c.data(), n.data(), r.data());
        uint64_t dw = 0, dn = 0;
        for (uint64_t i = 0; i < hits; ++i) dw += uint32_t(o8[i]) != o1[i].sub_id || uint32_t(o8[i] >> 32) != o1[i].qos_flags;
        for (uint32_t t = 0; t < ntiles; ++t) dn += n[t] != n1[t];
        std::printf("variant %d (8-byte hits) vs expand_kernel<true>: rc %d, %llu of %llu hits differ, %llu count words differ\n", variant, rc, (unsigned long long)dw,
                    (unsigned long long)hits, (unsigned long long)dn);
        bad += rc != 0 || dw || dn;
    }
    return bad ? 1 : 0;
}
