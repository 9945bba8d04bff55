// Host-only: distribution of the walk's work per publish topic over the product's own table (HostTable + walk_topic):
// dependent record reads, matched filters, and how much of the work sits in the topics that overflow the slot capacity.
// The walk kernels' duration has a floor = the longest dependent-read chain of ONE topic; this tool sizes that tail.
//
//   g++ -O2 -std=c++17 -I rmqtt_amd/csrc -I include tools/walk_tail.cpp rmqtt_amd/csrc/table.cpp rmqtt_amd/csrc/workload.cpp -o tools/walk_tail -pthread
//   tools/walk_tail [n_sub=10000000] [n_pub=2000000] [p_plus=0.028] [p_hash=0.1] [slot_cap=32]
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string_view>
#include <vector>

#include "kernels.hpp"
#include "match_core.hpp"
#include "table.hpp"

using namespace rgr;

struct wl_params { uint64_t seed, n; double p_plus, p_hash, p_sys, p_blank; uint64_t n_clients; int32_t fixed_depth, force_wildcard, distinct, reserved; };
extern "C" int wl_gen_subs(const wl_params* p, char** blob, uint64_t** offsets, uint32_t** client, uint8_t** qos);
extern "C" int wl_gen_topics(const wl_params* p, char** blob, uint64_t** offsets);

int main(int argc, char** argv) {
    const uint64_t n_sub = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 10000000;
    const uint64_t n_pub = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 2000000;
    const double p_plus = argc > 3 ? std::atof(argv[3]) : 0.028;
    const double p_hash = argc > 4 ? std::atof(argv[4]) : 0.1;
    const uint32_t cap = argc > 5 ? uint32_t(std::atoi(argv[5])) : 32;
    wl_params ps{0x5EED0003, n_sub, p_plus, p_hash, 0.005, 0.0, 0, 0, 0, 0, 0};
    char* fb; uint64_t* fo; uint32_t* cl; uint8_t* q;
    wl_gen_subs(&ps, &fb, &fo, &cl, &q);
    wl_params pt{0x9B1C0003, n_pub, 0, 0, 0.01, 0.01, 0, 0, 0, 0, 0};
    char* tb; uint64_t* to;
    wl_gen_topics(&pt, &tb, &to);
    HostTable table;
    uint64_t rej = 0;
    table.subscribe_bulk(reinterpret_cast<const uint8_t*>(fb), fo, n_sub, nullptr, q, nullptr, nullptr, &rej, 0);
    const auto& edges = table.edges();
    const uint32_t mask = uint32_t(edges.size() - 1);
    std::vector<uint32_t> reads(n_pub), cnts(n_pub), toks;
    uint64_t tot_reads = 0, tot_cnt = 0, ovf_topics = 0, ovf_reads = 0, ovf_cnt = 0;
    std::vector<uint32_t> path(70000);
    for (uint64_t t = 0; t < n_pub; ++t) {
        toks.clear();
        const uint8_t fl = table.tokenize_topic(std::string_view(tb + to[t], to[t + 1] - to[t]), toks);
        uint32_t c = 0, r = 0;
        if (!(fl & kTopicInvalid))
            walk_topic(
                table.root_header(), mask, uint32_t(toks.size()), (fl & kTopicMeta) != 0, [&](uint32_t d) { return toks[d]; }, [&](uint32_t d) { return path[d]; },
                [&](uint32_t d, uint32_t v) { path[d] = v; }, [&](uint32_t) { c++; },
                [&](uint32_t slot, U4& e0, U4& e1) { const U4* h = reinterpret_cast<const U4*>(&edges[slot]); e0 = h[0]; e1 = h[1]; r++; });
        reads[t] = r; cnts[t] = c;
        tot_reads += r; tot_cnt += c;
        if (c > cap) { ovf_topics++; ovf_reads += r; ovf_cnt += c; }
    }
    auto pct = [&](std::vector<uint32_t> v, const char* what) {
        std::sort(v.begin(), v.end());
        auto at = [&](double p) { return v[std::min<size_t>(v.size() - 1, size_t(p * v.size()))]; };
        std::printf("%-28s mean %.2f  p50 %u  p90 %u  p99 %u  p99.9 %u  p99.99 %u  max %u\n", what, double(what[0] == 'r' ? tot_reads : tot_cnt) / n_pub, at(0.5), at(0.9), at(0.99),
                    at(0.999), at(0.9999), v.back());
    };
    std::printf("table: %llu subs (p_plus %.3f, p_hash %.2f), %llu trie nodes; %llu topics, slot capacity %u\n", (unsigned long long)n_sub, p_plus, p_hash,
                (unsigned long long)table.n_nodes(), (unsigned long long)n_pub, cap);
    pct(reads, "record reads per topic:");
    pct(cnts, "matched filters per topic:");
    std::printf("topics over the slot capacity: %llu (%.2f %%), their reads %.1f %% of all reads (%.1f per topic), their matched filters %.1f %% of all (%.1f per topic)\n",
                (unsigned long long)ovf_topics, 100.0 * ovf_topics / n_pub, 100.0 * ovf_reads / tot_reads, ovf_topics ? double(ovf_reads) / ovf_topics : 0.0,
                100.0 * ovf_cnt / tot_cnt, ovf_topics ? double(ovf_cnt) / ovf_topics : 0.0);
    for (uint32_t c2 : {48u, 64u, 96u, 128u, 256u}) {
        uint64_t k = 0;
        for (uint64_t t = 0; t < n_pub; ++t) k += cnts[t] > c2;
        std::printf("  topics with more than %3u matched filters: %llu (%.3f %%)\n", c2, (unsigned long long)k, 100.0 * k / n_pub);
    }
    return 0;
}
