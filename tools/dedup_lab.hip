// Stand-alone A/B for the v5 per-client dedup of the delivery stage — NOT part of the product library.
//
// The problem (types.rs:524-539): among the v5 hits ("candidates") of ONE publish topic, the first hit of every client
// — lowest position in the topic's hit list — creates the client's entry; later ones are flagged RGR_HIT_V5_DUP.
// A window holds up to 2^28 hit positions; the expansion leaves, per tile of 2048 positions, the list of that tile's
// candidates (Cand{pos, client_idx, topic}) and a per-topic candidate count.
//
//   G  global table     the product's kernels (kernels.hip dedup_insert / dedup_flag): one open-addressed region per
//                       topic in HBM, atomicCAS + atomicMin per candidate (device-scope atomics through the fabric),
//                       then a second pass that flags
//   L  LDS tables       duplicates only exist among the hits of ONE topic and a topic's hits are consecutive positions:
//        L.A  block per tile: every topic that lies entirely inside the tile is resolved in a 32 KB LDS table
//        L.B  block per topic that spans tiles and has at most kBigCap candidates: its tiles' lists are walked into a
//             128 KB LDS table (first and last tile are shared with neighbours: filtered by topic)
//        L.G  topics with more candidates than that keep the global path (their own regions only)
// Both produce one flag byte per candidate position; they must agree with the CPU's first-occurrence rule.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rmqtt_amd/csrc -I include tools/dedup_lab.hip -o tools/dedup_lab
//   tools/dedup_lab [log2_positions=26] [mean_hits_per_topic=14800] [v5_fraction=0.1] [n_clients=2500000] [reps=5] [sigma=0.7]
//     hits per topic are log-normal; sigma 0.7 reproduces config 3 as the oracle measures it on the first 20 000 publish topics
//     (mean 14 786, median 11 688, max 34 029 hits per topic: every topic meets the same few hot wildcard filters)
//   tools/dedup_lab host [...]        CPU only: generator, reference and the A / B / G partition of the topics (no GPU)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "kernels.hpp"
#include "match_core.hpp"

using namespace rgr;

#define CHECK(x)                                                                                         \
    do {                                                                                                 \
        hipError_t e_ = (x);                                                                             \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); std::exit(2); } \
    } while (0)

constexpr uint32_t kT = 2048;               // positions per expansion tile (expand_tile_hits())
constexpr uint32_t kSmallSlots = 4096;      // L.A: 2 slots per possible candidate of a tile, 8 B each = 32 KB
constexpr uint32_t kBigCap = 8192;          // L.B: candidates per topic that fit ...
constexpr uint32_t kBigSlots = 16384;       //      ... a 128 KB table
constexpr unsigned long long kEmpty = ~0ull;

// ---- G: the product's kernels, verbatim in structure (flags go to a byte array instead of tuples[].qos_flags)
__global__ __launch_bounds__(256) void g_insert(const Cand* __restrict__ cand, const uint32_t* __restrict__ tile_ncand, const uint64_t* __restrict__ cand_off,
                                                unsigned long long* table, const uint8_t* __restrict__ topic_sel) {
    const uint32_t n = tile_ncand[blockIdx.x];
    const Cand* list = cand + uint64_t(blockIdx.x) * kT;
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const Cand c = list[i];
        if (topic_sel && !topic_sel[c.topic]) continue;
        const uint64_t b = 2 * cand_off[c.topic], len = 2 * (cand_off[c.topic + 1] - cand_off[c.topic]);
        const unsigned long long mine = (static_cast<unsigned long long>(c.client_idx) << 32) | c.pos;
        for (uint64_t s = dedup_slot(c.client_idx, len);; s = (s + 1 == len) ? 0 : s + 1) {
            const unsigned long long prev = atomicCAS(&table[b + s], kEmpty, mine);
            if (prev == kEmpty) break;
            if (uint32_t(prev >> 32) == c.client_idx) { atomicMin(&table[b + s], mine); break; }
        }
    }
}
__global__ __launch_bounds__(256) void g_flag(const Cand* __restrict__ cand, const uint32_t* __restrict__ tile_ncand, const uint64_t* __restrict__ cand_off,
                                              const unsigned long long* __restrict__ table, uint8_t* __restrict__ dup, const uint8_t* __restrict__ topic_sel) {
    const uint32_t n = tile_ncand[blockIdx.x];
    const Cand* list = cand + uint64_t(blockIdx.x) * kT;
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const Cand c = list[i];
        if (topic_sel && !topic_sel[c.topic]) continue;
        const uint64_t b = 2 * cand_off[c.topic], len = 2 * (cand_off[c.topic + 1] - cand_off[c.topic]);
        for (uint64_t s = dedup_slot(c.client_idx, len);; s = (s + 1 == len) ? 0 : s + 1) {
            const unsigned long long e = table[b + s];
            if (uint32_t(e >> 32) == c.client_idx && e != kEmpty) { if (uint32_t(e) != c.pos) dup[c.pos] = 1; break; }
            if (e == kEmpty) break;
        }
    }
}

// ---- L: LDS tables.  An entry is client_idx << 32 | position-in-window, as in G: equal clients compare by position, so
// atomicMin keeps the first hit.  The topic is part of the SLOT, not of the key: in L.A every topic that lies inside the tile
// owns a private region of the tile's LDS table, [2 * (candidates of the tile's earlier inside-topics), + 2 * its own), found
// through a per-tile prefix over the topics that start in the tile (at most kT) — the product's global layout scaled down
// to one tile; in L.B the whole table belongs to the block's topic.
__device__ __forceinline__ uint32_t lds_slot(uint32_t client, uint32_t len) { return uint32_t(dedup_slot(client, len)); }

// A topic t lies inside tile `tile` iff [hit_off[t], hit_off[t+1]) is inside [tile*kT, (tile+1)*kT).
__global__ __launch_bounds__(256) void l_tile(const Cand* __restrict__ cand, const uint32_t* __restrict__ tile_ncand, const uint64_t* __restrict__ hit_off,
                                              const uint32_t* __restrict__ tile_topic0, uint8_t* __restrict__ dup) {
    __shared__ unsigned long long s_tab[kSmallSlots];
    __shared__ uint32_t s_cnt[kT + 1];                 // candidates per inside-topic (index = topic - t0), then their exclusive prefix
    const uint32_t tile = blockIdx.x;
    const uint32_t n = tile_ncand[tile];
    if (n < 2) return;                                  // nothing can be a duplicate
    const Cand* list = cand + uint64_t(tile) * kT;
    const uint64_t lo = uint64_t(tile) * kT, hi = lo + kT;
    const uint32_t t0 = tile_topic0[tile], t1 = tile_topic0[tile + 1];     // topics that START in this tile: [t0, t1)
    const uint32_t nt = t1 - t0;
    for (uint32_t i = threadIdx.x; i <= nt; i += 256) s_cnt[i] = 0;
    for (uint32_t i = threadIdx.x; i < kSmallSlots; i += 256) s_tab[i] = kEmpty;
    __syncthreads();
    // a topic that starts here is inside iff it also ends here
    auto inside = [&](uint32_t t) { return t >= t0 && t < t1 && hit_off[t + 1] <= hi; };
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const Cand c = list[i];
        if (inside(c.topic)) atomicAdd(&s_cnt[c.topic - t0], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {                             // exclusive prefix over at most kT small counters (a wave scan in the product)
        uint32_t run = 0;
        for (uint32_t i = 0; i <= nt; ++i) { const uint32_t v = s_cnt[i]; s_cnt[i] = run; run += v; }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const Cand c = list[i];
        if (!inside(c.topic)) continue;
        const uint32_t b = 2 * s_cnt[c.topic - t0], len = 2 * (s_cnt[c.topic - t0 + 1] - s_cnt[c.topic - t0]);
        if (len < 4) continue;                          // a single candidate has no duplicate
        const unsigned long long mine = (static_cast<unsigned long long>(c.client_idx) << 32) | c.pos;
        for (uint32_t s = lds_slot(c.client_idx, len);; s = (s + 1 == len) ? 0 : s + 1) {
            const unsigned long long prev = atomicCAS(&s_tab[b + s], kEmpty, mine);
            if (prev == kEmpty) break;
            if (uint32_t(prev >> 32) == c.client_idx) { atomicMin(&s_tab[b + s], mine); break; }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const Cand c = list[i];
        if (!inside(c.topic)) continue;
        const uint32_t b = 2 * s_cnt[c.topic - t0], len = 2 * (s_cnt[c.topic - t0 + 1] - s_cnt[c.topic - t0]);
        if (len < 4) continue;
        for (uint32_t s = lds_slot(c.client_idx, len);; s = (s + 1 == len) ? 0 : s + 1) {
            const unsigned long long e = s_tab[b + s];
            if (uint32_t(e >> 32) == c.client_idx && e != kEmpty) { if (uint32_t(e) != c.pos) dup[c.pos] = 1; break; }
            if (e == kEmpty) break;
        }
    }
    (void)lo;
}

// One block per spanning topic with 2..kBigCap candidates: span_topic[blockIdx.x].
__global__ __launch_bounds__(512) void l_topic(const Cand* __restrict__ cand, const uint32_t* __restrict__ tile_ncand, const uint64_t* __restrict__ hit_off,
                                               const uint32_t* __restrict__ span_topic, uint8_t* __restrict__ dup) {
    extern __shared__ unsigned long long s_big[];      // kBigSlots entries
    const uint32_t t = span_topic[blockIdx.x];
    const uint64_t h0 = hit_off[t], h1 = hit_off[t + 1];
    const uint32_t tile0 = uint32_t(h0 / kT), tile1 = uint32_t((h1 - 1) / kT);
    for (uint32_t i = threadIdx.x; i < kBigSlots; i += 512) s_big[i] = kEmpty;
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
        for (uint32_t tile = tile0; tile <= tile1; ++tile) {
            const uint32_t n = tile_ncand[tile];
            const Cand* list = cand + uint64_t(tile) * kT;
            for (uint32_t i = threadIdx.x; i < n; i += 512) {
                const Cand c = list[i];
                if (c.topic != t) continue;
                const unsigned long long mine = (static_cast<unsigned long long>(c.client_idx) << 32) | c.pos;
                for (uint32_t s = lds_slot(c.client_idx, kBigSlots);; s = (s + 1) & (kBigSlots - 1)) {
                    if (pass == 0) {
                        const unsigned long long prev = atomicCAS(&s_big[s], kEmpty, mine);
                        if (prev == kEmpty) break;
                        if (uint32_t(prev >> 32) == c.client_idx) { atomicMin(&s_big[s], mine); break; }
                    } else {
                        const unsigned long long e = s_big[s];
                        if (uint32_t(e >> 32) == c.client_idx && e != kEmpty) { if (uint32_t(e) != c.pos) dup[c.pos] = 1; break; }
                        if (e == kEmpty) break;
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---- synthetic window
struct Window {
    uint64_t n_pos = 0;
    std::vector<uint64_t> hit_off;        // [n_topics + 1]
    std::vector<Cand> cand;               // tile i owns cand[i * kT ...], tile_ncand[i] used
    std::vector<uint32_t> tile_ncand;
    std::vector<uint64_t> cand_off;       // per-topic candidate prefix (the product's scan)
    std::vector<uint32_t> tile_topic0;    // first topic that STARTS in tile i (n_tiles + 1 entries)
    std::vector<uint8_t> ref;             // expected flag per position
    uint64_t n_cand = 0, n_dup = 0;
};

static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static inline uint64_t rnd() { g_rng ^= g_rng << 7; g_rng ^= g_rng >> 9; return g_rng * 0x2545F4914F6CDD1Dull; }
static inline double urand() { return double(rnd() >> 11) * (1.0 / 9007199254740992.0); }

static Window make_window(uint64_t n_pos, double mean_hits, double v5, uint64_t n_clients, double sigma) {
    Window w;
    w.n_pos = n_pos;
    // per-topic hit counts: log-normal around mean_hits
    const double mu = std::log(mean_hits) - sigma * sigma / 2;
    w.hit_off.push_back(0);
    while (w.hit_off.back() < n_pos) {
        const double z = std::sqrt(-2.0 * std::log(1.0 - urand())) * std::cos(6.283185307179586 * urand());
        uint64_t h = uint64_t(std::max(1.0, std::exp(mu + sigma * z)));
        h = std::min<uint64_t>(h, n_pos - w.hit_off.back());
        w.hit_off.push_back(w.hit_off.back() + h);
    }
    const uint32_t n_topics = uint32_t(w.hit_off.size() - 1), n_tiles = uint32_t((n_pos + kT - 1) / kT);
    w.cand.resize(uint64_t(n_tiles) * kT);
    w.tile_ncand.assign(n_tiles, 0);
    w.cand_off.assign(size_t(n_topics) + 1, 0);
    w.ref.assign(n_pos, 0);
    // clients: Zipf(1.0) over n_clients by inversion of the continuous approximation
    const double lnN = std::log(double(n_clients));
    std::unordered_map<uint32_t, uint32_t> first;      // client -> first position, per topic
    for (uint32_t t = 0; t < n_topics; ++t) {
        first.clear();
        for (uint64_t p = w.hit_off[t]; p < w.hit_off[t + 1]; ++p) {
            if (urand() >= v5) continue;
            const uint32_t client = uint32_t(std::min<double>(double(n_clients - 1), std::exp(urand() * lnN) - 1.0));
            const uint32_t tile = uint32_t(p / kT);
            w.cand[uint64_t(tile) * kT + w.tile_ncand[tile]++] = Cand{uint32_t(p), client, t, 0u};
            w.cand_off[t + 1]++;
            w.n_cand++;
            if (!first.emplace(client, uint32_t(p)).second) { w.ref[p] = 1; w.n_dup++; }
        }
    }
    for (uint32_t t = 0; t < n_topics; ++t) w.cand_off[t + 1] += w.cand_off[t];
    // the expansion fills a tile's list through an LDS counter, in no particular order: shuffle every list
    for (uint32_t tile = 0; tile < n_tiles; ++tile) {
        Cand* l = w.cand.data() + uint64_t(tile) * kT;
        for (uint32_t i = w.tile_ncand[tile]; i > 1; --i) std::swap(l[i - 1], l[rnd() % i]);
    }
    w.tile_topic0.assign(size_t(n_tiles) + 1, n_topics);
    for (uint32_t t = n_topics; t-- > 0;) w.tile_topic0[w.hit_off[t] / kT] = t;
    for (uint32_t tile = n_tiles; tile-- > 0;) if (w.tile_topic0[tile] == n_topics || w.tile_topic0[tile] > w.tile_topic0[tile + 1]) w.tile_topic0[tile] = w.tile_topic0[tile + 1];
    return w;
}

// which path a topic takes under L: 0 = inside one tile (L.A), 1 = spans tiles, LDS (L.B), 2 = spans tiles, global (L.G), 3 = fewer than 2 candidates
static int topic_class(const Window& w, uint32_t t) {
    const uint64_t nc = w.cand_off[t + 1] - w.cand_off[t];
    if (nc < 2) return 3;
    if (w.hit_off[t] / kT == (w.hit_off[t + 1] - 1) / kT) return 0;
    return nc <= kBigCap ? 1 : 2;
}

int main(int argc, char** argv) {
    const bool host_only = argc > 1 && std::strcmp(argv[1], "host") == 0;
    if (host_only) { --argc; ++argv; }
    const int lg = argc > 1 ? std::atoi(argv[1]) : 26;
    const double mean_hits = argc > 2 ? std::atof(argv[2]) : 14800.0;
    const double v5 = argc > 3 ? std::atof(argv[3]) : 0.1;
    const uint64_t n_clients = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : 2500000;
    const int reps = argc > 5 ? std::atoi(argv[5]) : 5;
    const double sigma = argc > 6 ? std::atof(argv[6]) : 0.7;
    const Window w = make_window(1ull << lg, mean_hits, v5, n_clients, sigma);
    const uint32_t n_topics = uint32_t(w.hit_off.size() - 1), n_tiles = uint32_t(w.tile_ncand.size());
    uint64_t cls_topics[4] = {0, 0, 0, 0}, cls_cands[4] = {0, 0, 0, 0};
    std::vector<uint32_t> span;           // L.B topics
    std::vector<uint8_t> sel_global(n_topics, 0);
    for (uint32_t t = 0; t < n_topics; ++t) {
        const int c = topic_class(w, t);
        cls_topics[c]++; cls_cands[c] += w.cand_off[t + 1] - w.cand_off[t];
        if (c == 1) span.push_back(t);
        if (c == 2) sel_global[t] = 1;
    }
    std::printf("window: 2^%d positions, %u topics (mean %.0f hits), %u tiles, %llu v5 candidates (%.1f %%), %llu duplicates\n", lg, n_topics,
                double(w.n_pos) / n_topics, n_tiles, (unsigned long long)w.n_cand, 100.0 * w.n_cand / w.n_pos, (unsigned long long)w.n_dup);
    std::printf("L partition: inside one tile %llu topics / %llu candidates; spanning, LDS %llu / %llu; spanning, global %llu / %llu; <2 candidates %llu / %llu\n",
                (unsigned long long)cls_topics[0], (unsigned long long)cls_cands[0], (unsigned long long)cls_topics[1], (unsigned long long)cls_cands[1],
                (unsigned long long)cls_topics[2], (unsigned long long)cls_cands[2], (unsigned long long)cls_topics[3], (unsigned long long)cls_cands[3]);
    if (host_only) {
        // the partition covers every topic exactly once and tile_topic0 is what l_tile assumes
        uint64_t bad = 0;
        for (uint32_t tile = 0; tile < n_tiles; ++tile) {
            const uint32_t t0 = w.tile_topic0[tile], t1 = w.tile_topic0[tile + 1];
            for (uint32_t t = t0; t < t1; ++t) bad += w.hit_off[t] / kT != tile;
            if (t0 > 0 && t0 <= n_topics && tile > 0) bad += w.hit_off[t0 - 1] / kT == tile;      // a topic starting here but left out of [t0, t1)
            uint32_t inside_cands = 0;
            for (uint32_t i = 0; i < w.tile_ncand[tile]; ++i) {
                const Cand& c = w.cand[uint64_t(tile) * kT + i];
                bad += c.pos / kT != tile || c.pos < w.hit_off[c.topic] || c.pos >= w.hit_off[c.topic + 1];
                inside_cands += topic_class(w, c.topic) == 0;
            }
            bad += 2 * inside_cands > kSmallSlots;
        }
        std::printf("host check: %llu inconsistencies\n", (unsigned long long)bad);
        return bad != 0;
    }

    Cand* d_cand; uint32_t *d_ncand, *d_topic0, *d_span; uint64_t *d_hit_off, *d_cand_off; unsigned long long* d_tab; uint8_t *d_dup, *d_sel;
    CHECK(hipMalloc(&d_cand, w.cand.size() * sizeof(Cand))); CHECK(hipMemcpy(d_cand, w.cand.data(), w.cand.size() * sizeof(Cand), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_ncand, n_tiles * 4)); CHECK(hipMemcpy(d_ncand, w.tile_ncand.data(), n_tiles * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_topic0, (n_tiles + 1) * 4)); CHECK(hipMemcpy(d_topic0, w.tile_topic0.data(), (n_tiles + 1) * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_span, std::max<size_t>(1, span.size()) * 4)); CHECK(hipMemcpy(d_span, span.data(), span.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_hit_off, (n_topics + 1) * 8)); CHECK(hipMemcpy(d_hit_off, w.hit_off.data(), (n_topics + 1) * 8, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_cand_off, (n_topics + 1) * 8)); CHECK(hipMemcpy(d_cand_off, w.cand_off.data(), (n_topics + 1) * 8, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_tab, std::max<uint64_t>(1, 2 * w.n_cand) * 8));
    CHECK(hipMalloc(&d_dup, w.n_pos));
    CHECK(hipMalloc(&d_sel, n_topics)); CHECK(hipMemcpy(d_sel, sel_global.data(), n_topics, hipMemcpyHostToDevice));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(l_topic), hipFuncAttributeMaxDynamicSharedMemorySize, kBigSlots * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto timeit = [&](auto launch) {
        launch(); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        return ms / reps;
    };
    auto check = [&](const char* name) {
        std::vector<uint8_t> got(w.n_pos);
        CHECK(hipMemcpy(got.data(), d_dup, w.n_pos, hipMemcpyDeviceToHost));
        uint64_t diff = 0;
        for (uint64_t p = 0; p < w.n_pos; ++p) diff += got[p] != w.ref[p];
        std::printf("%s: %llu positions differ from the CPU's first-occurrence rule\n", name, (unsigned long long)diff);
        return diff;
    };
    auto run_g = [&] {      // the table memset is part of the product's pass
        CHECK(hipMemsetAsync(d_tab, 0xFF, std::max<uint64_t>(1, 2 * w.n_cand) * 8));
        g_insert<<<n_tiles, 256>>>(d_cand, d_ncand, d_cand_off, d_tab, nullptr);
        g_flag<<<n_tiles, 256>>>(d_cand, d_ncand, d_cand_off, d_tab, d_dup, nullptr);
    };
    auto run_l = [&] {
        l_tile<<<n_tiles, 256>>>(d_cand, d_ncand, d_hit_off, d_topic0, d_dup);
        if (!span.empty()) l_topic<<<uint32_t(span.size()), 512, kBigSlots * 8>>>(d_cand, d_ncand, d_hit_off, d_span, d_dup);
        if (cls_topics[2]) {
            CHECK(hipMemsetAsync(d_tab, 0xFF, std::max<uint64_t>(1, 2 * w.n_cand) * 8));      // (the product would size it by the big topics only)
            g_insert<<<n_tiles, 256>>>(d_cand, d_ncand, d_cand_off, d_tab, d_sel);
            g_flag<<<n_tiles, 256>>>(d_cand, d_ncand, d_cand_off, d_tab, d_dup, d_sel);
        }
    };
    CHECK(hipMemset(d_dup, 0, w.n_pos));
    const float tg = timeit(run_g);
    uint64_t bad = check("G global table");
    CHECK(hipMemset(d_dup, 0, w.n_pos));
    const float tl = timeit(run_l);
    bad += check("L LDS tables  ");
    std::printf("G global table (atomicCAS + atomicMin in HBM, 2 passes)      %9.3f ms  %8.2f G candidates/s\n", tg, w.n_cand / tg / 1e6);
    std::printf("L LDS tables   (tile-local + topic-local, big topics global) %9.3f ms  %8.2f G candidates/s   (%.2fx G)\n", tl, w.n_cand / tl / 1e6, tl / tg);
    return bad != 0;
}
