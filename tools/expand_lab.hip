// Stand-alone lab for the expansion kernel — NOT part of the product library.
//
// Builds a synthetic (topic, subscriber-run) pair list with a heavy-tailed run-length distribution,
// expands it with
//   A  expand_ref    the product kernel's structure (one 2048-hit tile per block, staged pair view,
//                    loads-then-stores, nontemporal 12-byte tuple stores), and
//   B  expand_pipe   a persistent variant: each block walks tiles b, b+G, b+2G, ... and prefetches the
//                    NEXT tile's pair view into registers (and the tile-after-next's tile_first) while
//                    it expands the current one; LDS pair view double-buffered,
// checks that both produce identical tuples (and a host reference on a sample), and prints the
// HIP-event time and store bandwidth of each.  Uses the product's own index helpers
// (rmqtt_amd/csrc/match_core.hpp).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rmqtt_amd/csrc -I include tools/expand_lab.hip -o tools/expand_lab
//   tools/expand_lab [hits_log2=28] [mean_run=3000] [pool_entries=10000000] [reps=10]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "kernels.hpp"
#include "match_core.hpp"

using namespace rgr;

#define CHECK(x)                                                                                         \
    do {                                                                                                 \
        hipError_t e_ = (x);                                                                             \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); std::exit(2); } \
    } while (0)

constexpr int kThreads = 512;
constexpr int kPer = 4;
constexpr int kTileHits = kThreads * kPer;

__global__ __launch_bounds__(256) void tiles_k(const uint64_t* pair_off, uint64_t pair_lo, uint64_t pair_hi, uint64_t hit_lo, uint32_t* tile_first) {
    const uint64_t p = pair_lo + uint64_t(blockIdx.x) * 256 + threadIdx.x;
    if (p < pair_hi) tiles_pair(pair_off, p, pair_lo, hit_lo, kTileHits, tile_first);
}

__device__ __forceinline__ void store_tuple(Tuple* o, uint32_t topic, SubEntry se) {
    __builtin_nontemporal_store(topic, &o->topic_idx);
    __builtin_nontemporal_store(se.sub_id, &o->sub_id);
    __builtin_nontemporal_store(se.qos_flags, &o->qos_flags);
}

// The per-tile expansion shared by both variants: pair view of the tile in LDS -> tuples.
__device__ __forceinline__ void expand_tile(const SubEntry* __restrict__ subs, const int32_t* s_off, const uint32_t* s_src, const uint32_t* s_topic,
                                            uint32_t np, uint32_t len, Tuple* __restrict__ o) {
    uint32_t topic[kPer];
    const SubEntry* src[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const uint32_t pos = uint32_t(j) * kThreads + threadIdx.x;
        const bool live = pos < len;
        const uint32_t i = (np == 1 || !live) ? 0u : locate_pair([&](uint32_t m) { return s_off[m]; }, np, int32_t(pos));
        topic[j] = s_topic[i];
        src[j] = subs + (uint64_t(s_src[i]) + (live ? uint32_t(int32_t(pos) - s_off[i]) : 0u));
    }
    SubEntry se[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) se[j] = *src[j];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const uint32_t pos = uint32_t(j) * kThreads + threadIdx.x;
        if (pos < len) store_tuple(o + pos, topic[j], se[j]);
    }
}

// ---- A: one tile per block (the product kernel's shape)
__global__ __launch_bounds__(kThreads) void expand_ref(const SubEntry* __restrict__ subs, ChunkArrays c, uint64_t pair_lo, uint64_t pair_hi,
                                                       uint64_t hit_lo, uint64_t hit_hi, const uint32_t* __restrict__ tile_first, uint32_t ntiles,
                                                       Tuple* __restrict__ out) {
    __shared__ int32_t s_off[kTileHits + 2];
    __shared__ uint32_t s_src[kTileHits + 2];
    __shared__ uint32_t s_topic[kTileHits + 2];
    const uint32_t tile = blockIdx.x;
    const uint64_t base = hit_lo + uint64_t(tile) * kTileHits;
    const uint32_t len = (hit_hi - base) < uint64_t(kTileHits) ? uint32_t(hit_hi - base) : uint32_t(kTileHits);
    const uint64_t a = pair_lo + tile_first[tile];
    const uint64_t b = (tile + 1 < ntiles) ? pair_lo + tile_first[tile + 1] + 1 : pair_hi;
    const uint32_t np = uint32_t(b - a);
    for (uint32_t i = threadIdx.x; i < np; i += kThreads) tile_pair_view(c, a, i, base, s_off[i], s_src[i], s_topic[i]);
    __syncthreads();
    expand_tile(subs, s_off, s_src, s_topic, np, len, out + (base - hit_lo));
}

// ---- B: persistent blocks, next tile's pair view prefetched while the current tile is stored
__global__ __launch_bounds__(kThreads) void expand_pipe(const SubEntry* __restrict__ subs, ChunkArrays c, uint64_t pair_lo, uint64_t pair_hi,
                                                        uint64_t hit_lo, uint64_t hit_hi, const uint32_t* __restrict__ tile_first, uint32_t ntiles,
                                                        Tuple* __restrict__ out) {
    __shared__ int32_t s_off[2][kTileHits + 2];
    __shared__ uint32_t s_src[2][kTileHits + 2];
    __shared__ uint32_t s_topic[2][kTileHits + 2];
    const uint32_t G = gridDim.x;
    uint32_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    auto range = [&](uint32_t t, uint32_t tf0, uint32_t tf1, uint64_t& a, uint32_t& np) {
        a = pair_lo + tf0;
        const uint64_t b = (t + 1 < ntiles) ? pair_lo + tf1 + 1 : pair_hi;
        np = uint32_t(b - a);
    };
    // prologue: stage the first tile the plain way
    uint64_t a; uint32_t np;
    range(tile, tile_first[tile], tile + 1 < ntiles ? tile_first[tile + 1] : 0u, a, np);
    {
        const uint64_t base = hit_lo + uint64_t(tile) * kTileHits;
        for (uint32_t i = threadIdx.x; i < np; i += kThreads) tile_pair_view(c, a, i, base, s_off[0][i], s_src[0][i], s_topic[0][i]);
    }
    uint32_t nt = tile + G;                                   // next tile of this block
    uint32_t tf0 = 0, tf1 = 0;                                // its tile_first entries, loaded one iteration ahead
    if (nt < ntiles) { tf0 = tile_first[nt]; tf1 = nt + 1 < ntiles ? tile_first[nt + 1] : 0u; }
    __syncthreads();
    int buf = 0;
    for (;;) {
        const uint64_t base = hit_lo + uint64_t(tile) * kTileHits;
        const uint32_t len = (hit_hi - base) < uint64_t(kTileHits) ? uint32_t(hit_hi - base) : uint32_t(kTileHits);
        const bool has_next = nt < ntiles;
        // (1) issue the next tile's pair loads (first kThreads pairs) and the tile-after-next's tile_first
        uint64_t a2 = 0, base2 = 0; uint32_t np2 = 0;
        uint64_t po2 = 0; uint32_t sr2 = 0, tp2 = 0;
        uint32_t tf0n = 0, tf1n = 0;
        if (has_next) {
            range(nt, tf0, tf1, a2, np2);
            base2 = hit_lo + uint64_t(nt) * kTileHits;
            if (threadIdx.x < np2) { po2 = c.pair_off[a2 + threadIdx.x]; sr2 = c.pair_src[a2 + threadIdx.x]; tp2 = c.pair_topic[a2 + threadIdx.x]; }
            const uint32_t nn = nt + G;
            if (nn < ntiles) { tf0n = tile_first[nn]; tf1n = nn + 1 < ntiles ? tile_first[nn + 1] : 0u; }
        }
        // (2) expand the current tile from LDS buffer `buf`
        expand_tile(subs, s_off[buf], s_src[buf], s_topic[buf], np, len, out + (base - hit_lo));
        // (3) park the prefetched pair view in the other buffer (tile_pair_view's arithmetic)
        if (has_next) {
            const int nb = buf ^ 1;
            if (threadIdx.x < np2) {
                const uint32_t i = threadIdx.x;
                s_off[nb][i] = i == 0 ? 0 : int32_t(po2 - base2);
                s_src[nb][i] = sr2 + (i == 0 ? uint32_t(base2 - po2) : 0u);
                s_topic[nb][i] = tp2;
            }
            for (uint32_t i = threadIdx.x + kThreads; i < np2; i += kThreads)      // tiles with more than kThreads pairs: rare
                tile_pair_view(c, a2, i, base2, s_off[nb][i], s_src[nb][i], s_topic[nb][i]);
        }
        __syncthreads();
        if (!has_next) break;
        tile = nt; nt += G; np = np2; tf0 = tf0n; tf1 = tf1n; buf ^= 1;
    }
}

int main(int argc, char** argv) {
    const int hits_log2 = argc > 1 ? std::atoi(argv[1]) : 28;
    const double mean_run = argc > 2 ? std::atof(argv[2]) : 3000.0;
    const uint64_t pool = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 10000000ull;
    const int reps = argc > 4 ? std::atoi(argv[4]) : 10;
    const uint64_t H_target = 1ull << hits_log2;
    // ---- synthetic pair list: log-normal-ish run lengths (many short runs, a few very long ones), ~40 pairs per topic
    std::mt19937_64 rng(12345);
    std::lognormal_distribution<double> ln(std::log(mean_run) - 2.0, 2.0);
    std::vector<uint64_t> pair_off;
    std::vector<uint32_t> pair_src, pair_topic;
    uint64_t H = 0;
    uint32_t topic = 0, in_topic = 0;
    while (H < H_target) {
        uint64_t n = std::max<uint64_t>(1, std::min<uint64_t>(uint64_t(ln(rng)), pool / 2));
        if (H + n > H_target) n = H_target - H;
        pair_off.push_back(H);
        pair_src.push_back(uint32_t(rng() % (pool - n + 1)));
        pair_topic.push_back(topic);
        H += n;
        if (++in_topic == 40) { in_topic = 0; ++topic; }
    }
    pair_off.push_back(H);
    const uint64_t P = pair_src.size();
    std::vector<SubEntry> subs(pool);
    for (uint64_t i = 0; i < pool; ++i) subs[i] = SubEntry{uint32_t(i * 2654435761u), uint32_t(i % 3)};
    std::printf("pairs %llu, hits %llu (%.2f GiB of tuples), pool %llu entries (%.0f MiB), mean run %.0f\n", (unsigned long long)P,
                (unsigned long long)H, H * 12.0 / (1ull << 30), (unsigned long long)pool, pool * 8.0 / (1 << 20), double(H) / P);
    // ---- device
    SubEntry* d_subs; uint64_t* d_off; uint32_t *d_src, *d_topic, *d_tf; Tuple *d_outA, *d_outB;
    const uint32_t ntiles = uint32_t((H + kTileHits - 1) / kTileHits);
    CHECK(hipMalloc(&d_subs, pool * sizeof(SubEntry)));
    CHECK(hipMalloc(&d_off, (P + 2) * 8));
    CHECK(hipMalloc(&d_src, (P + 1) * 4));
    CHECK(hipMalloc(&d_topic, (P + 1) * 4));
    CHECK(hipMalloc(&d_tf, (size_t(ntiles) + 1) * 4));
    CHECK(hipMalloc(&d_outA, H * sizeof(Tuple)));
    CHECK(hipMalloc(&d_outB, H * sizeof(Tuple)));
    CHECK(hipMemcpy(d_subs, subs.data(), pool * sizeof(SubEntry), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_off, pair_off.data(), (P + 1) * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_src, pair_src.data(), P * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_topic, pair_topic.data(), P * 4, hipMemcpyHostToDevice));
    ChunkArrays c{};
    c.pair_off = d_off; c.pair_src = d_src; c.pair_topic = d_topic;
    tiles_k<<<uint32_t((P + 255) / 256), 256>>>(d_off, 0, P, 0, d_tf);
    CHECK(hipDeviceSynchronize());
    int dev = 0, cus = 0;
    CHECK(hipGetDevice(&dev));
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto time_it = [&](const char* name, auto launch) {
        launch();                                             // warm-up
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        std::printf("%-28s %8.4f ms   %7.1f G hits/s   stores %6.2f TB/s   algorithmic (20 B/hit) %6.2f TB/s\n", name, ms, H / ms / 1e6,
                    H * 12.0 / ms / 1e9, H * 20.0 / ms / 1e9);
    };
    time_it("A expand_ref (tile/block)", [&] { expand_ref<<<ntiles, kThreads>>>(d_subs, c, 0, P, 0, H, d_tf, ntiles, d_outA); });
    for (int per_cu : {2, 3, 4}) {
        const uint32_t G = std::min<uint32_t>(ntiles, uint32_t(cus) * per_cu);
        char name[64];
        std::snprintf(name, sizeof name, "B expand_pipe (%d blocks/CU)", per_cu);
        time_it(name, [&] { expand_pipe<<<G, kThreads>>>(d_subs, c, 0, P, 0, H, d_tf, ntiles, d_outB); });
    }
    // ---- identical results?
    std::vector<Tuple> ha(std::min<uint64_t>(H, 1u << 22)), hb(ha.size());
    bool same = true;
    for (uint64_t off : {uint64_t(0), H / 2, H - ha.size()}) {
        off = std::min(off, H - ha.size());
        CHECK(hipMemcpy(ha.data(), d_outA + off, ha.size() * sizeof(Tuple), hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hb.data(), d_outB + off, hb.size() * sizeof(Tuple), hipMemcpyDeviceToHost));
        // host reference for this slice
        uint64_t p = uint64_t(std::upper_bound(pair_off.begin(), pair_off.end(), off) - pair_off.begin()) - 1;
        for (uint64_t k = 0; k < ha.size(); ++k) {
            const uint64_t pos = off + k;
            while (pair_off[p + 1] <= pos) ++p;
            const SubEntry se = subs[pair_src[p] + (pos - pair_off[p])];
            const Tuple ref{pair_topic[p], se.sub_id, se.qos_flags};
            if (ha[k].topic_idx != ref.topic_idx || ha[k].sub_id != ref.sub_id || ha[k].qos_flags != ref.qos_flags ||
                hb[k].topic_idx != ref.topic_idx || hb[k].sub_id != ref.sub_id || hb[k].qos_flags != ref.qos_flags) {
                if (same) std::printf("MISMATCH at hit %llu\n", (unsigned long long)pos);
                same = false;
            }
        }
    }
    std::printf("results %s\n", same ? "identical (A == B == host reference on 3 slices)" : "DIFFER");
    return same ? 0 : 1;
}
