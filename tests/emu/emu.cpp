// TEST INFRASTRUCTURE ONLY — host emulator of the HIP matching pipeline.
//
// Runs the *same* per-lane / per-item functions the kernels run (rmqtt_amd/csrc/
// match_core.hpp) and the same host table compiler (table.cpp), sequentially on the CPU,
// with the same chunk / slot-overflow / window / tile orchestration as c_abi.cpp.  It lets
// the `-m "not gpu"` suite check the compiled table and the index arithmetic against the
// oracle without a GPU.  It is NOT a CPU fallback: the product library neither contains
// nor links this file, and rgr_create fails without a HIP device.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string_view>
#include <vector>

#include "match_core.hpp"
#include "retain.hpp"
#include "rmqtt_gpu_router.h"
#include "table.hpp"

using namespace rgr;

namespace {
struct Emu {
    HostTable table;
    RetainTable retain;
    TieredRetain tiers;                     // two-tier retained set with its two "device" images
    RetainImage tier_base, tier_delta;
    bool tier_has_delta = false;
    uint64_t tier_merges = 0, tier_delta_compiles = 0;
    uint32_t slot_cap = 32, chunk_topics = 1u << 21, lds_window = 2560, tile = 2048;
    uint64_t window_hits = 1ull << 28;
    uint64_t visited = 0, overflow_topics = 0, windows = 0, pairs = 0;
};
template <class T> T* dup(const std::vector<T>& v) {
    T* p = static_cast<T*>(std::malloc(std::max<size_t>(1, v.size()) * sizeof(T)));
    if (!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}
}  // namespace

extern "C" {

void* emu_new(uint32_t slot_cap, uint32_t chunk_topics, uint64_t window_hits, uint32_t lds_window, uint32_t tile) {
    auto* e = new Emu();
    if (slot_cap) e->slot_cap = slot_cap;
    if (chunk_topics) e->chunk_topics = chunk_topics;
    if (window_hits) e->window_hits = window_hits;
    e->lds_window = lds_window;
    if (tile) e->tile = tile;
    return e;
}
void emu_free(void* e) { delete static_cast<Emu*>(e); }
void emu_free_buf(void* p) { std::free(p); }

int32_t emu_filter_add(void* e, const char* f, uint32_t len, uint32_t* fid) { return static_cast<Emu*>(e)->table.filter_add(std::string_view(f, len), fid); }
int32_t emu_filter_find(void* e, const char* f, uint32_t len, uint32_t* fid) { return static_cast<Emu*>(e)->table.filter_find(std::string_view(f, len), fid); }
int32_t emu_filter_remove(void* e, uint32_t fid) { return static_cast<Emu*>(e)->table.filter_remove(fid); }
int32_t emu_sub_add(void* e, uint32_t fid, uint32_t sid, uint8_t qos, uint8_t flags) { return static_cast<Emu*>(e)->table.sub_add(fid, sid, qos, flags); }
int32_t emu_sub_add_ex(void* e, uint32_t fid, uint32_t sid, uint8_t qos, uint8_t flags, uint16_t node, uint32_t owner, uint32_t client) {
    auto& t = static_cast<Emu*>(e)->table;
    const int32_t rc = t.sub_add(fid, sid, qos, flags, node);
    if (rc == RGR_OK) t.sub_set_attr(sid, owner, client);
    return rc;
}
int32_t emu_sub_remove(void* e, uint32_t fid, uint32_t sid) { return static_cast<Emu*>(e)->table.sub_remove(fid, sid); }
uint64_t emu_n_nodes(void* e) { return static_cast<Emu*>(e)->table.n_nodes(); }
uint64_t emu_n_filters(void* e) { return static_cast<Emu*>(e)->table.n_filters(); }
uint64_t emu_n_subs(void* e) { return static_cast<Emu*>(e)->table.n_subs(); }
uint64_t emu_visited(void* e) { return static_cast<Emu*>(e)->visited; }
uint64_t emu_overflow_topics(void* e) { return static_cast<Emu*>(e)->overflow_topics; }
uint64_t emu_windows(void* e) { return static_cast<Emu*>(e)->windows; }

int32_t emu_snapshot_save(void* e, const char* path) { return static_cast<Emu*>(e)->table.save(path, nullptr) ? RGR_OK : RGR_EINVAL; }
int32_t emu_snapshot_load(void* e, const char* path) { return static_cast<Emu*>(e)->table.load(path, nullptr) ? RGR_OK : RGR_EINVAL; }

int32_t emu_subscribe_bulk(void* ev, const uint8_t* blob, const uint64_t* offs, uint64_t n, const uint32_t* sub_ids,
                           const uint8_t* qos, const uint8_t* flags, uint64_t* rejected) {
    auto* e = static_cast<Emu*>(ev);
    e->table.subscribe_bulk(blob, offs, n, sub_ids, qos, flags, nullptr, rejected, 3);
    return RGR_OK;
}

// The per-packet PUBLISH scan the device runs (match_core.hpp publish_scan), on the host: PubInfo per packet.
void emu_publish_scan(const uint8_t* blob, const uint64_t* offs, uint32_t n, int version, rgr::PubInfo* out) {
    for (uint32_t i = 0; i < n; ++i) rgr::publish_scan(blob, offs[i], offs[i + 1] - offs[i], version, out[i]);
}
}  // extern "C" (helpers below are C++)

namespace {
// The chunk / overflow / window / tile orchestration of c_abi.cpp, sequential on the host.
// walk_one(gt, ovf_pass, staged_words, rel, s_path, emit) runs the per-lane walk of topic gt.
// With prefill != nullptr the walk is skipped and prefill(begin, cn, pair_cnt, ovf_base, arena)
// provides every item's descriptor list in the arena (slot capacity 0), as the retain rounds do.
template <class WalkOne, class Prefill>
int32_t run_pipeline(Emu* e, uint32_t n, const std::vector<uint64_t>& tok_off, const std::vector<uint8_t>& tflags, const TrieView& tv,
                     WalkOne walk_one, Prefill prefill, bool use_prefill, uint64_t** hit_offsets_out, rgr_tuple** tuples_out, uint64_t* n_hits_out,
                     uint64_t** pair_offsets_out, uint32_t** pair_fids_out, const PublishAttr* pub = nullptr) {
    std::vector<uint64_t> hit_offsets(size_t(n) + 1, 0), pair_offsets(size_t(n) + 1, 0);
    std::vector<rgr_tuple> tuples;
    std::vector<uint32_t> pair_fids;
    const uint32_t C = use_prefill ? 0 : e->slot_cap;
    for (uint32_t begin = 0; begin < n; begin += e->chunk_topics) {
        const uint32_t cn = std::min<uint32_t>(e->chunk_topics, n - begin);
        std::vector<uint32_t> slots(size_t(C) * cn, 0xDEADBEEF), pair_cnt(cn, 0), hit_cnt(cn), pair_live(cn), ovf_list;
        std::vector<uint64_t> hit_off(size_t(cn) + 1), pair_base(size_t(cn) + 1), ovf_base(cn, 0);
        std::vector<uint32_t> arena;
        uint64_t ovf_cursor = 0;
        std::vector<uint32_t> s_path(std::max<uint32_t>(1, e->lds_window));
        auto walk = [&](uint32_t tl, bool ovf_pass) {
            const uint32_t gt = begin + tl;
            const uint32_t t0 = tl / 256 * 256;
            const uint64_t win_base = tok_off[begin + t0];
            const uint64_t span = tok_off[begin + std::min(t0 + 256, cn)] - win_base;
            const uint64_t staged = ovf_pass ? 0 : std::min<uint64_t>(span, e->lds_window);
            const uint64_t rel = tok_off[gt] - win_base;
            uint32_t cnt = 0;
            if (tflags[gt] & kTopicInvalid) { if (!ovf_pass) pair_cnt[tl] = 0; return; }
            auto emit = [&](uint32_t fid) {
                if (ovf_pass) arena[ovf_base[tl] + cnt] = fid;
                else if (cnt < C) slots[size_t(cnt) * cn + tl] = fid;
                cnt++;
            };
            const uint32_t v = walk_one(gt, staged, rel, s_path, emit);
            if (!ovf_pass) {
                e->visited += v;
                pair_cnt[tl] = cnt;
                if (cnt > C) { ovf_list.push_back(tl); ovf_base[tl] = ovf_cursor; ovf_cursor += cnt; }
            }
        };
        if (use_prefill) {
            prefill(begin, cn, pair_cnt, ovf_base, arena);
            arena.push_back(0xDEADBEEF);
        } else {
            for (uint32_t tl = 0; tl < cn; ++tl) walk(tl, false);
            arena.assign(ovf_cursor + 1, 0xDEADBEEF);
            for (uint32_t tl : ovf_list) walk(tl, true);
            e->overflow_topics += ovf_list.size();
        }
        uint32_t err = 0;
        ChunkArrays ca{};
        ca.n = cn; ca.slot_cap = C; ca.slots = slots.data(); ca.pair_cnt = pair_cnt.data(); ca.hit_cnt = hit_cnt.data();
        ca.pair_live = pair_live.data(); ca.hit_off = hit_off.data(); ca.pair_base = pair_base.data();
        ca.ovf_base = ovf_base.data(); ca.ovf_arena = arena.data(); ca.ovf_arena_cap = arena.size(); ca.error_flag = &err;
        for (uint32_t t = 0; t < cn; ++t) count_topic(tv, ca, t);
        if (err) return RGR_ECAPACITY;
        hit_off[0] = pair_base[0] = 0;
        for (uint32_t t = 0; t < cn; ++t) { hit_off[t + 1] = hit_off[t] + hit_cnt[t]; pair_base[t + 1] = pair_base[t] + pair_live[t]; }
        const uint64_t P = pair_base[cn], H = hit_off[cn];
        std::vector<uint32_t> pair_src(P + 1), pair_topic(P + 1);
        std::vector<uint64_t> pair_off(P + 2, ~0ull);
        ca.pair_src = pair_src.data(); ca.pair_topic = pair_topic.data(); ca.pair_off = pair_off.data();
        std::vector<uint8_t> pair_qr(P + 1, 0xEE);
        if (pub) { ca.pub = pub; ca.pair_qr = pair_qr.data(); }
        for (uint32_t t = 0; t < cn; ++t) compact_topic(tv, ca, begin, t);
        // EMU_DUMP_PAIRS=<file> (tests): the chunk's dense pair arrays as the expansion kernels see them — tests/hipsim runs the kernel
        // sources over them (u64 P, then P u32 sources, then P + 1 u64 offsets; one record per chunk, appended)
        if (const char* dump = std::getenv("EMU_DUMP_PAIRS")) {
            if (FILE* f = std::fopen(dump, "ab")) {
                std::fwrite(&P, 8, 1, f);
                std::fwrite(pair_src.data(), 4, P, f);
                pair_off[P] = H;
                std::fwrite(pair_off.data(), 8, P + 1, f);
                std::fclose(f);
            }
        }
        e->pairs += P;
        for (uint32_t t = 0; t < cn; ++t) {
            for (uint32_t j = 0; j < pair_cnt[t]; ++j) pair_fids.push_back(pair_fid(ca, t, pair_cnt[t], j));
            pair_offsets[begin + t + 1] = pair_fids.size();
        }
        const size_t out_base = tuples.size();
        tuples.resize(out_base + H);
        uint32_t lc = 0;
        while (lc < cn) {
            uint32_t le;
            if (hit_off[cn] - hit_off[lc] <= e->window_hits) le = cn;
            else {
                le = uint32_t(std::upper_bound(hit_off.begin() + lc, hit_off.end(), hit_off[lc] + e->window_hits) - hit_off.begin()) - 1;
                if (le <= lc) le = lc + 1;
            }
            const uint64_t hit_lo = hit_off[lc], hit_hi = hit_off[le], pair_lo = pair_base[lc], pair_hi = pair_base[le];
            const uint64_t nh = hit_hi - hit_lo;
            e->windows++;
            if (nh) {
                const uint32_t T = e->tile;
                const uint32_t ntiles = uint32_t((nh + T - 1) / T);
                std::vector<uint32_t> tile_first(ntiles, 0xDEADBEEF);
                std::vector<TileRec> tile_rec(ntiles, TileRec{0xDEADBEEF, 0, 0, 0});
                for (uint64_t p = pair_lo; p < pair_hi; ++p) {
                    tiles_pair(pair_off.data(), p, pair_lo, hit_lo, T, tile_first.data());
                    tiles_pair_rec(ca, p, pair_lo, hit_lo, T, tile_rec.data());          // what tiles_kernel writes on the device
                }
                std::vector<int32_t> s_off(T + 2);
                std::vector<uint32_t> s_src(T + 2), s_topic(T + 2);
                rgr_tuple* out = tuples.data() + out_base + hit_lo;
                std::vector<Cand> cand;
                for (uint32_t tile = 0; tile < ntiles; ++tile) {
                    const uint64_t base = hit_lo + uint64_t(tile) * T;
                    const uint32_t len = uint32_t(std::min<uint64_t>(T, hit_hi - base));
                    if (tile_first[tile] == 0xDEADBEEF) return RGR_EINVAL;
                    const uint64_t a = pair_lo + tile_first[tile];
                    const uint64_t b = (tile + 1 < ntiles) ? pair_lo + tile_first[tile + 1] + 1 : pair_hi;
                    const uint32_t np = uint32_t(b - a);
                    if (np > T + 1) return RGR_EINVAL;
                    if (tile_rec[tile].first != tile_first[tile]) return RGR_EINVAL;
                    for (uint32_t i = 0; i < np; ++i) tile_pair_view(ca, a, i, base, s_off[i], s_src[i], s_topic[i]);
                    if (np == 1) {      // the kernels' single-run fast path reads the record instead of the pair arrays: must be the same view
                        const TileRec& rc = tile_rec[tile];
                        if (s_off[0] != 0 || rc.src != s_src[0] || rc.topic != s_topic[0] || (pub && uint8_t(rc.qr) != ca.pair_qr[a])) return RGR_ESTATE;
                        s_src[0] = rc.src; s_topic[0] = rc.topic;
                    }
                    for (uint32_t pos = 0; pos < len; ++pos) {
                        const uint32_t i = locate_pair([&](uint32_t m) { return s_off[m]; }, np, int32_t(pos));
                        const uint64_t src = uint64_t(s_src[i]) + uint32_t(int32_t(pos) - s_off[i]);
                        SubEntry se = tv.subs[src];
                        if (pub) {   // expand_kernel<true>
                            const uint32_t fl = se.qos_flags >> 8;
                            PublishAttr pa{kNone, ca.pair_qr[a + i]};                       // s_qr
                            SubAttr at{kNone, kNone};
                            if ((fl & kSubV5) && tv.attrs) {
                                at = tv.attrs[src];
                                if (fl & kSubNoLocal) pa.from_id = pub[s_topic[i]].from_id;
                            }
                            bool is_cand;
                            se.qos_flags = deliver_word(se.qos_flags, pa, at, is_cand);
                            if (is_cand && at.client_idx != kNone) cand.push_back(Cand{uint32_t(base - hit_lo) + pos, at.client_idx});
                        }
                        out[(base - hit_lo) + pos] = rgr_tuple{s_topic[i], se.sub_id, se.qos_flags};
                    }
                }
                if (!cand.empty()) {   // launch_dedup: tile tables, classification, topic tables (match_core.hpp), sequentially
                    const uint32_t topic_lo = begin + lc, nt = le - lc;
                    std::vector<std::vector<Cand>> lists(ntiles);     // DeliverArgs::cand: every tile's own list, in arbitrary order
                    std::vector<uint32_t> topic_cand(nt, 0);          // DeliverArgs::topic_cand
                    auto h_off = [&](uint32_t t) { return hit_off[lc + t] - hit_lo; };       // window-relative first position of window topic t
                    auto h_off32 = [&](uint32_t t) { return uint32_t(h_off(t)); };
                    // a candidate carries its position only: its topic is the one whose hit range holds that position (topic_of_pos,
                    // match_core.hpp — what dedup_tile_kernel does); it must be the topic the tuple at that position names
                    auto topic_of = [&](const Cand& c) { return topic_of_pos(c.pos, 0u, nt - 1, h_off32); };
                    for (const Cand& c : cand) {
                        if (topic_of(c) != out[c.pos].topic_idx - topic_lo) return RGR_ESTATE;
                        lists[c.pos / T].push_back(c);
                        topic_cand[topic_of(c)]++;
                    }
                    for (auto& l : lists) std::reverse(l.begin(), l.end());
                    std::map<uint64_t, uint32_t> first;                // independent statement of types.rs:524-539
                    auto key_of = [&](const Cand& c) { return (uint64_t(out[c.pos].topic_idx) << 32) | c.client_idx; };
                    for (const Cand& c : cand) { auto it = first.find(key_of(c)); if (it == first.end() || c.pos < it->second) first[key_of(c)] = c.pos; }
                    std::vector<uint8_t> decided(nh, 0);
                    auto flag = [&](const Cand& c, bool dup) {
                        if (dup != (first[key_of(c)] != c.pos)) return false;
                        if (dup) out[c.pos].qos_flags |= kHitV5Dup;
                        decided[c.pos] = 1;
                        return true;
                    };
                    // the single-pass topic tables flag a LOSER by position (dedup_topic_insert_once): it must be a candidate that is not
                    // its client's first hit of the topic
                    std::map<uint32_t, const Cand*> by_pos;
                    for (const Cand& c : cand) by_pos[c.pos] = &c;
                    auto flag_loser = [&](uint32_t pos) {
                        auto it = by_pos.find(pos);
                        if (it == by_pos.end() || first[key_of(*it->second)] == pos) return false;
                        out[pos].qos_flags |= kHitV5Dup;
                        return true;
                    };
                    // dedup_tile_kernel
                    uint32_t tslots = 2; while (tslots < 2 * T) tslots <<= 1;
                    for (uint32_t tile = 0; tile < ntiles; ++tile) {
                        const auto& l = lists[tile];
                        if (l.size() < 2) continue;
                        if (l.size() > T || T > (1u << kDedupIdxBits)) return RGR_EINVAL;
                        const uint64_t lo = uint64_t(tile) * T, hi = lo + T;
                        std::vector<uint32_t> k_topic(l.size()), tab(tslots, kNone);
                        uint32_t inside = 0;
                        const uint32_t t_lo = topic_of_pos(uint32_t(lo), 0u, nt - 1, h_off32), t_hi = topic_of_pos(uint32_t(std::min<uint64_t>(hi, nh) - 1), 0u, nt - 1, h_off32);
                        for (size_t i = 0; i < l.size(); ++i) {
                            const uint32_t ti = topic_of_pos(l[i].pos, t_lo, t_hi, h_off32);       // the tile kernel's bounded search
                            if (ti != topic_of(l[i])) return RGR_ESTATE;
                            const bool in = h_off(ti) >= lo && h_off(ti + 1) <= hi;
                            k_topic[i] = in ? ti : kNone; inside += in;
                        }
                        if (inside < 2) continue;
                        auto kt = [&](uint32_t k) { return k_topic[k]; };
                        auto kc = [&](uint32_t k) { return l[k].client_idx; };
                        auto ld = [&](uint32_t sl) { return tab[sl]; };
                        for (uint32_t i = 0; i < l.size(); ++i)
                            if (k_topic[i] != kNone)
                                dedup_tile_insert(i, l[i].pos - uint32_t(lo), tslots - 1, kt, kc, ld,
                                                  [&](uint32_t sl, uint32_t v) { const uint32_t o = tab[sl]; if (o == kNone) tab[sl] = v; return o; },
                                                  [&](uint32_t sl, uint32_t v) { tab[sl] = std::min(tab[sl], v); });
                        for (uint32_t i = 0; i < l.size(); ++i)
                            if (k_topic[i] != kNone && !flag(l[i], dedup_tile_is_dup(i, tslots - 1, kt, kc, ld))) return RGR_ESTATE;
                    }
                    // dedup_classify_kernel + dedup_topic_kernel; a deliberately tiny table (8 slots, 4 candidates per part) so that
                    // small tests already exercise several parts and the overflow re-split
                    const uint32_t max_slots = 8, cap = 4;
                    for (uint32_t t = 0; t < nt; ++t) {
                        const uint32_t nc = topic_cand[t];
                        if (nc < 2) continue;
                        const uint64_t h0 = h_off(t), h1 = h_off(t + 1);
                        if (h0 / T == (h1 - 1) / T) continue;
                        const uint32_t parts = (nc + cap - 1) / cap;
                        for (uint32_t part = 0; part < parts; ++part) {
                            const uint32_t tile0 = uint32_t(h0 / T), tile1 = uint32_t((h1 - 1) / T);
                            const uint32_t mask = dedup_topic_slots(nc, parts, max_slots) - 1;
                            for (uint32_t S = 1;; S <<= 1) {
                                bool over = false;
                                for (uint32_t sub = 0; sub < S && !over; ++sub) {
                                    const uint64_t nparts = uint64_t(parts) * S;
                                    const uint32_t mine = part * S + sub;
                                    std::vector<unsigned long long> tab(size_t(mask) + 1, kDedupEmpty);
                                    auto sel = [&](const Cand& c) { return c.pos >= h0 && c.pos < h1 && (nparts == 1 || dedup_part(c.client_idx, nparts) == mine); };
                                    // dedup_topic_kernel's single pass: every insertion names the position that just lost
                                    for (uint32_t tile = tile0; tile <= tile1 && !over; ++tile)
                                        for (const Cand& c : lists[tile]) {
                                            if (!sel(c)) continue;
                                            bool full = false;
                                            const uint32_t loser = dedup_topic_insert_once(c.client_idx, c.pos, mask,
                                                [&](uint32_t sl, unsigned long long v) { const unsigned long long o = tab[sl]; if (o == kDedupEmpty) tab[sl] = v; return o; },
                                                [&](uint32_t sl, unsigned long long v) { const unsigned long long o = tab[sl]; tab[sl] = std::min(o, v); return o; }, full);
                                            if (full) { over = true; break; }
                                            if (loser != kNone && !flag_loser(loser)) return RGR_ESTATE;
                                        }
                                    if (over) break;
                                    // the part is complete: exactly the non-first hits of its clients carry the flag
                                    for (uint32_t tile = tile0; tile <= tile1; ++tile)
                                        for (const Cand& c : lists[tile])
                                            if (sel(c)) {
                                                if (((out[c.pos].qos_flags & kHitV5Dup) != 0) != (first[key_of(c)] != c.pos)) return RGR_ESTATE;
                                                decided[c.pos] = 1;
                                            }
                                }
                                if (!over) break;
                                if (S > (1u << 20)) return RGR_ESTATE;
                            }
                        }
                    }
                    // every candidate of a topic with at least two candidates was decided by exactly one of the two passes' rules
                    for (const Cand& c : cand)
                        if (!decided[c.pos] && topic_cand[topic_of(c)] >= 2 && first[key_of(c)] != c.pos) return RGR_ESTATE;
                }
            }
            lc = le;
        }
        for (uint32_t t = 0; t <= cn; ++t) hit_offsets[begin + t] = out_base + hit_off[t];
    }
    *hit_offsets_out = dup(hit_offsets);
    *tuples_out = dup(tuples);
    *n_hits_out = tuples.size();
    if (pair_offsets_out) *pair_offsets_out = dup(pair_offsets);
    if (pair_fids_out) *pair_fids_out = dup(pair_fids);
    return RGR_OK;
}
}  // namespace

extern "C" {

// Same outputs as rgr_match_batch (+ the matched filter ids per topic).  Arrays malloc'ed.
int32_t emu_match_deliver(void* ev, const uint8_t* blob, const uint64_t* offs, uint32_t n, const rgr_publish_attr* pub_attrs, int32_t* status,
                          uint64_t** hit_offsets_out, rgr_tuple** tuples_out, uint64_t* n_hits_out, uint64_t** pair_offsets_out,
                          uint32_t** pair_fids_out);
int32_t emu_match(void* ev, const uint8_t* blob, const uint64_t* offs, uint32_t n, int32_t* status, uint64_t** hit_offsets_out,
                  rgr_tuple** tuples_out, uint64_t* n_hits_out, uint64_t** pair_offsets_out, uint32_t** pair_fids_out) {
    return emu_match_deliver(ev, blob, offs, n, nullptr, status, hit_offsets_out, tuples_out, n_hits_out, pair_offsets_out, pair_fids_out);
}
int32_t emu_match_deliver(void* ev, const uint8_t* blob, const uint64_t* offs, uint32_t n, const rgr_publish_attr* pub_attrs, int32_t* status,
                          uint64_t** hit_offsets_out, rgr_tuple** tuples_out, uint64_t* n_hits_out, uint64_t** pair_offsets_out,
                          uint32_t** pair_fids_out) {
    auto* e = static_cast<Emu*>(ev);
    const HostTable& tb = e->table;
    std::vector<uint32_t> tokens;
    std::vector<uint64_t> tok_off(size_t(n) + 1, 0);
    std::vector<uint8_t> tflags(n);
    for (uint32_t i = 0; i < n; ++i) {
        tflags[i] = tb.tokenize_topic(std::string_view(reinterpret_cast<const char*>(blob) + offs[i], offs[i + 1] - offs[i]), tokens);
        tok_off[i + 1] = tokens.size();
        status[i] = (tflags[i] & kTopicInvalid) ? RGR_TOPIC_INVALID : RGR_TOPIC_OK;
    }
    {   // cross-check: the device tokeniser's code (match_core.hpp) must agree with the host tokeniser
        const StringDict& sd = tb.dict();
        DictView dv{sd.slots().data(), sd.slots().size() - 1, sd.entries().data(), sd.arena().data()};
        std::vector<uint32_t> tk2;
        for (uint32_t i = 0; i < n; ++i) {
            const uint8_t* p = blob + offs[i];
            const uint64_t len = offs[i + 1] - offs[i];
            uint8_t fl;
            const uint32_t L = topic_level_count(p, len, &fl);
            if (fl != tflags[i] || L != tok_off[i + 1] - tok_off[i]) return RGR_ESTATE;
            tk2.assign(L + 1, 0xDEADBEEF);
            if (!(fl & kTopicInvalid)) topic_tokens(dv, p, len, tk2.data());
            for (uint32_t d = 0; d < L; ++d) if (tk2[d] != tokens[tok_off[i] + d]) return RGR_ESTATE;
        }
    }
    tokens.push_back(0);
    std::vector<uint32_t> path_scratch(tokens.size(), 0xDEADBEEF);
    std::vector<FilterDesc> filt;
    std::vector<SubEntry> subs;
    tb.flatten_filters(filt, subs);
    filt.push_back(FilterDesc{0, 0}); subs.push_back(SubEntry{0, 0});
    TrieView tv{tb.edges().data(), uint32_t(tb.edges().size() - 1), tb.root_header(), filt.data(), subs.data()};
    std::vector<SubAttr> attrs(subs.size());
    for (size_t i = 0; i < subs.size(); ++i) attrs[i] = tb.sub_attr(subs[i].sub_id);
    if (tb.has_attrs()) tv.attrs = attrs.data();
    static_assert(sizeof(rgr_publish_attr) == sizeof(PublishAttr), "layout");
    auto walk_one = [&](uint32_t gt, uint64_t staged, uint64_t rel, std::vector<uint32_t>& s_path, auto& emit) {
        const uint64_t off0 = tok_off[gt];
        const uint32_t L = uint32_t(tok_off[gt + 1] - off0);
        return walk_topic(
            tv.root, tv.mask, L, (tflags[gt] & kTopicMeta) != 0, [&](uint32_t d) { return tokens[off0 + d]; },
            [&](uint32_t d) { return rel + d < staged ? s_path[rel + d] : path_scratch[off0 + d]; },
            [&](uint32_t d, uint32_t v) { if (rel + d < staged) s_path[rel + d] = v; else path_scratch[off0 + d] = v; }, emit,
            [&](uint32_t slot, U4& e0, U4& e1) {
                const EdgeEntry& en = tv.edges[slot];
                e0 = U4{en.parent, en.token, en.child, en.plus_slot};
                e1 = U4{en.hash_fid, en.term_fid, en.lit_lo, en.lit_hi};
            });
    };
    auto no_prefill = [](uint32_t, uint32_t, std::vector<uint32_t>&, std::vector<uint64_t>&, std::vector<uint32_t>&) {};
    return run_pipeline(e, n, tok_off, tflags, tv, walk_one, no_prefill, false, hit_offsets_out, tuples_out, n_hits_out, pair_offsets_out, pair_fids_out,
                        reinterpret_cast<const PublishAttr*>(pub_attrs));
}

// ---- RetainTree twin -----------------------------------------------------------------
int32_t emu_retain_add(void* e, const char* t, uint32_t len, uint32_t id) { return static_cast<Emu*>(e)->retain.topic_add(std::string_view(t, len), id); }
int32_t emu_retain_remove(void* e, const char* t, uint32_t len) { return static_cast<Emu*>(e)->retain.topic_remove(std::string_view(t, len)); }
uint64_t emu_retain_version(void* e) { return static_cast<Emu*>(e)->retain.version(); }
uint64_t emu_retain_topics(void* e) { return static_cast<Emu*>(e)->retain.n_topics(); }
uint64_t emu_retain_nodes(void* e) { return static_cast<Emu*>(e)->retain.n_nodes(); }
int32_t emu_retain_add_bulk(void* ev, const uint8_t* blob, const uint64_t* offs, uint64_t n, const uint32_t* ids, uint64_t* rejected) {
    auto* e = static_cast<Emu*>(ev);
    uint64_t rej = 0;
    for (uint64_t i = 0; i < n; ++i)
        if (e->retain.topic_add(std::string_view(reinterpret_cast<const char*>(blob) + offs[i], offs[i + 1] - offs[i]), ids ? ids[i] : uint32_t(i)) != RGR_OK) rej++;
    if (rejected) *rejected = rej;
    return RGR_OK;
}

}  // extern "C"

namespace {
// sentinels behind the arrays the per-lane code may read one element past
void pad_image(RetainImage& img) {
    img.vals.push_back(SubEntry{0, 0});
    img.child_ids.push_back(0);
    img.gc_ids.push_back(0);
}

// rgr_retain_match_batch against ONE compiled (and padded) image; `tk` tokenises the filters.
int32_t retain_match_on(Emu* e, const RetainTable& tk, const RetainImage& img, const uint8_t* blob, const uint64_t* offs, uint32_t n, int32_t* status,
                        uint64_t** hit_offsets_out, rgr_tuple** tuples_out, uint64_t* n_hits_out) {
    std::vector<uint32_t> tokens;
    std::vector<uint64_t> tok_off(size_t(n) + 1, 0);
    std::vector<uint8_t> tflags(n);
    for (uint32_t i = 0; i < n; ++i) {
        tflags[i] = tk.tokenize_filter(std::string_view(reinterpret_cast<const char*>(blob) + offs[i], offs[i + 1] - offs[i]), tokens);
        tok_off[i + 1] = tokens.size();
        status[i] = (tflags[i] & kTopicInvalid) ? RGR_TOPIC_INVALID : RGR_TOPIC_OK;
    }
    tokens.push_back(0);
    RetainView rv{};
    rv.edges = img.edges.data(); rv.mask = uint32_t(img.edges.size() - 1);
    rv.gc_edges = img.gc_edges.data(); rv.gc_mask = uint32_t(img.gc_edges.size() - 1); rv.gc_ids = img.gc_ids.data();
    rv.child_off = img.child_off.data(); rv.child_ids = img.child_ids.data(); rv.root_nonmeta = img.root_nonmeta;
    rv.n_nodes = img.n_nodes; rv.desc = img.desc.data(); rv.vals = img.vals.data();
    TrieView tv{};
    tv.filt = rv.desc; tv.subs = rv.vals;
    auto probe = [&](uint32_t parent, uint32_t token) -> uint32_t {
        for (uint32_t s = edge_hash(parent, token) & rv.mask;; s = (s + 1) & rv.mask) {
            const REdge& en = rv.edges[s];
            if (en.parent == kEdgeEmpty) return kNone;
            if (en.parent == parent && en.token == token) return en.child;
        }
    };
    // the level-synchronous frontier rounds of c_abi.cpp::retain_rounds, sequentially
    auto probe_gc = [&](uint32_t g, uint32_t t, uint32_t& b0, uint32_t& c0) {
        for (uint32_t s = edge_hash(g, t) & rv.gc_mask;; s = (s + 1) & rv.gc_mask) {
            const GcEdge& en = rv.gc_edges[s];
            if (en.gparent == kEdgeEmpty) { b0 = 0; c0 = 0; return; }
            if (en.gparent == g && en.token == t) { b0 = en.begin; c0 = en.count; return; }
        }
    };
    auto prefill = [&](uint32_t begin, uint32_t cn, std::vector<uint32_t>& pair_cnt, std::vector<uint64_t>& ovf_base, std::vector<uint32_t>& arena) {
        std::vector<uint32_t> ff(cn), fn(cn, 0), fdepth(cn, 0);
        for (uint32_t i = 0; i < cn; ++i) ff[i] = i;
        arena.clear();
        while (!ff.empty()) {
            const size_t m = ff.size();
            std::vector<RetainStep> st(m);
            for (size_t i = 0; i < m; ++i) {
                const uint32_t gt = begin + ff[i];
                st[i] = RetainStep{0, 0, kNone, kNone};
                if (tflags[gt] & kTopicInvalid) continue;
                const uint64_t off0 = tok_off[gt];
                const uint32_t L = uint32_t(tok_off[gt + 1] - off0);
                const uint32_t d = fdepth[ff[i]];
                st[i] = retain_step(rv, fn[i], d, L, d < L ? tokens[off0 + d] : 0u, d + 1 < L ? tokens[off0 + d + 1] : 0u, probe, probe_gc);
            }
            for (uint32_t f = 0; f < cn; ++f) {          // retain_advance_kernel
                const uint32_t gt = begin + f;
                const uint64_t off0 = tok_off[gt];
                const uint32_t L = uint32_t(tok_off[gt + 1] - off0), d = fdepth[f];
                if (d > L) continue;
                bool jump = false;
                if (d < L && !(tflags[gt] & kTopicInvalid)) jump = retain_jumps(tokens[off0 + d], d + 1 < L, d + 1 < L ? tokens[off0 + d + 1] : 0u);
                fdepth[f] = d + (jump ? 2u : 1u);
            }
            e->visited += m;
            // emit (item order) + next frontier (scatter at the exclusive scan of cnt)
            std::vector<uint32_t> nf, nn;
            for (size_t i = 0; i < m; ++i) {
                const uint32_t f = ff[i];
                if (i == 0 || ff[i - 1] != f) ovf_base[f] = arena.size();
                if (st[i].e0 != kNone) { arena.push_back(st[i].e0); pair_cnt[f]++; }
                if (st[i].e1 != kNone) { arena.push_back(st[i].e1); pair_cnt[f]++; }
                for (uint32_t k = 0; k < st[i].cnt; ++k) { nf.push_back(f); nn.push_back(retain_child(rv, st[i].payload, k)); }
            }
            ff.swap(nf); fn.swap(nn);
        }
    };
    auto no_walk = [](uint32_t, uint64_t, uint64_t, std::vector<uint32_t>&, auto&) { return 0u; };
    return run_pipeline(e, n, tok_off, tflags, tv, no_walk, prefill, true, hit_offsets_out, tuples_out, n_hits_out, nullptr, nullptr);
}
}  // namespace

extern "C" {

// Same outputs as rgr_retain_match_batch, as tuples (topic_idx = filter index, sub_id = topic id).
int32_t emu_retain_match(void* ev, const uint8_t* blob, const uint64_t* offs, uint32_t n, int32_t* status, uint64_t** hit_offsets_out,
                         rgr_tuple** tuples_out, uint64_t* n_hits_out) {
    auto* e = static_cast<Emu*>(ev);
    RetainImage img;
    e->retain.compile(img);
    pad_image(img);
    return retain_match_on(e, e->retain, img, blob, offs, n, status, hit_offsets_out, tuples_out, n_hits_out);
}

// ---- two-tier retained set: the orchestration rgr_retain_commit / rgr_retain_match_batch run in
// tiered mode (rgr_config.retain_delta_max > 0), over host images instead of device epochs.
int32_t emu_tier_add(void* e, const char* t, uint32_t len, uint32_t id) { return static_cast<Emu*>(e)->tiers.topic_add(std::string_view(t, len), id); }
int32_t emu_tier_remove(void* e, const char* t, uint32_t len) { return static_cast<Emu*>(e)->tiers.topic_remove(std::string_view(t, len)); }
uint64_t emu_tier_counter(void* ev, int which) {
    auto* e = static_cast<Emu*>(ev);
    switch (which) {
        case 0: return e->tiers.n_topics();
        case 1: return e->tiers.n_delta();
        case 2: return e->tiers.n_dead();
        case 3: return e->tier_merges;
        default: return e->tier_delta_compiles;
    }
}
int32_t emu_tier_commit(void* ev, uint64_t delta_max) {
    auto* e = static_cast<Emu*>(ev);
    if (e->tiers.wants_merge(delta_max)) {
        e->tiers.compile_base(e->tier_base);
        pad_image(e->tier_base);
        e->tier_has_delta = false;
        e->tier_merges++;
        return RGR_OK;
    }
    if (e->tiers.delta_dirty()) {
        e->tiers.compile_delta(e->tier_delta);
        pad_image(e->tier_delta);
        e->tier_has_delta = true;
        e->tier_delta_compiles++;
    }
    for (const auto& d : e->tiers.take_dead()) {          // the scatter of {topic_id, dead} into the base epoch's vals[]
        if (d.val_index >= e->tier_base.vals.size() || e->tier_base.vals[d.val_index].sub_id != d.topic_id) return RGR_ESTATE;
        e->tier_base.vals[d.val_index].qos_flags |= kRetainDead;
    }
    return RGR_OK;
}
// ids only (the merged answer); arrays malloc'ed
int32_t emu_tier_match(void* ev, const uint8_t* blob, const uint64_t* offs, uint32_t n, int32_t* status, uint64_t** hit_offsets_out,
                       uint32_t** ids_out, uint64_t* n_hits_out) {
    auto* e = static_cast<Emu*>(ev);
    uint64_t *bo = nullptr, *dof = nullptr, nb = 0, nd = 0;
    rgr_tuple *bt = nullptr, *dt = nullptr;
    int32_t rc = retain_match_on(e, e->tiers.base_table(), e->tier_base, blob, offs, n, status, &bo, &bt, &nb);
    if (rc != RGR_OK) return rc;
    std::vector<int32_t> st2(n);
    if (e->tier_has_delta) {
        rc = retain_match_on(e, e->tiers.delta_table(), e->tier_delta, blob, offs, n, st2.data(), &dof, &dt, &nd);
        if (rc != RGR_OK) return rc;
        for (uint32_t i = 0; i < n; ++i) if (st2[i] != status[i]) return RGR_ESTATE;
    }
    std::vector<uint32_t> bids(nb), bfl(nb), dids(nd);
    for (uint64_t k = 0; k < nb; ++k) { bids[k] = bt[k].sub_id; bfl[k] = bt[k].qos_flags; }
    for (uint64_t k = 0; k < nd; ++k) dids[k] = dt[k].sub_id;
    std::vector<uint64_t> oo;
    std::vector<uint32_t> oi;
    merge_tier_hits(n, bo, bids.data(), bfl.data(), dof, dids.data(), oo, oi);
    std::free(bo); std::free(bt); std::free(dof); std::free(dt);
    *hit_offsets_out = dup(oo);
    *ids_out = dup(oi);
    *n_hits_out = oi.size();
    return RGR_OK;
}

}  // extern "C"
