// Per-lane / per-item logic of the matching kernels, written once as host+device inline
// functions.  kernels.hip instantiates them inside the HIP kernels (LDS / HBM accessors);
// tests/emu instantiates the same code on the host to check the table compiler and the
// index arithmetic without a GPU.  (The emulator is test infrastructure: nothing in the
// product library calls these on the host.)
#pragma once
#include <cstdint>

#include "kernels.hpp"

namespace rgr {

// ---- tokeniser (rmqtt/src/topic.rs:357-394, 231-243), one topic per call ------------------
// FNV-1a 64 + avalanche; must equal StringDict::hash (table.cpp).
RGR_HD inline uint64_t dict_hash_finish(uint64_t h) { h ^= h >> 32; h *= 0xd6e8feb86659fd93ull; h ^= h >> 32; return h; }
constexpr uint64_t kFnvBasis = 0xcbf29ce484222325ull, kFnvPrime = 0x100000001b3ull;

RGR_HD inline uint32_t dict_find(const DictView& d, const uint8_t* s, uint32_t len, uint64_t h) {
    for (uint64_t i = h & d.mask;; i = (i + 1) & d.mask) {
        const uint32_t v = d.slots[i];
        if (!v) return kTokUnknown;
        const DictEntry e = d.entries[v - 1];
        if (e.hash == h && e.len == len) {
            const char* a = d.arena + e.off;
            bool eq = true;
            for (uint32_t k = 0; k < len; ++k) if (uint8_t(a[k]) != s[k]) { eq = false; break; }
            if (eq) return v - 1 + kTokFirst;
        }
    }
}

// One pass over the bytes of a topic.  on_level(index, token_or_hash_state...) is called per
// level with (seg_ptr, seg_len, fnv_hash_of_segment, kind) where kind: 0 literal, 1 '+', 2 '#'.
// Returns the level count, or -1 when Topic::from_str would fail; *meta = first level starts
// with '$' (Level::Metadata).
// (r7h) The scan reads its bytes through `at(i)` (every index at most once, in ascending order — the device kernels hand in a source that holds the
// aligned 8-byte word around the last index, one load per eight bytes instead of eight) and reports a level as (index, start, length, hash, kind).
template <class At, class OnLevel> RGR_HD inline int64_t scan_topic_at(At&& at, uint64_t len, bool* meta, OnLevel on_level) {
    // One outer iteration per LEVEL, a tight inner loop over its bytes (hash + "contains a wildcard character"); everything Topic::from_str decides
    // is decided where a level ends.  (Until r7h one flat loop carried all of it per byte: ~60 instructions per byte on the device, where a micro-batch
    // has one wave per SIMD and every instruction costs its full latency — 40 us to count the levels of 2 600 topics.)
    int64_t levels = 0;
    bool hash_seen = false;
    *meta = false;
    uint64_t i = 0;
    for (;;) {
        const uint64_t seg = i;
        uint64_t h = kFnvBasis;
        bool wild = false, more = false;                            // more: the level ended at a '/', another level follows
        const uint8_t first = i < len ? at(i) : uint8_t(0);         // the level's first byte (looked at only when the level has one)
        while (i < len) {
            const uint8_t c = at(i);
            ++i;
            if (c == '/') { more = true; break; }
            wild |= (c == '+') | (c == '#');
            h ^= c; h *= kFnvPrime;
        }
        const uint64_t sl = (more ? i - 1 : i) - seg;
        if (hash_seen) return -1;                                   // '#' was not the last level
        int kind = 0;
        if (sl == 1 && first == '+') kind = 1;
        else if (sl == 1 && first == '#') { kind = 2; hash_seen = true; }
        else if (wild) return -1;                                   // level merely contains '+'/'#'
        else if (sl > 0 && first == '$') { if (levels != 0) return -1; *meta = true; }
        on_level(levels, seg, uint32_t(sl), dict_hash_finish(h), kind);
        ++levels;
        if (!more) break;
    }
    return levels;
}
template <class OnLevel> RGR_HD inline int64_t scan_topic(const uint8_t* s, uint64_t len, bool* meta, OnLevel on_level) {
    return scan_topic_at([&](uint64_t i) { return s[i]; }, len, meta,
                         [&](int64_t idx, uint64_t seg, uint32_t sl, uint64_t h, int kind) { on_level(idx, s + seg, sl, h, kind); });
}
// Byte source over aligned 8-byte words (device kernels): the words around a topic lie inside its blob's allocation — the blob starts at an
// allocation's first byte and the allocation is padded to a multiple of eight (c_abi.cpp).
struct WordBytes {
    const uint8_t* s;
    uint64_t base = ~0ull, w = 0;
    RGR_HD explicit WordBytes(const uint8_t* p) : s(p) {}
    RGR_HD uint8_t operator()(uint64_t i) {
        const uint64_t a = reinterpret_cast<uint64_t>(s) + i, al = a & ~7ull;
        if (al != base) { w = *reinterpret_cast<const uint64_t*>(al); base = al; }
        return uint8_t(w >> ((a & 7ull) * 8));
    }
};

RGR_HD inline uint32_t topic_level_count(const uint8_t* s, uint64_t len, uint8_t* flags) {
    bool meta;
    const int64_t n = scan_topic(s, len, &meta, [](int64_t, const uint8_t*, uint32_t, uint64_t, int) {});
    if (n < 0) { *flags = kTopicInvalid; return 0; }
    *flags = meta ? kTopicMeta : 0;
    return uint32_t(n);
}

RGR_HD inline void topic_tokens(const DictView& d, const uint8_t* s, uint64_t len, uint32_t* out) {
    bool meta;
    scan_topic(s, len, &meta, [&](int64_t idx, const uint8_t* seg, uint32_t sl, uint64_t h, int kind) {
        out[idx] = kind == 1 ? kTokPlus : kind == 2 ? kTokHash : dict_find(d, seg, sl, h);
    });
}

// ---- PUBLISH packet scan, one framed packet per call -------------------------------------------------------
// Restates what the reference's codec does to reach the topic of a PUBLISH:
//   framing     first byte + variable-length "remaining length" (rmqtt-codec/src/v3/codec.rs:63-97,
//               utils.rs:142-155: at most 4 bytes, a 5th continuation is InvalidLength); the packet handed in
//               must be exactly one frame
//   v3 body     topic = u16 length + UTF-8 bytes (utils.rs:102-114), qos = flags bits 1-2 (3 is an error),
//               packet id (non-zero u16) when qos > 0, payload = the rest            (v3/decode.rs:110-128)
//   v5 body     the same, then the property block: varint length + properties; PUBLISH allows
//               0x01 bool, 0x02 u32 != 0, 0x03 / 0x08 utf8, 0x09 binary, 0x0B varint != 0 (repeatable),
//               0x23 u16 != 0, 0x26 utf8 pair (repeatable); anything else, or a second occurrence of a
//               single-valued one, is MalformedPacket                                 (v5/packet/publish.rs:31-101)
// Strict UTF-8 as std::str::from_utf8 (ByteString::try_from): no overlongs, no surrogates, max U+10FFFF.
RGR_HD inline bool utf8_valid(const uint8_t* s, uint32_t n) {
    uint32_t i = 0;
    while (i < n) {
        const uint8_t c = s[i];
        if (c < 0x80) { ++i; continue; }
        uint32_t need; uint8_t lo = 0x80, hi = 0xBF;
        if (c >= 0xC2 && c <= 0xDF) need = 1;
        else if (c == 0xE0) { need = 2; lo = 0xA0; }
        else if (c >= 0xE1 && c <= 0xEC) need = 2;
        else if (c == 0xED) { need = 2; hi = 0x9F; }
        else if (c == 0xEE || c == 0xEF) need = 2;
        else if (c == 0xF0) { need = 3; lo = 0x90; }
        else if (c >= 0xF1 && c <= 0xF3) need = 3;
        else if (c == 0xF4) { need = 3; hi = 0x8F; }
        else return false;
        if (need > n - 1 - i) return false;                                 // truncated sequence
        if (s[i + 1] < lo || s[i + 1] > hi) return false;
        for (uint32_t k = 2; k <= need; ++k) if ((s[i + k] & 0xC0) != 0x80) return false;
        i += need + 1;
    }
    return true;
}

// MQTT variable-length integer at p[pos..end): value and new position; 0 ok, kPubErrMalformed = ran out of bytes,
// kPubErrLength = more than 4 bytes (utils.rs:142-155)
RGR_HD inline uint8_t mqtt_varint(const uint8_t* p, uint64_t& pos, uint64_t end, uint32_t& val) {
    uint32_t shift = 0;
    val = 0;
    for (;;) {
        if (pos >= end) return kPubErrMalformed;
        const uint8_t b = p[pos++];
        val += uint32_t(b & 0x7F) << shift;
        if (!(b & 0x80)) return 0;
        if (shift >= 21) return kPubErrLength;
        shift += 7;
    }
}

RGR_HD inline void publish_scan(const uint8_t* blob, uint64_t off, uint64_t len, int version, PubInfo& out) {
    out = PubInfo{};
    const uint8_t* p = blob + off;
    auto fail = [&](uint8_t e) { out.error = e; out.topic_len = 0; };
    if (len < 2 || (p[0] >> 4) != 3) return fail(kPubErrNotPublish);
    const uint8_t flags = p[0] & 0x0F;
    uint64_t pos = 1;
    uint32_t rem = 0;
    if (uint8_t e = mqtt_varint(p, pos, len, rem)) return fail(e);
    if (uint64_t(rem) != len - pos) return fail(kPubErrLength);            // the blob entry must be exactly one frame
    const uint64_t end = len;
    if (end - pos < 2) return fail(kPubErrLength);                          // u16::decode, utils.rs:76-81
    const uint32_t tl = (uint32_t(p[pos]) << 8) | p[pos + 1];
    pos += 2;
    if (end - pos < tl) return fail(kPubErrLength);                         // Bytes::decode, utils.rs:102-108
    if (!utf8_valid(p + pos, tl)) return fail(kPubErrUtf8);                 // ByteString::try_from, utils.rs:110-114
    const uint64_t topic_pos = pos;
    pos += tl;
    const uint8_t qos = (flags >> 1) & 3;
    if (qos == 3) return fail(kPubErrMalformed);                            // QoS::try_from
    uint16_t pid = 0;
    if (qos) {
        if (end - pos < 2) return fail(kPubErrLength);
        pid = uint16_t((uint32_t(p[pos]) << 8) | p[pos + 1]);
        pos += 2;
        if (!pid) return fail(kPubErrMalformed);                            // NonZeroU16::decode
    }
    if (version >= 5) {                                                     // parse_publish_properties, publish.rs:64-101
        uint32_t plen = 0;
        if (uint8_t e = mqtt_varint(p, pos, end, plen)) return fail(e);
        if (end - pos < plen) return fail(kPubErrLength);
        const uint64_t pend = pos + plen;
        uint32_t seen = 0;                                                  // single-valued properties already read
        auto once = [&](uint32_t bit) { if (seen & bit) return false; seen |= bit; return true; };
        auto str = [&](bool utf8) -> uint8_t {                              // u16 length + bytes
            if (pend - pos < 2) return kPubErrLength;
            const uint32_t n = (uint32_t(p[pos]) << 8) | p[pos + 1];
            pos += 2;
            if (pend - pos < n) return kPubErrLength;
            if (utf8 && !utf8_valid(p + pos, n)) return kPubErrUtf8;
            pos += n;
            return 0;
        };
        while (pos < pend) {
            const uint8_t id = p[pos++];
            uint8_t e = 0;
            switch (id) {
                case 0x01: if (!once(1)) e = kPubErrMalformed; else if (pend - pos < 1) e = kPubErrLength; else if (p[pos++] > 1) e = kPubErrMalformed; break;
                case 0x02: if (!once(2)) e = kPubErrMalformed; else if (pend - pos < 4) e = kPubErrLength;
                           else { const uint32_t v = (uint32_t(p[pos]) << 24) | (uint32_t(p[pos + 1]) << 16) | (uint32_t(p[pos + 2]) << 8) | p[pos + 3]; pos += 4; if (!v) e = kPubErrMalformed; } break;
                case 0x03: if (!once(4)) e = kPubErrMalformed; else e = str(true); break;
                case 0x08: if (!once(8)) e = kPubErrMalformed; else e = str(true); break;
                case 0x09: if (!once(16)) e = kPubErrMalformed; else e = str(false); break;
                case 0x0B: { uint32_t v = 0; e = mqtt_varint(p, pos, pend, v); if (!e && !v) e = kPubErrMalformed; } break;
                case 0x23: if (!once(32)) e = kPubErrMalformed; else if (pend - pos < 2) e = kPubErrLength;
                           else { const uint32_t v = (uint32_t(p[pos]) << 8) | p[pos + 1]; pos += 2; if (!v) e = kPubErrMalformed; } break;
                case 0x26: e = str(true); if (!e) e = str(true); break;
                default: e = kPubErrMalformed;
            }
            if (e) return fail(e);
        }
    }
    out.topic_off = off + topic_pos;
    out.topic_len = tl;
    out.payload_off = uint32_t(pos);
    out.packet_id = pid;
    out.qos = qos; out.retain = flags & 1; out.dup = (flags >> 3) & 1;
}

// Depth-first walk of the subscription trie for ONE publish topic, emitting matched filter
// ids in exactly TopicTree::matches' iteration order (rmqtt/src/trie.rs:327-375;
// SURVEY.md App. A.2):
//   remaining path empty : node's own filter, then its '#' child ("parent match";
//                          trie.rs:328-338 pushes them in the opposite order and pops LIFO)
//   otherwise            : unless (root && first level is '$'-metadata, trie.rs:342-346):
//                          the '#' child's filter, then the whole '+' subtree;
//                          then the exact-child subtree.
// The DFS stack is one word per level (`path`): the node whose exact-child lookup is still
// pending at that depth, or kNone.  It is only written when a '+' branch is taken, and is
// found again by scanning down from the depth where the walk died.
//
//   tok_at(d)        level token d of the topic
//   path_get/set(d)  stack word of depth d
//   emit(fid)        next matched filter id
//   load(slot,e0,e1) read the 32-byte edge record `slot` as two 16-byte halves
// Returns the number of trie nodes visited (= MatchedIter instantiations, the nV of
// SURVEY.md §8(d)).
template <class TokAt, class PathGet, class PathSet, class Emit, class Load>
RGR_HD inline uint32_t walk_topic(const NodeHeader& root, uint32_t mask, uint32_t L, bool meta, TokAt tok_at,
                                  PathGet path_get, PathSet path_set, Emit emit, Load load) {
    enum : int { kArrive = 0, kPop = 1, kProbe = 2, kDirect = 3 };
    uint32_t visited = 0;
    uint32_t node = 0, d = 0;
    NodeHeader h = root;
    int mode = kArrive;
    int64_t scan = -1;
    uint32_t slot = 0, want_parent = 0, want_tok = 0;
    for (;;) {
        if (mode == kArrive) {
            visited++;
            if (d == L) {
                if (h.term_fid != kNone) emit(h.term_fid);
                if (h.hash_fid != kNone) emit(h.hash_fid);
                mode = kPop; scan = int64_t(d) - 1;
            } else {
                const bool wild = !(d == 0 && meta);
                if (wild && h.hash_fid != kNone) emit(h.hash_fid);      // trie.rs:349-355
                const uint32_t tk = tok_at(d);
                // miss filter: the header carries a 64-bit bitmap of the literal edges that leave this node; a clear bit
                // proves there is no such child, so dead-end probes are (almost all) skipped.  Wildcard tokens in a
                // publish topic (tk < kTokFirst) always probe (App. A.4 quirk).
                const uint32_t lb = lit_bit(tk);
                const bool ex = tk != kTokUnknown && (tk < kTokFirst || (((lb & 32u) ? h.lit_hi : h.lit_lo) >> (lb & 31u)) & 1u);
                const bool pl = wild && h.plus_slot != kNone;
                if (pl) {                                                // trie.rs:358-362
                    path_set(d, ex ? node : kNone);                      // exact lookup deferred
                    slot = h.plus_slot; mode = kDirect;
                } else if (ex) {                                         // trie.rs:366-370
                    path_set(d, kNone);
                    want_parent = node; want_tok = tk;
                    slot = edge_hash(node, tk) & mask; mode = kProbe;
                } else {
                    path_set(d, kNone);
                    mode = kPop; scan = int64_t(d) - 1;
                }
            }
        }
        if (mode == kPop) {
            uint32_t pn = kNone;
            int64_t s = scan;
            for (; s >= 0; --s) { pn = path_get(uint32_t(s)); if (pn != kNone) break; }
            if (s < 0) break;                                            // walk finished
            path_set(uint32_t(s), kNone);
            d = uint32_t(s);
            want_parent = pn; want_tok = tok_at(d);
            slot = edge_hash(pn, want_tok) & mask; mode = kProbe;
        }
        U4 e0, e1;                                                       // one 32-byte record per step
        load(slot, e0, e1);
        if (mode == kProbe) {
            if (e0.x == kEdgeEmpty) { mode = kPop; scan = int64_t(d) - 1; continue; }
            if (e0.x != want_parent || e0.y != want_tok) { slot = (slot + 1) & mask; continue; }
        }
        node = e0.z;
        h.plus_slot = e0.w; h.hash_fid = e1.x; h.term_fid = e1.y; h.lit_lo = e1.z; h.lit_hi = e1.w;
        d += 1;
        mode = kArrive;
    }
    return visited;
}

// RetainTree::matches (rmqtt/src/retain.rs:457-526), one frontier item at a
// time.  The filter batch is matched level-synchronously: the frontier of level d holds
// (filter, node) items; retain_step() says what one item does at its filter's level d:
//   cnt / payload  how many items it contributes to level d+1: 0 (dead end), 1 (exact child:
//                  payload = child | kLitFlag) or the node's whole child list (payload = first
//                  index into child_ids; the root skips '$' children, retain.rs:486-490)
//   e0 / e1        run descriptors it emits (RetainView::desc indices), kNone if none:
//     path exhausted at node p             -> 2p   (p's own value, retain.rs:465-470)
//     next level is the trailing '#' at p  -> 2p+1 (p itself = "parent match" retain.rs:476-481
//                                             / 494-499, plus every descendant retain.rs:502-524)
//                                             or 2N at the root (no '$' subtrees, retain.rs:505-509)
// An exact key equal to a wildcard token takes precedence over the wildcard meaning (the
// else-if chain at retain.rs:472/483/502).  Expansion keeps items in order, so every filter's
// descriptors come out in ascending trie preorder — deterministic.
constexpr uint32_t kLitFlag = 0x80000000u;   // payload = one node id
constexpr uint32_t kGcFlag = 0x40000000u;    // payload = first index of a gc_ids run
struct RetainStep { uint32_t cnt, payload, e0, e1; };

// A '+' level followed by a literal level is taken in one jump of two levels (grandchild index).
// Depends on the filter's tokens only, so all items of a filter always stand at the same level.
RGR_HD inline bool retain_jumps(uint32_t tok, bool has_next, uint32_t tok_next) {
    return tok == kTokPlus && has_next && tok_next >= kTokFirst;
}

template <class Probe, class ProbeGc>
RGR_HD inline RetainStep retain_step(const RetainView& rv, uint32_t node, uint32_t d, uint32_t L, uint32_t tok, uint32_t tok_next,
                                     Probe probe, ProbeGc probe_gc) {
    RetainStep r{0, 0, kNone, kNone};
    if (d == L) { r.e0 = 2 * node; return r; }
    if (tok == kTokHash) {
        // trailing '#': the node's own value ("parent match", retain.rs:476-481/494-499) + everything
        // `node._matches(["#"])` yields (retain.rs:502-524).  The second run is laid out by
        // RetainTable::compile, including the reference's exact-first rule (retain.rs:472): below a
        // node that stores a literal "#" level only that child's value is visible.
        if (node != 0) r.e0 = 2 * node;
        r.e1 = node == 0 ? 2 * rv.n_nodes : 2 * node + 1;
        return r;
    }
    if (tok == kTokPlus) {
        const bool jump = retain_jumps(tok, d + 1 < L, tok_next);
        const uint32_t c = probe(node, kTokPlus);
        if (c != kNone) {                                // literal "+" key: exact-first (retain.rs:472)
            if (!jump) { r.cnt = 1; r.payload = c | kLitFlag; return r; }
            const uint32_t c2 = tok_next == kTokUnknown ? kNone : probe(c, tok_next);   // both levels in this round
            if (c2 != kNone) { r.cnt = 1; r.payload = c2 | kLitFlag; }
            return r;
        }
        if (jump) {                                      // all grandchildren carrying tok_next, one probe
            if (tok_next == kTokUnknown) return r;
            uint32_t begin = 0, count = 0;
            probe_gc(node, tok_next, begin, count);
            r.cnt = count; r.payload = begin | kGcFlag;
            return r;
        }
        const uint32_t b = rv.child_off[node];
        const uint32_t e = node == 0 ? b + rv.root_nonmeta : rv.child_off[node + 1];
        r.cnt = e - b; r.payload = b;
        return r;
    }
    if (tok == kTokUnknown) return r;
    const uint32_t c = probe(node, tok);
    if (c != kNone) { r.cnt = 1; r.payload = c | kLitFlag; }
    return r;
}

// k-th item an expansion (cnt, payload) contributes to the next frontier.
RGR_HD inline uint32_t retain_child(const RetainView& rv, uint32_t payload, uint32_t k) {
    if (payload & kLitFlag) return payload & ~kLitFlag;
    if (payload & kGcFlag) return rv.gc_ids[(payload & ~kGcFlag) + k];
    return rv.child_ids[payload + k];
}

// j-th matched filter of chunk-local topic t (slots are j-major; topics whose count
// exceeded the slot capacity live in the overflow arena).
RGR_HD inline uint32_t pair_fid(const ChunkArrays& c, uint32_t t, uint32_t cnt, uint32_t j) {
    return cnt <= c.slot_cap ? c.slots[uint64_t(j) * c.n + t] : c.ovf_arena[c.ovf_base[t] + j];
}

// Per-topic hit count and number of matched filters that have subscribers.
RGR_HD inline void count_topic(const TrieView& tv, const ChunkArrays& c, uint32_t t) {
    const uint32_t cnt = c.pair_cnt[t];
    uint32_t hits = 0, live = 0;
    if (cnt > c.slot_cap && c.ovf_base[t] + cnt > c.ovf_arena_cap) {
        *c.error_flag = 1; c.hit_cnt[t] = 0; c.pair_live[t] = 0;
        return;
    }
    for (uint32_t j = 0; j < cnt; ++j) {
        const uint32_t n = tv.filt[pair_fid(c, t, cnt, j)].count;
        hits += n; live += n != 0;
    }
    c.hit_cnt[t] = hits;
    c.pair_live[t] = live;
}

// Dense (topic, subscriber-run) pairs of topic t with their chunk-local output offsets.
RGR_HD inline void compact_topic(const TrieView& tv, const ChunkArrays& c, uint32_t topic_base, uint32_t t) {
    const uint32_t cnt = c.pair_cnt[t];
    uint64_t o = c.hit_off[t];
    uint64_t p = c.pair_base[t];
    if (c.pair_live[t]) {
        for (uint32_t j = 0; j < cnt; ++j) {
            const FilterDesc fd = tv.filt[pair_fid(c, t, cnt, j)];
            if (fd.count) {
                c.pair_src[p] = fd.begin;
                c.pair_topic[p] = c.topic_ids ? c.topic_ids[topic_base + t] : topic_base + t;
                if (c.pair_qr) c.pair_qr[p] = uint8_t(c.pub[topic_base + t].qos_retain);
                c.pair_off[p] = o;
                ++p; o += fd.count;
            }
        }
    }
    if (t == c.n - 1) c.pair_off[p] = o;   // sentinel: total hits of the chunk
}

// Pair p covers output positions [pair_off[p], pair_off[p+1]); it owns every tile whose first
// position falls inside that range.
RGR_HD inline void tiles_pair(const uint64_t* pair_off, uint64_t p, uint64_t pair_lo, uint64_t hit_lo, uint32_t tile,
                              uint32_t* tile_first) {
    const uint64_t s = pair_off[p] - hit_lo, e = pair_off[p + 1] - hit_lo;
    for (uint64_t k = (s + tile - 1) / tile; k * tile < e; ++k) tile_first[k] = uint32_t(p - pair_lo);
}

// The same with the tile's record: besides the first pair, that pair's own view at the tile's first position (where in
// subs[] the tile starts reading, its topic, its publish qos|retain) — a tile that lies inside one run needs nothing else.
RGR_HD inline void tiles_pair_rec(const ChunkArrays& c, uint64_t p, uint64_t pair_lo, uint64_t hit_lo, uint32_t tile, TileRec* tile_rec) {
    const uint64_t s = c.pair_off[p] - hit_lo, e = c.pair_off[p + 1] - hit_lo;
    const uint32_t src = c.pair_src[p], topic = c.pair_topic[p];
    const uint32_t qr = c.pair_qr ? c.pair_qr[p] : 0u;
    for (uint64_t k = (s + tile - 1) / tile; k * tile < e; ++k)
        tile_rec[k] = TileRec{uint32_t(p - pair_lo), src + uint32_t(k * tile - s), topic, qr};
}

// Largest i in [0,np) with off[i] <= pos (off[0] <= pos is guaranteed by the caller).
template <class OffAt> RGR_HD inline uint32_t locate_pair(OffAt off_at, uint32_t np, int32_t pos) {
    uint32_t lo = 0, hi = np;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off_at(mid) <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

// Tile-local view of pair i of a tile starting at output position `base`: offset of the run
// relative to the tile (0 for the first pair, whose skipped prefix is folded into src).
RGR_HD inline void tile_pair_view(const ChunkArrays& c, uint64_t a, uint32_t i, uint64_t base, int32_t& off, uint32_t& src,
                                  uint32_t& topic) {
    const uint64_t po = c.pair_off[a + i];
    off = i == 0 ? 0 : int32_t(po - base);
    src = c.pair_src[a + i] + (i == 0 ? uint32_t(base - po) : 0u);
    topic = c.pair_topic[a + i];
}

// ---- delivery stage --------------------------------------------------------
// The per-hit part of DefaultRouter::_matches (router.rs:194-201) and forwards_to
// (shared.rs:886-903): qos downgrade, Retain-As-Published, No Local.  `candidate` = the hit goes
// through the v5 collector keyed by client (types.rs:524-539): v5, not a $share member (those go
// through SharedSubscription::choice on the host, router.rs:202-255), not dropped.
RGR_HD inline uint32_t deliver_word(uint32_t qos_flags, PublishAttr pa, SubAttr at, bool& candidate) {
    const uint32_t fl = (qos_flags >> 8) & 0xFFu;
    const uint32_t sq = qos_flags & 0xFFu, pq = pa.qos_retain & 3u;
    uint32_t w = (qos_flags & 0xFFFFFF00u) | (sq < pq ? sq : pq);
    const bool v5 = (fl & kSubV5) != 0;
    if (v5 && (fl & kSubRap) && (pa.qos_retain & 4u)) w |= kHitRetain;
    const bool dropped = v5 && (fl & kSubNoLocal) && pa.from_id != kNone && at.owner_id == pa.from_id;
    if (dropped) w |= kHitNoLocal;
    candidate = v5 && !(fl & kSubShared) && !dropped;
    return w;
}

// ---- v5 per-client dedup (types.rs:524-539) in LDS tables (r3) --------------------------------------------------
// Among the v5 hits ("candidates") of ONE publish topic the first hit of every client — lowest position — keeps filter and
// options; every later one is flagged kHitV5Dup.  Duplicates only exist among the hits of one topic and a topic's hits are
// consecutive positions, so the (topic, client) -> first-position table never has to be global:
//   tile table   every topic that lies entirely inside one expansion tile is resolved by that tile's block: u32 slots,
//                value = position-in-tile << 11 | index in the tile's candidate list (min = first position; the index leads
//                to the key), key = (topic, client) of the candidate the value names
//   topic table  a topic that spans tiles is resolved by a block of its own walking the candidate lists of its tiles: u64
//                slots, client << 32 | position.  A topic with more candidates than a table holds is split into PARTS by
//                client (dedup_part: a range partition of a bijective mix of the client index, so no client is in two
//                parts); a part that still overflows (distinct clients > slots) is re-split on the fly.
// The same functions run in kernels.hip (LDS, ds atomics) and in tests/emu (plain memory).
// (mix32: kernels.hpp)
RGR_HD inline uint32_t dedup_part(uint32_t client, uint64_t nparts) { return uint32_t((uint64_t(mix32(client)) * nparts) >> 32); }

constexpr uint32_t kDedupIdxBits = 11;               // candidates per tile (and positions per tile) <= 2048
template <class KeyTopic, class KeyClient, class Load, class Cas, class Min>
RGR_HD inline void dedup_tile_insert(uint32_t i, uint32_t pos_in_tile, uint32_t mask, KeyTopic key_topic, KeyClient key_client, Load tab_load, Cas tab_cas,
                                     Min tab_min) {
    const uint32_t topic = key_topic(i), client = key_client(i);
    const uint32_t v = (pos_in_tile << kDedupIdxBits) | i;
    for (uint32_t s = edge_hash(topic, client) & mask;; s = (s + 1) & mask) {      // the table has more slots than a tile has candidates
        uint32_t cur = tab_load(s);
        if (cur == kNone) { cur = tab_cas(s, v); if (cur == kNone) return; }
        const uint32_t k = cur & ((1u << kDedupIdxBits) - 1u);
        if (key_topic(k) == topic && key_client(k) == client) { tab_min(s, v); return; }
    }
}
template <class KeyTopic, class KeyClient, class Load>
RGR_HD inline bool dedup_tile_is_dup(uint32_t i, uint32_t mask, KeyTopic key_topic, KeyClient key_client, Load tab_load) {
    const uint32_t topic = key_topic(i), client = key_client(i);
    for (uint32_t s = edge_hash(topic, client) & mask;; s = (s + 1) & mask) {
        const uint32_t cur = tab_load(s);
        if (cur == kNone) return false;                                              // unreachable: every candidate was inserted
        const uint32_t k = cur & ((1u << kDedupIdxBits) - 1u);
        if (key_topic(k) == topic && key_client(k) == client) return k != i;
    }
}

constexpr unsigned long long kDedupEmpty = ~0ull;
// false = the table is full (the caller re-splits the part)
template <class Cas, class Min>
RGR_HD inline bool dedup_topic_insert(uint32_t client, uint32_t pos, uint32_t mask, Cas tab_cas, Min tab_min) {
    const unsigned long long mine = (static_cast<unsigned long long>(client) << 32) | pos;
    uint32_t steps = 0;
    for (uint32_t s = mix32(client) & mask;; s = (s + 1) & mask) {
        const unsigned long long prev = tab_cas(s, mine);
        if (prev == kDedupEmpty) return true;
        if (uint32_t(prev >> 32) == client) { tab_min(s, mine); return true; }        // same client: the smaller position wins
        if (++steps > mask) return false;
    }
}
// Single pass (r4): insert and learn at once who LOST.  tab_min returns the slot's value from before the min.  Every candidate that
// is not its client's first position loses exactly once — either on arrival (a smaller position is already there: it is the loser
// itself) or later, when a smaller position arrives and gets it back as the old value — so flagging the returned position flags
// exactly the duplicates, without a second pass over the candidate lists.  Returns kNone when nobody lost; full = the table has no
// room (the caller re-splits the part; flags set so far stay valid).
// step_mode 0: linear probing; 1: double hashing — an odd step taken from the upper half of the client's mix (the table length is a power of
// two, so every odd step visits every slot): no primary clustering, shorter worst probe of a wave's 64 lanes.
template <class Cas, class Min>
RGR_HD inline uint32_t dedup_topic_insert_once(uint32_t client, uint32_t pos, uint32_t mask, Cas tab_cas, Min tab_min, bool& full, uint32_t step_mode = 0) {
    const unsigned long long mine = (static_cast<unsigned long long>(client) << 32) | pos;
    uint32_t steps = 0;
    const uint32_t hm = mix32(client);
    const uint32_t step = step_mode ? ((hm >> 16) | 1u) : 1u;
    for (uint32_t s = hm & mask;; s = (s + step) & mask) {
        const unsigned long long prev = tab_cas(s, mine);
        if (prev == kDedupEmpty) return kNone;
        if (uint32_t(prev >> 32) == client) {
            const unsigned long long old = tab_min(s, mine);
            return old < mine ? pos : uint32_t(old);
        }
        if (++steps > mask) { full = true; return kNone; }
    }
}
// window topic of a window-relative position: the last t in [t_lo, t_hi] with hit_off(t) <= pos (hit_off: window-relative first positions)
template <class HitOff>
RGR_HD inline uint32_t topic_of_pos(uint32_t pos, uint32_t t_lo, uint32_t t_hi, HitOff hit_off) {
    while (t_lo < t_hi) {
        const uint32_t mid = t_lo + (t_hi - t_lo + 1) / 2;
        if (hit_off(mid) <= pos) t_lo = mid; else t_hi = mid - 1;
    }
    return t_lo;
}
template <class Load>
RGR_HD inline bool dedup_topic_is_dup(uint32_t client, uint32_t pos, uint32_t mask, Load tab_load) {
    for (uint32_t s = mix32(client) & mask;; s = (s + 1) & mask) {
        const unsigned long long e = tab_load(s);
        if (e == kDedupEmpty) return false;                                          // unreachable
        if (uint32_t(e >> 32) == client) return uint32_t(e) != pos;
    }
}
// Slots a part's table uses: a power of two >= 2 x its expected candidates (+ slack), within [64, max_slots].
RGR_HD inline uint32_t dedup_topic_slots(uint32_t nc, uint32_t parts, uint32_t max_slots, uint32_t factor = 2u) {
    const uint32_t want = factor * ((nc + parts - 1) / parts) + 16u;
    uint32_t len = 64;
    while (len < want && len < max_slots) len <<= 1;
    return len;
}

}  // namespace rgr
