// See raft_snapshot.hpp.  Host-only code: no device calls here.
#include "raft_snapshot.hpp"

#include <dlfcn.h>
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <exception>
#include <mutex>

namespace rmqtt {
namespace raft {

namespace {

// ---------------------------------------------------------------------------------------------
// postcard 1.x wire format (the `postcard::from_bytes` side of router.rs:505-509)
//   u8 / i8: one byte.  u16..u64, usize: LEB128 varint (at most 3 / 5 / 10 / 10 bytes; an encoding
//   that overflows the type is an error, a padded one is accepted).  i16..i64, isize: zigzag, then varint.
//   bool: 0 | 1.  Option: 0 | 1 + value.  str / bytes: varint length + bytes (str: valid UTF-8).
//   seq / map: varint length + items (map: key, value pairs).  tuple / struct: the fields, in order.
//   enum: varint(u32) variant index + the variant's content.
// ---------------------------------------------------------------------------------------------
struct Reader {
    const uint8_t* p;
    size_t n, pos = 0;
    std::string err;        // first error; once set every take fails
    const char* section;

    bool fail(const std::string& what) {
        if (err.empty()) err = std::string(section) + ": " + what + " at byte " + std::to_string(pos);
        return false;
    }
    bool byte(uint8_t& b) {
        if (!err.empty()) return false;
        if (pos >= n) return fail("unexpected end of data");
        b = p[pos++];
        return true;
    }
    bool varint(uint64_t& out, int bits) {
        const int max_bytes = (bits + 6) / 7;
        const uint8_t last_max = uint8_t((1u << (bits - 7 * (max_bytes - 1))) - 1);   // what the top byte may hold
        out = 0;
        for (int i = 0; i < max_bytes; ++i) {
            uint8_t b = 0;
            if (!byte(b)) return false;
            out |= uint64_t(b & 0x7F) << (7 * i);
            if (!(b & 0x80)) {
                if (i == max_bytes - 1 && b > last_max) return fail("varint overflows its type");
                return true;
            }
        }
        return fail("varint too long");
    }
    bool u16(uint16_t& v) { uint64_t x; if (!varint(x, 16)) return false; v = uint16_t(x); return true; }
    bool u32(uint32_t& v) { uint64_t x; if (!varint(x, 32)) return false; v = uint32_t(x); return true; }
    bool u64(uint64_t& v) { return varint(v, 64); }
    bool i64(int64_t& v) {
        uint64_t x;
        if (!varint(x, 64)) return false;
        v = int64_t(x >> 1) ^ -int64_t(x & 1);
        return true;
    }
    bool boolean(bool& v) {
        uint8_t b = 0;
        if (!byte(b)) return false;
        if (b > 1) { --pos; return fail("bad bool"); }
        v = b != 0;
        return true;
    }
    bool option(bool& some) {
        uint8_t b = 0;
        if (!byte(b)) return false;
        if (b > 1) { --pos; return fail("bad Option tag"); }
        some = b != 0;
        return true;
    }
    // a length that the remaining bytes can still hold (every item takes at least min_item bytes): a corrupt
    // length must not become a huge allocation
    bool length(uint64_t& len, size_t min_item) {
        if (!u64(len)) return false;
        if (len > (n - pos) / (min_item ? min_item : 1)) return fail("length " + std::to_string(len) + " exceeds the data");
        return true;
    }
    bool str(std::string& s) {
        uint64_t len;
        if (!length(len, 1)) return false;
        if (!utf8(p + pos, size_t(len))) return fail("string is not UTF-8");
        s.assign(reinterpret_cast<const char*>(p + pos), size_t(len));
        pos += size_t(len);
        return true;
    }
    bool raw(uint8_t* dst, size_t k) {
        if (!err.empty()) return false;
        if (n - pos < k) return fail("unexpected end of data");
        std::memcpy(dst, p + pos, k);
        pos += k;
        return true;
    }
    // core::str::from_utf8: no overlongs, no surrogates, nothing above U+10FFFF
    static bool utf8(const uint8_t* s, size_t len) {
        size_t i = 0;
        while (i < len) {
            const uint8_t c = s[i];
            if (c < 0x80) { ++i; continue; }
            int k; uint32_t cp, lo;
            if ((c & 0xE0) == 0xC0) { k = 1; cp = c & 0x1F; lo = 0x80; }
            else if ((c & 0xF0) == 0xE0) { k = 2; cp = c & 0x0F; lo = 0x800; }
            else if ((c & 0xF8) == 0xF0) { k = 3; cp = c & 0x07; lo = 0x10000; }
            else return false;
            if (len - i <= size_t(k)) return false;
            for (int j = 1; j <= k; ++j) {
                if ((s[i + j] & 0xC0) != 0x80) return false;
                cp = (cp << 6) | (s[i + j] & 0x3F);
            }
            if (cp < lo || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
            i += size_t(k) + 1;
        }
        return true;
    }
};

// std::net::Ipv6Addr's Display: "::ffff:a.b.c.d" for IPv4-mapped, otherwise lower-case hex groups
// with the longest run (two or more) of zero groups — the first one on ties — written as "::".
std::string ipv6_text(const uint8_t* o) {
    uint16_t g[8];
    for (int i = 0; i < 8; ++i) g[i] = uint16_t(o[2 * i] << 8 | o[2 * i + 1]);
    char buf[64];
    if (!g[0] && !g[1] && !g[2] && !g[3] && !g[4] && g[5] == 0xFFFF) {
        std::snprintf(buf, sizeof buf, "::ffff:%u.%u.%u.%u", unsigned(o[12]), unsigned(o[13]), unsigned(o[14]), unsigned(o[15]));
        return buf;
    }
    int best = -1, best_len = 0;
    for (int i = 0; i < 8;) {
        if (g[i]) { ++i; continue; }
        int j = i;
        while (j < 8 && !g[j]) ++j;
        if (j - i > best_len) { best = i; best_len = j - i; }
        i = j;
    }
    if (best_len < 2) best = -1;
    std::string s;
    for (int i = 0; i < 8;) {
        if (i == best) { s += "::"; i += best_len; continue; }
        if (!s.empty() && s.back() != ':') s.push_back(':');
        std::snprintf(buf, sizeof buf, "%x", unsigned(g[i]));
        s += buf;
        ++i;
    }
    return s;
}

// Option<SocketAddr> (types.rs:1903-1906) -> "" | "a.b.c.d:port" | "[v6]:port" (std's Display)
bool option_socket_addr(Reader& r, std::string& out) {
    out.clear();
    bool some = false;
    if (!r.option(some)) return false;
    if (!some) return true;
    uint32_t variant;
    if (!r.u32(variant)) return false;
    uint8_t oct[16];
    uint16_t port;
    if (variant == 0) {
        if (!r.raw(oct, 4) || !r.u16(port)) return false;
        char buf[32];
        std::snprintf(buf, sizeof buf, "%u.%u.%u.%u:%u", unsigned(oct[0]), unsigned(oct[1]), unsigned(oct[2]), unsigned(oct[3]), unsigned(port));
        out = buf;
    } else if (variant == 1) {
        if (!r.raw(oct, 16) || !r.u16(port)) return false;
        out = "[" + ipv6_text(oct) + "]:" + std::to_string(port);
    } else {
        return r.fail("SocketAddr variant " + std::to_string(variant));
    }
    return true;
}

// _Id (types.rs:1899-1911)
bool read_id(Reader& r, Id& id) {
    bool some = false;
    if (!r.u64(id.node_id) || !r.u16(id.lid) || !option_socket_addr(r, id.local_addr) || !option_socket_addr(r, id.remote_addr) ||
        !r.str(id.client_id) || !r.option(some))
        return false;
    id.username.clear();
    if (some && !r.str(id.username)) return false;
    return r.i64(id.create_time);
}

bool read_qos(Reader& r, uint8_t& qos) {       // types.rs:717-731: one u8, 0..=2
    if (!r.byte(qos)) return false;
    if (qos > 2) { --r.pos; return r.fail("invalid QoS value, " + std::to_string(qos)); }
    return true;
}

// SubscriptionOptions (types.rs:607-610) = V3(SubOptionsV3 :769-779) | V5(SubOptionsV5 :803-821)
bool read_opts(Reader& r, const Features& f, SubscriptionOptions& o, std::optional<uint64_t>& limit_subs) {
    uint32_t variant;
    if (!r.u32(variant)) return false;
    if (variant > 1) return r.fail("SubscriptionOptions variant " + std::to_string(variant));
    o = SubscriptionOptions{};
    o.v5 = variant == 1;
    limit_subs.reset();
    bool some = false;
    if (!read_qos(r, o.qos)) return false;
    if (f.shared_subscription) {
        if (!r.option(some)) return false;
        if (some) { std::string g; if (!r.str(g)) return false; o.shared_group = std::move(g); }
    }
    if (f.limit_subscription) {
        if (!r.option(some)) return false;
        if (some) { uint64_t v; if (!r.u64(v)) return false; limit_subs = v; }
    }
    if (!o.v5) return true;
    if (!r.boolean(o.no_local) || !r.boolean(o.retain_as_published) || !r.byte(o.retain_handling)) return false;
    if (o.retain_handling > 2) { --r.pos; return r.fail("invalid RetainHandling value, " + std::to_string(o.retain_handling)); }   // types.rs:864-877
    if (!r.option(some)) return false;
    if (some) {
        if (!r.u32(o.subscription_identifier)) return false;
        if (o.subscription_identifier == 0) return r.fail("NonZeroU32 subscription identifier is 0");
    }
    return true;
}

bool read_counter(Reader& r, CounterState& c) {   // counter.rs:39: (isize, isize, StatsMergeMode :337-343)
    if (!r.i64(c.count) || !r.i64(c.max) || !r.u32(c.merge_mode)) return false;
    if (c.merge_mode > 4) return r.fail("StatsMergeMode variant " + std::to_string(c.merge_mode));
    return true;
}

// ---------------------------------------------------------------------------------------------
// decompression (router.rs:470-493)
// ---------------------------------------------------------------------------------------------
using Bytes = std::vector<uint8_t>;
constexpr size_t kMaxPlain = size_t(1) << 36;     // refuse to inflate anything to more than 64 GiB

// lz4_flex::block::decompress_size_prepended: u32 LE size of the plain data, then one LZ4 block.
Result<Bytes> lz4_block(const uint8_t* p, size_t n) {
    using R = Result<Bytes>;
    if (n < 4) return R::Err("lz4: no size prefix");
    const size_t want = size_t(p[0]) | size_t(p[1]) << 8 | size_t(p[2]) << 16 | size_t(p[3]) << 24;
    if (want > (n - 4) * 255 + 64) return R::Err("lz4: size prefix larger than this block can expand to");   // a match extends by at most 255 per byte
    Bytes out;
    out.reserve(want);
    size_t i = 4;
    while (i < n) {
        const uint8_t token = p[i++];
        size_t lit = token >> 4;
        if (lit == 15) {
            uint8_t b = 0;
            do { if (i >= n) return R::Err("lz4: truncated literal length"); b = p[i++]; lit += b; } while (b == 255);
        }
        if (n - i < lit || want - out.size() < lit) return R::Err("lz4: literal run past the end");
        out.insert(out.end(), p + i, p + i + lit);
        i += lit;
        if (i == n) break;                                    // the last sequence has no match part
        if (n - i < 2) return R::Err("lz4: truncated match offset");
        const size_t off = size_t(p[i]) | size_t(p[i + 1]) << 8;
        i += 2;
        size_t len = token & 15;
        if (len == 15) {
            uint8_t b = 0;
            do { if (i >= n) return R::Err("lz4: truncated match length"); b = p[i++]; len += b; } while (b == 255);
        }
        len += 4;
        if (off == 0 || off > out.size()) return R::Err("lz4: match offset outside the output");
        if (want - out.size() < len) return R::Err("lz4: output larger than the size prefix");
        size_t src = out.size() - off;
        for (size_t k = 0; k < len; ++k) out.push_back(out[src + k]);   // may overlap its own output
    }
    if (out.size() != want) return R::Err("lz4: plain size " + std::to_string(out.size()) + " differs from the prefix " + std::to_string(want));
    return R::Ok(std::move(out));
}

// flate2's ZlibDecoder: one zlib stream, read to its end.
Result<Bytes> zlib_stream(const uint8_t* p, size_t n) {
    using R = Result<Bytes>;
    z_stream z{};
    if (inflateInit(&z) != Z_OK) return R::Err("zlib: inflateInit failed");
    Bytes out(std::max<size_t>(n * 4, 1 << 16));
    z.next_in = const_cast<Bytef*>(p);
    size_t in_left = n, produced = 0;
    for (;;) {
        if (produced == out.size()) {
            if (out.size() >= kMaxPlain) { inflateEnd(&z); return R::Err("zlib: plain data too large"); }
            out.resize(out.size() * 2);
        }
        if (z.avail_in == 0 && in_left > 0) {                  // next_in advances by itself
            const size_t chunk = std::min<size_t>(in_left, 1u << 30);
            z.avail_in = uInt(chunk);
            in_left -= chunk;
        }
        const size_t out_now = std::min<size_t>(out.size() - produced, 1u << 30);
        z.next_out = out.data() + produced;
        z.avail_out = uInt(out_now);
        const int rc = inflate(&z, Z_NO_FLUSH);
        produced += out_now - z.avail_out;
        if (rc == Z_STREAM_END) break;
        const bool starved = z.avail_in == 0 && in_left == 0 && z.avail_out != 0;
        if (rc == Z_OK && !starved) continue;
        if (rc == Z_BUF_ERROR && z.avail_out == 0) continue;   // output full: grow and go on
        const std::string what = (rc == Z_OK || rc == Z_BUF_ERROR) ? "truncated stream" : (z.msg ? z.msg : "corrupt stream");
        inflateEnd(&z);
        return R::Err("zlib: " + what);
    }
    inflateEnd(&z);
    out.resize(produced);
    return R::Ok(std::move(out));
}

// zstd::decode_all: every frame of the input, streaming (encode_all writes no content size).
// libzstd ships without headers in this image; the three entry points and the two buffer structs
// used here are part of its stable ABI.
struct ZstdIn { const void* src; size_t size, pos; };
struct ZstdOut { void* dst; size_t size, pos; };
struct Zstd {
    void* (*create)() = nullptr;
    size_t (*free_)(void*) = nullptr;
    size_t (*step)(void*, ZstdOut*, ZstdIn*) = nullptr;
    unsigned (*is_error)(size_t) = nullptr;
    const char* (*error_name)(size_t) = nullptr;
    bool ok = false;
    static const Zstd& get() {
        static Zstd z;
        static std::once_flag once;
        std::call_once(once, [] {
            void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("libzstd.so", RTLD_NOW | RTLD_LOCAL);
            if (!h) return;
            z.create = reinterpret_cast<void* (*)()>(dlsym(h, "ZSTD_createDStream"));
            z.free_ = reinterpret_cast<size_t (*)(void*)>(dlsym(h, "ZSTD_freeDStream"));
            z.step = reinterpret_cast<size_t (*)(void*, ZstdOut*, ZstdIn*)>(dlsym(h, "ZSTD_decompressStream"));
            z.is_error = reinterpret_cast<unsigned (*)(size_t)>(dlsym(h, "ZSTD_isError"));
            z.error_name = reinterpret_cast<const char* (*)(size_t)>(dlsym(h, "ZSTD_getErrorName"));
            z.ok = z.create && z.free_ && z.step && z.is_error && z.error_name;
        });
        return z;
    }
};
Result<Bytes> zstd_stream(const uint8_t* p, size_t n) {
    using R = Result<Bytes>;
    const Zstd& z = Zstd::get();
    if (!z.ok) return R::Err("zstd: libzstd.so.1 is not available on this host");
    void* d = z.create();
    if (!d) return R::Err("zstd: ZSTD_createDStream failed");
    Bytes out(std::max<size_t>(n * 4, 1 << 16));
    ZstdIn in{p, n, 0};
    ZstdOut o{out.data(), out.size(), 0};
    size_t hint = n ? 1 : 0;                          // 0 = a frame just ended (decode_all of nothing is nothing)
    while (in.pos < in.size || (hint != 0 && o.pos == o.size)) {
        if (o.pos == o.size) {
            if (out.size() >= kMaxPlain) { z.free_(d); return R::Err("zstd: plain data too large"); }
            out.resize(out.size() * 2);
            o.dst = out.data(); o.size = out.size();
        }
        const size_t before_in = in.pos, before_out = o.pos;
        hint = z.step(d, &o, &in);
        if (z.is_error(hint)) { std::string e = z.error_name(hint); z.free_(d); return R::Err("zstd: " + e); }
        if (in.pos == before_in && o.pos == before_out && o.pos < o.size) break;   // no progress: input exhausted mid-frame
    }
    z.free_(d);
    if (hint != 0) return R::Err("zstd: truncated frame");
    out.resize(o.pos);
    return R::Ok(std::move(out));
}

// snap::read::FrameDecoder: the Snappy framing format (stream identifier, then chunks of at most
// 65536 plain bytes, each with a masked CRC-32C of its plain data) around raw Snappy blocks.
uint32_t crc32c(const uint8_t* p, size_t n) {
    static uint32_t table[256];
    static std::once_flag once;
    std::call_once(once, [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            table[i] = c;
        }
    });
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
bool snappy_raw(const uint8_t* p, size_t n, Bytes& out, std::string& err) {
    size_t i = 0;
    uint64_t want = 0;
    for (int shift = 0;; shift += 7) {
        if (i >= n || shift > 28) { err = "snappy: bad length header"; return false; }
        const uint8_t b = p[i++];
        want |= uint64_t(b & 0x7F) << shift;
        if (!(b & 0x80)) break;
    }
    if (want > 65536) { err = "snappy: chunk larger than 65536 bytes"; return false; }
    const size_t base = out.size();
    while (i < n) {
        const uint8_t tag = p[i++];
        size_t len, off;
        switch (tag & 3) {
            case 0: {
                len = tag >> 2;
                if (len >= 60) {
                    const size_t extra = len - 59;
                    if (n - i < extra) { err = "snappy: truncated literal length"; return false; }
                    len = 0;
                    for (size_t k = 0; k < extra; ++k) len |= size_t(p[i + k]) << (8 * k);
                    i += extra;
                }
                len += 1;
                if (n - i < len || want - (out.size() - base) < len) { err = "snappy: literal past the end"; return false; }
                out.insert(out.end(), p + i, p + i + len);
                i += len;
                continue;
            }
            case 1:
                if (i >= n) { err = "snappy: truncated copy"; return false; }
                len = 4 + ((tag >> 2) & 7);
                off = size_t(tag >> 5) << 8 | p[i++];
                break;
            case 2:
                if (n - i < 2) { err = "snappy: truncated copy"; return false; }
                len = size_t(tag >> 2) + 1;
                off = size_t(p[i]) | size_t(p[i + 1]) << 8;
                i += 2;
                break;
            default:
                if (n - i < 4) { err = "snappy: truncated copy"; return false; }
                len = size_t(tag >> 2) + 1;
                off = size_t(p[i]) | size_t(p[i + 1]) << 8 | size_t(p[i + 2]) << 16 | size_t(p[i + 3]) << 24;
                i += 4;
                break;
        }
        if (off == 0 || off > out.size() - base) { err = "snappy: copy offset outside the chunk"; return false; }
        if (want - (out.size() - base) < len) { err = "snappy: chunk larger than its header says"; return false; }
        const size_t src = out.size() - off;
        for (size_t k = 0; k < len; ++k) out.push_back(out[src + k]);
    }
    if (out.size() - base != want) { err = "snappy: chunk shorter than its header says"; return false; }
    return true;
}
Result<Bytes> snappy_frames(const uint8_t* p, size_t n) {
    using R = Result<Bytes>;
    static const uint8_t kIdent[6] = {'s', 'N', 'a', 'P', 'p', 'Y'};
    Bytes out;
    size_t i = 0;
    bool seen_ident = false;
    while (i < n) {
        if (n - i < 4) return R::Err("snappy: truncated chunk header");
        const uint8_t type = p[i];
        const size_t len = size_t(p[i + 1]) | size_t(p[i + 2]) << 8 | size_t(p[i + 3]) << 16;
        i += 4;
        if (n - i < len) return R::Err("snappy: truncated chunk");
        const uint8_t* c = p + i;
        i += len;
        if (type == 0xFF) {
            if (len != 6 || std::memcmp(c, kIdent, 6) != 0) return R::Err("snappy: bad stream identifier");
            seen_ident = true;
            continue;
        }
        if (!seen_ident) return R::Err("snappy: no stream identifier");
        if (type >= 0x80) continue;                              // skippable chunks and padding
        if (type > 0x01) return R::Err("snappy: reserved unskippable chunk type " + std::to_string(type));
        if (len < 4) return R::Err("snappy: chunk without a checksum");
        const uint32_t masked = uint32_t(c[0]) | uint32_t(c[1]) << 8 | uint32_t(c[2]) << 16 | uint32_t(c[3]) << 24;
        const size_t base = out.size();
        if (type == 0x01) {
            if (len - 4 > 65536) return R::Err("snappy: chunk larger than 65536 bytes");
            out.insert(out.end(), c + 4, c + len);
        } else {
            std::string e;
            if (!snappy_raw(c + 4, len - 4, out, e)) return R::Err(e);
        }
        const uint32_t crc = crc32c(out.data() + base, out.size() - base);
        if (uint32_t(((crc >> 15) | (crc << 17)) + 0xA282EAD8u) != masked) return R::Err("snappy: checksum mismatch");
    }
    return R::Ok(std::move(out));
}

uint64_t le64(const uint8_t* p) {
    uint64_t v = 0;
    for (int i = 7; i >= 0; --i) v = v << 8 | p[i];
    return v;
}

}  // namespace

static Result<std::vector<uint8_t>> uncompress_impl(Compression c, const uint8_t* p, size_t n) {
    switch (c) {
        case Compression::None: return Result<Bytes>::Ok(Bytes(p, p + n));
        case Compression::Zstd: return zstd_stream(p, n);
        case Compression::Lz4: return lz4_block(p, n);
        case Compression::Zlib: return zlib_stream(p, n);
        case Compression::Snappy: return snappy_frames(p, n);
    }
    return Result<Bytes>::Err("unknown compression " + std::to_string(int(c)));
}

// allocation failures (a crafted length can ask for gigabytes) come back as Err like everything else
Result<std::vector<uint8_t>> uncompress(Compression c, const uint8_t* p, size_t n) {
    try { return uncompress_impl(c, p, n); }
    catch (const std::exception& e) { return Result<Bytes>::Err(std::string("uncompress: ") + e.what()); }
}

static Result<Snapshot> decode_snapshot_impl(const uint8_t* p, size_t n, Compression comp, Features f) {
    using R = Result<Snapshot>;
    // router.rs:512-531: [len][relations][len][client_states][len][topics_count][len][relations_count]
    const uint8_t* sec[4];
    size_t sec_len[4], pos = 0;
    static const char* const names[4] = {"relations", "client_states", "topics_count", "relations_count"};
    for (int s = 0; s < 4; ++s) {
        if (n - pos < 8) return R::Err(std::string(names[s]) + ": no length prefix (snapshot is " + std::to_string(n) + " bytes)");
        const uint64_t len = le64(p + pos);
        pos += 8;
        if (len > n - pos) return R::Err(std::string(names[s]) + ": length prefix " + std::to_string(len) + " runs past the snapshot");
        sec[s] = p + pos;
        sec_len[s] = size_t(len);
        pos += size_t(len);
    }
    Snapshot snap;
    Bytes plain[2];
    for (int s = 0; s < 2; ++s) {
        if (comp == Compression::None) continue;
        auto u = uncompress(comp, sec[s], sec_len[s]);
        if (!u.ok()) return R::Err(std::string(names[s]) + ": " + u.error);
        plain[s] = std::move(*u.value);
        sec[s] = plain[s].data();
        sec_len[s] = plain[s].size();
    }
    {   // Vec<(TopicFilter, HashMap<ClientId, (Id, SubscriptionOptions)>)>
        Reader r{sec[0], sec_len[0], 0, {}, names[0]};
        uint64_t n_filters;
        if (!r.length(n_filters, 2)) return R::Err(r.err);
        snap.n_filters = n_filters;
        for (uint64_t i = 0; i < n_filters; ++i) {
            std::string filter;
            uint64_t n_rel;
            if (!r.str(filter) || !r.length(n_rel, 10)) return R::Err(r.err);
            for (uint64_t k = 0; k < n_rel; ++k) {
                Relation rel;
                rel.topic_filter = filter;
                if (!r.str(rel.client_id) || !read_id(r, rel.id) || !read_opts(r, f, rel.opts, rel.limit_subs)) return R::Err(r.err);
                snap.relations.push_back(std::move(rel));
            }
        }
    }
    {   // Vec<(ClientId, ClientStatus)>
        Reader r{sec[1], sec_len[1], 0, {}, names[1]};
        uint64_t n_cs;
        if (!r.length(n_cs, 11)) return R::Err(r.err);
        snap.client_states.reserve(size_t(n_cs));
        for (uint64_t i = 0; i < n_cs; ++i) {
            ClientStatus cs;
            if (!r.str(cs.client_id) || !read_id(r, cs.id) || !r.boolean(cs.online) || !r.boolean(cs.handshaking) || !r.i64(cs.handshak_duration))
                return R::Err(r.err);
            snap.client_states.push_back(std::move(cs));
        }
    }
    for (int s = 2; s < 4; ++s) {
        Reader r{sec[s], sec_len[s], 0, {}, names[s]};
        if (!read_counter(r, s == 2 ? snap.topics_count : snap.relations_count)) return R::Err(r.err);
    }
    return R::Ok(std::move(snap));
}

Result<Snapshot> decode_snapshot(const uint8_t* p, size_t n, Compression comp, Features f) {
    try { return decode_snapshot_impl(p, n, comp, f); }
    catch (const std::exception& e) { return Result<Snapshot>::Err(std::string("decode_snapshot: ") + e.what()); }
}

}  // namespace raft
}  // namespace rmqtt
