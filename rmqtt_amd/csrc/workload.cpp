// Deterministic synthetic MQTT workload generator G(seed)  (SURVEY.md §8(d)).
//
// Shared by the oracle, the parity tests and bench.py so that every leg sees
// byte-identical inputs.  Pure host C++, no GPU, no dependency on the product
// library.  PRNG: splitmix64(seed) -> xoshiro256**.
//
//   vocab_l      = min(16 * 4^l, 65536)           token string "l{l}x{k}"
//   rank k       ~ Zipf(s = 1.1) over vocab_l      (inverse-CDF table)
//   depth L      = clamp(round(N(8, 2^2)), 1, 16)  (fixed 4 when fixed_depth != 0)
//   filter       = L tokens; each level -> "+" with prob p_plus; with prob
//                  p_hash keep the first d ~ U[1, L] levels and append "#";
//                  with prob p_sys the first level is replaced by "$SYS"
//   client       = j ~ Zipf(s = 1.0) over n_clients; (filter, client) pairs
//                  that repeat are re-drawn; qos ~ U{0,1,2}
//   publish topic= L tokens from the same per-level distributions, no
//                  wildcards; p_sys_topic start with "$SYS"; p_blank get a
//                  leading or a trailing blank level
//
// C ABI (ctypes-friendly): all outputs are malloc'ed by the library and
// released with wl_free().
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_set>
#include <vector>
#include <algorithm>

namespace {

struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t& x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed) {
        for (auto& w : s) w = splitmix(seed);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9;
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return double(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint64_t below(uint64_t n) { return uint64_t(uniform() * double(n)); }
};

struct Zipf {
    std::vector<double> cdf;
    Zipf() = default;
    Zipf(uint32_t n, double s) : cdf(n) {
        double acc = 0;
        for (uint32_t i = 0; i < n; ++i) { acc += std::pow(double(i) + 1.0, -s); cdf[i] = acc; }
        for (auto& c : cdf) c /= acc;
    }
    uint32_t sample(Rng& r) const {
        const double u = r.uniform();
        size_t k = std::upper_bound(cdf.begin(), cdf.end(), u) - cdf.begin();
        if (k >= cdf.size()) k = cdf.size() - 1;
        return uint32_t(k);
    }
};

constexpr int kMaxDepth = 16;

struct LevelVocab {
    Zipf z[kMaxDepth];
    LevelVocab() {
        for (int l = 0; l < kMaxDepth; ++l) {
            uint64_t v = 16;
            for (int i = 0; i < l && v < 65536; ++i) v *= 4;
            if (v > 65536) v = 65536;
            z[l] = Zipf(uint32_t(v), 1.1);
        }
    }
};

const LevelVocab& vocab() { static LevelVocab v; return v; }

int draw_depth(Rng& r, int fixed_depth) {
    if (fixed_depth > 0) return fixed_depth;
    const double u1 = 1.0 - r.uniform(), u2 = r.uniform();
    const double z = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    long L = std::lround(8.0 + 2.0 * z);
    if (L < 1) L = 1;
    if (L > kMaxDepth) L = kMaxDepth;
    return int(L);
}

void append_token(std::string& out, int level, uint32_t k) {
    char buf[32];
    int n = std::snprintf(buf, sizeof buf, "l%dx%u", level, k);
    out.append(buf, size_t(n));
}

struct Blob {
    std::string bytes;
    std::vector<uint64_t> offsets{0};
    void push(const std::string& s) { bytes += s; offsets.push_back(bytes.size()); }
};

template <class T> T* dup(const std::vector<T>& v) {
    T* p = static_cast<T*>(std::malloc(std::max<size_t>(1, v.size()) * sizeof(T)));
    if (!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}
char* dup_bytes(const std::string& s) {
    char* p = static_cast<char*>(std::malloc(std::max<size_t>(1, s.size())));
    if (!s.empty()) std::memcpy(p, s.data(), s.size());
    return p;
}

}  // namespace

extern "C" {

struct wl_params {
    uint64_t seed;
    uint64_t n;            // items to generate
    double p_plus;         // per-level '+' probability (filters)
    double p_hash;         // probability the filter ends in '#'
    double p_sys;          // probability first level is "$SYS"
    double p_blank;        // probability of a leading/trailing blank level (topics)
    uint64_t n_clients;    // client population (filters); 0 -> n/4 (min 1)
    int32_t fixed_depth;   // >0: every item has exactly this many levels
    int32_t force_wildcard;  // filters: re-draw until the filter has '+' or '#'
    int32_t distinct;        // topics: re-draw repeated topic strings
    int32_t reserved;
};

// Subscriptions: filter strings + client index + qos.
int wl_gen_subs(const wl_params* p, char** blob, uint64_t** offsets, uint32_t** client, uint8_t** qos) {
    Rng r(p->seed);
    const auto& V = vocab();
    const uint64_t nc = p->n_clients ? p->n_clients : std::max<uint64_t>(1, p->n / 4);
    Zipf zc(uint32_t(nc), 1.0);
    Blob b;
    b.bytes.reserve(p->n * 48);
    b.offsets.reserve(p->n + 1);
    std::vector<uint32_t> cl; cl.reserve(p->n);
    std::vector<uint8_t> q; q.reserve(p->n);
    std::unordered_set<uint64_t> seen;
    seen.reserve(p->n * 2);
    std::string f;
    uint64_t made = 0;
    while (made < p->n) {
        f.clear();
        const int L = draw_depth(r, p->fixed_depth);
        int keep = L;
        bool hash = false;
        if (p->p_hash > 0 && r.uniform() < p->p_hash) { hash = true; keep = 1 + int(r.below(uint64_t(L))); }
        bool wild = hash;
        const bool sys = p->p_sys > 0 && r.uniform() < p->p_sys;
        for (int l = 0; l < keep; ++l) {
            if (l) f.push_back('/');
            const uint32_t k = V.z[l].sample(r);      // always consume the draw
            const bool plus = p->p_plus > 0 && r.uniform() < p->p_plus;
            if (l == 0 && sys) f += "$SYS";
            else if (plus) { f.push_back('+'); wild = true; }
            else append_token(f, l, k);
        }
        if (hash) f += "/#";
        const uint32_t c = zc.sample(r);
        const uint8_t qv = uint8_t(r.below(3));
        if (p->force_wildcard && !wild) continue;
        // (filter, client) uniqueness: 64-bit FNV-1a of the filter mixed with the client
        uint64_t h = 1469598103934665603ull;
        for (unsigned char ch : f) { h ^= ch; h *= 1099511628211ull; }
        h ^= (uint64_t(c) + 0x9E3779B97F4A7C15ull) * 0xD6E8FEB86659FD93ull;
        if (!seen.insert(h).second) continue;
        b.push(f); cl.push_back(c); q.push_back(qv);
        ++made;
    }
    *blob = dup_bytes(b.bytes); *offsets = dup(b.offsets); *client = dup(cl); *qos = dup(q);
    return 0;
}

// Publish topics (no wildcards).
int wl_gen_topics(const wl_params* p, char** blob, uint64_t** offsets) {
    Rng r(p->seed);
    const auto& V = vocab();
    Blob b;
    b.bytes.reserve(p->n * 56);
    b.offsets.reserve(p->n + 1);
    std::string t;
    std::unordered_set<uint64_t> seen;
    if (p->distinct) seen.reserve(p->n * 2);
    for (uint64_t i = 0; i < p->n;) {
        t.clear();
        const int L = draw_depth(r, p->fixed_depth);
        const bool sys = p->p_sys > 0 && r.uniform() < p->p_sys;
        int blank = 0;   // 1 leading, 2 trailing
        if (p->p_blank > 0 && r.uniform() < p->p_blank) blank = 1 + int(r.below(2));
        if (blank == 1) t.push_back('/');
        for (int l = 0; l < L; ++l) {
            if (l) t.push_back('/');
            const uint32_t k = V.z[l].sample(r);
            if (l == 0 && sys) t += "$SYS"; else append_token(t, l, k);
        }
        if (blank == 2) t.push_back('/');
        if (p->distinct) {
            uint64_t h = 1469598103934665603ull;
            for (unsigned char ch : t) { h ^= ch; h *= 1099511628211ull; }
            if (!seen.insert(h).second) continue;
        }
        b.push(t);
        ++i;
    }
    *blob = dup_bytes(b.bytes); *offsets = dup(b.offsets);
    return 0;
}

// Gather strings idx[0..m) of (blob, offsets) into a new (blob, offsets).
int wl_take(const char* blob, const uint64_t* offsets, const uint64_t* idx, uint64_t m, char** out_blob, uint64_t** out_offsets) {
    std::vector<uint64_t> off(m + 1, 0);
    for (uint64_t i = 0; i < m; ++i) off[i + 1] = off[i] + (offsets[idx[i] + 1] - offsets[idx[i]]);
    char* b = static_cast<char*>(std::malloc(std::max<uint64_t>(1, off[m])));
    for (uint64_t i = 0; i < m; ++i) std::memcpy(b + off[i], blob + offsets[idx[i]], off[i + 1] - off[i]);
    *out_blob = b; *out_offsets = dup(off);
    return 0;
}

void wl_free(void* p) { std::free(p); }

}  // extern "C"
