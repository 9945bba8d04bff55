// Host-side topic / filter tokenisation rules of the product path.
//
// Follows the behaviour of rmqtt's parser (rmqtt/src/topic.rs:357-394 Level::from_str /
// Topic::from_str, 231-243 Topic::is_valid): split on '/', classify each segment, reject a
// segment that merely *contains* '+' or '#', '#' only as the last level, a '$'-prefixed
// ("metadata") level only at index 0.  The same rules apply to publish topics and filters.
#pragma once
#include <cstdint>
#include <string_view>

namespace rgr {

enum class LevelKind : uint8_t { Normal, Metadata, Blank, Plus, Hash, Bad };

inline LevelKind classify_level(std::string_view s) {
    if (s.empty()) return LevelKind::Blank;
    if (s.size() == 1) {
        if (s[0] == '+') return LevelKind::Plus;
        if (s[0] == '#') return LevelKind::Hash;
    }
    for (char c : s)
        if (c == '+' || c == '#') return LevelKind::Bad;
    return s[0] == '$' ? LevelKind::Metadata : LevelKind::Normal;
}

// Calls fn(index, segment, kind) for every level; returns the level count, or -1 if the
// topic is invalid (in which case fn may have been called for a prefix).
template <class Fn> inline int64_t for_each_level(std::string_view s, Fn&& fn) {
    int64_t idx = 0;
    size_t start = 0;
    bool hash_seen = false;
    for (;;) {
        size_t pos = s.find('/', start);
        std::string_view seg = s.substr(start, pos == std::string_view::npos ? std::string_view::npos : pos - start);
        if (hash_seen) return -1;                     // '#' was not the last level
        LevelKind k = classify_level(seg);
        if (k == LevelKind::Bad) return -1;
        if (k == LevelKind::Metadata && idx != 0) return -1;
        if (k == LevelKind::Hash) hash_seen = true;
        fn(idx, seg, k);
        ++idx;
        if (pos == std::string_view::npos) break;
        start = pos + 1;
    }
    return idx;
}

}  // namespace rgr
