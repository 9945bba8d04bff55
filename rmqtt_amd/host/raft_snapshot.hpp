// Reader for the reference's Raft snapshot of the routing table:
//   rmqtt-plugins/rmqtt-cluster-raft/src/router.rs:387-463  (snapshot: what is written)
//   rmqtt-plugins/rmqtt-cluster-raft/src/router.rs:466-580  (restore: how it is read back)
//
// Layout (router.rs:414-450): four sections, each prefixed by its length as a little-endian usize
// (8 bytes): [relations][client_states][topics_count][relations_count].  The first two are
// postcard (1.x wire format) of
//     Vec<(TopicFilter, HashMap<ClientId, (Id, SubscriptionOptions)>)>      (router.rs:540)
//     Vec<(ClientId, ClientStatus)>                                          (router.rs:541, :39-45)
// run through the configured compression (config.rs:415-420: zstd | lz4 (lz4_flex block, size
// prepended) | zlib | snappy (frame format) | none); the two counters are postcard of
// rmqtt-utils Counter(isize, isize, StatsMergeMode) (counter.rs:39, :337), never compressed.
//
// The struct layouts follow the serde derives of rmqtt/src/types.rs: _Id :1899-1911,
// SubscriptionOptions :607-610, SubOptionsV3 :769-779, SubOptionsV5 :803-821 (qos and
// retain_handling go out as one u8 each, :717-731, :880-890).  `shared_group` and `limit_subs` exist
// only under the `shared-subscription` / `limit-subscription` cargo features; the broker binary
// builds rmqtt with "full" (rmqtt-bin/Cargo.toml:25), so both default to on here.
//
// PARITY NOTE: there is no Rust toolchain in this image, so no snapshot written by the reference
// itself could be used as a fixture.  The reader is pinned on postcard's published wire format and
// on serde's documented impls for the std types involved (SocketAddr as a variant index + (octets,
// port), NonZeroU32 as u32, atomics as their integer); oracle/raft_snapshot.py restates the writer
// side and is the test vector generator (tests/test_raft_snapshot.py).
#pragma once
#include <cstdint>
#include <optional>
#include <string>
#include <vector>

#include "gpu_router.hpp"

namespace rmqtt {
namespace raft {

enum class Compression : int { None = 0, Zstd = 1, Lz4 = 2, Zlib = 3, Snappy = 4 };   // config.rs:415-420 (+ None = Option::None)

struct Features {
    bool shared_subscription = true;   // types.rs:775-776, :814-815
    bool limit_subscription = true;    // types.rs:777-778, :816-817
};

struct Relation {                     // one (filter, client) entry of AllRelationsMap (types.rs:476)
    TopicFilter topic_filter;
    ClientId client_id;               // the map key (router.rs:447: id.client_id)
    Id id;
    SubscriptionOptions opts;
    std::optional<uint64_t> limit_subs;
};
struct ClientStatus {                 // router.rs:39-45
    ClientId client_id;
    Id id;
    bool online = false, handshaking = false;
    int64_t handshak_duration = 0;
};
struct CounterState { int64_t count = 0, max = 0; uint32_t merge_mode = 0; };   // counter.rs:39

struct Snapshot {
    uint64_t n_filters = 0;           // entries of the outer Vec
    std::vector<Relation> relations;  // flattened, in wire order
    std::vector<ClientStatus> client_states;
    CounterState topics_count, relations_count;
};

// Never throws; Err carries what was wrong and where (section + byte offset).
Result<std::vector<uint8_t>> uncompress(Compression c, const uint8_t* p, size_t n);
Result<Snapshot> decode_snapshot(const uint8_t* p, size_t n, Compression c, Features f = Features{});

}  // namespace raft
}  // namespace rmqtt
