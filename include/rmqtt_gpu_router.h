/*
 * rmqtt_gpu_router.h — C ABI of the MI355X-native publish-time topic matcher.
 *
 * Drop-in boundary for rmqtt's `Router::matches` hot path (and the RetainTree twin).
 * The reference has no C ABI (it is in-process Rust); each entry point below names the
 * reference interface it replaces.  A Rust `GpuRouter` plugin wraps `DefaultRouter`,
 * delegates every `Router` method to it except `matches`, and mirrors `add`/`remove`
 * into this table (INTEGRATION.md shows the binding).
 *
 *   reference interface                                         replaced by
 *   ----------------------------------------------------------  --------------------------
 *   DefaultRouter::new            rmqtt/src/router.rs:131-139    rgr_create
 *   Router::add  (trie insert)    rmqtt/src/router.rs:434-453    rgr_filter_add + rgr_sub_add
 *                                 rmqtt/src/trie.rs:113-126
 *   Router::remove (trie prune)   rmqtt/src/router.rs:456-496    rgr_sub_remove + rgr_filter_remove
 *                                 rmqtt/src/trie.rs:129-149
 *   ClusterRouter restore loop    rmqtt-plugins/rmqtt-cluster-raft/src/router.rs:557-566
 *                                                                rgr_subscribe_bulk
 *   Router::matches / _matches    rmqtt/src/router.rs:499-501,   rgr_match_batch,
 *     (Topic::from_str, TopicTree   174-265; topic.rs:379-394;   rgr_batch_* (windowed,
 *      ::matches, relations walk)   trie.rs:157-159, 301-409      device-resident form)
 *   _has_matches / _get_routes    rmqtt/src/router.rs:151-170    rgr_match_filters
 *   RetainTree insert/remove      rmqtt/src/retain.rs:373-413    rgr_retain_topic_add/_remove
 *   RetainTree::matches           rmqtt/src/retain.rs:450-526    rgr_retain_match_batch
 *     (DefaultRetainStorage::get_message retain.rs:250-267,
 *      rmqtt-retainer storage.rs:604-611)
 *
 * Conventions
 *   - plain C99, no STL / torch types; every function returns an int32_t status
 *     (RGR_OK == 0, negative = error) and never throws across the boundary;
 *     rgr_last_error() returns a thread-local description of the last failure.
 *   - strings are (ptr,len) byte ranges borrowed for the call only; batches are a
 *     byte blob plus n+1 uint64 offsets.
 *   - ownership: the caller (Rust) owns the subscription table objects; it assigns a
 *     dense `sub_id` per (filter, client) relation.  `filter_id`s are assigned by the
 *     library (one per distinct filter string).  The device side only ever sees ids.
 *   - threading: handles are thread-safe.  Mutations are serialised internally and
 *     become visible to matches at rgr_commit() (immutable epoch snapshot; matches in
 *     flight keep the epoch they started with).  Match calls are re-entrant.
 *   - result order (bit-exact contract): per topic the matched
 *     filters appear in exactly TopicTree::matches' iteration order; within a filter
 *     subscribers are ascending by sub_id.
 *   - there is NO CPU fallback: if no HIP device is usable rgr_create fails with
 *     RGR_EDEVICE.
 */
#ifndef RMQTT_GPU_ROUTER_H
#define RMQTT_GPU_ROUTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------ */
enum {
    RGR_OK = 0,
    RGR_EOF = 1,              /* rgr_batch_next_window: no more windows               */
    RGR_EINVAL = -1,          /* bad argument                                         */
    RGR_EINVAL_TOPIC = -2,    /* topic / filter rejected by the parser (topic.rs Err) */
    RGR_ENOMEM = -3,
    RGR_EDEVICE = -4,         /* HIP error / no usable device                         */
    RGR_ECAPACITY = -5,       /* a fixed capacity (window, arena) was exceeded        */
    RGR_ENOENT = -6,          /* unknown filter_id / sub_id                           */
    RGR_ESTATE = -7           /* call sequence error                                  */
};

/* per-topic status values in result/status arrays */
enum {
    RGR_TOPIC_OK = 0,
    RGR_TOPIC_INVALID = -2,   /* Topic::from_str would return Err => no subscribers
                                 (rmqtt/src/shared.rs:774-777)                       */
    RGR_PACKET_MALFORMED = -8 /* rgr_batch_create_from_publish: the codec would reject the packet
                                 (DecodeError; rgr_publish_info.error says which)     */
};

/* subscription flag bits carried next to qos (opaque to the matcher) */
enum {
    RGR_SUB_V5 = 1u << 0,         /* SubscriptionOptions::V5 (types.rs:607-610)       */
    RGR_SUB_NO_LOCAL = 1u << 1,   /* v5 No Local (router.rs:196-201, applied by host) */
    RGR_SUB_SHARED = 1u << 2,     /* member of a $share group (host post-filter)      */
    RGR_SUB_RAP = 1u << 3         /* v5 Retain As Published (shared.rs:889-897)       */
};
/* Flag bits 4-7 are a caller-defined TABLE id, opaque to the matcher: the
 * reference keeps further `TopicTree`s that are matched against the same publish topic — the
 * egress bridges (rmqtt-bridge-egress-mqtt/src/bridge.rs:103,202: `topics.read().matches(&topic)`
 * per publish), ACL rule topics (rmqtt-acl/src/config.rs:170,330).  Their entries can live in
 * the same handle as "subscriptions" of table 1, 2, ...: one device pass returns the hits of
 * every table, still in TopicTree::matches filter order, and the consumer demultiplexes on
 * RGR_SUB_TABLE(flags).  Table 0 = the router's relations. */
#define RGR_SUB_TABLE_SHIFT 4
#define RGR_SUB_TABLE_MASK 0xF0u
#define RGR_SUB_TABLE(flags) (((flags) & RGR_SUB_TABLE_MASK) >> RGR_SUB_TABLE_SHIFT)

/* Delivery word: what rgr_tuple.qos_flags holds for a batch that carries publish
 * attributes (rgr_batch_set_publish_attrs / rgr_match_batch_deliver) — the per-hit part of
 * DefaultRouter::_matches + forwards_to done on the device:
 *   bits 0-1   delivery qos = min(publish qos, subscription qos)   (shared.rs:902)
 *   bit  2     deliver with retain=1: v5 Retain-As-Published && publish.retain (shared.rs:889-897)
 *   bit  3     dropped by v5 No Local: subscriber Id == publisher Id (router.rs:196-201)
 *   bit  4     later v5 hit of a client already collected for this topic: only its
 *              subscription identifier is appended to the first one (types.rs:526-534)
 *   bits 8-15  RGR_SUB_* flags, bits 16-31 node_idx — as without publish attributes     */
enum {
    RGR_HIT_QOS_MASK = 3u,
    RGR_HIT_RETAIN = 1u << 2,
    RGR_HIT_NO_LOCAL = 1u << 3,
    RGR_HIT_V5_DUP = 1u << 4
};
#define RGR_ID_NONE 0xFFFFFFFFu

typedef struct rgr_handle rgr_handle;
typedef struct rgr_batch rgr_batch;

typedef struct rgr_config {
    int32_t device;             /* HIP device ordinal                                   */
    uint32_t slot_cap;          /* matched-filter slots per topic before the overflow
                                   arena is used (0 = default 64)                       */
    uint64_t window_hits;       /* capacity of one expansion window in hits.  0 = default: 2^30 (12 GiB of
                                   tuples) for device-resident passes, 2^28 for passes that stage windows on the
                                   host; device-resident passes of the delivery stage: 2^28 in 8-byte hits (RGR_FORMAT_DELIVER8), 2^27 as
                                   tuples (measured, rounds 5 and 6).
                                   A non-zero value is used as given.                                       */
    uint32_t chunk_topics;      /* topics walked per pass (0 = default 2^21)            */
    uint32_t host_threads;      /* tokeniser threads (0 = hardware concurrency)         */
    uint32_t collect_walk_stats;/* nonzero: count visited trie nodes in the walk kernel */
    uint32_t host_tokenize;     /* nonzero: tokenise topics on the host (threads above) instead
                                   of with the device tokeniser kernels (two-tier retain
                                   batches are always tokenised on the device)           */
    uint32_t retain_delta_max;  /* retained-topic twin in two tiers: 0 = off —
                                   one compiled table, every structural change recompiles it;
                                   N > 0 = topics added since the last full compile live in a
                                   small DELTA table (recompiled alone), removed ones get a dead
                                   bit, and the tiers are merged when the delta exceeds N topics
                                   or a quarter of the base is dead                          */
    uint32_t _reserved0;
} rgr_config;

/* One emitted hit: exactly the (topic_idx, subscriber_id, qos) tuple of BASELINE.json. */
typedef struct rgr_tuple {
    uint32_t topic_idx;         /* index of the publish topic inside the batch          */
    uint32_t sub_id;            /* caller-assigned relation id                          */
    uint32_t qos_flags;         /* bits 0-7 qos, bits 8-15 RGR_SUB_* flags, bits 16-31
                                   node_idx (rgr_sub_add_ex); a delivery word (above)
                                   when the batch carries publish attributes             */
} rgr_tuple;

/* One hit of a delivery pass in RGR_FORMAT_DELIVER8: what forwards_to (shared.rs:876-963) needs per recipient — which relation, and the
 * delivery word (RGR_HIT_*: qos', retain, No Local drop, v5 duplicate; sub flags; node index).  The publish it belongs to is the topic whose
 * CSR range [d_hit_offsets[i], d_hit_offsets[i+1]) holds the position. */
typedef struct rgr_hit8 {
    uint32_t sub_id;
    uint32_t word;
} rgr_hit8;

/* Per-publish attributes of a batch (one per topic): who published it and with what
 * qos / retain bit — `from.id`, `publish.qos`, `publish.retain` of shared.rs:772,880-902. */
typedef struct rgr_publish_attr {
    uint32_t from_id;           /* owner_id (rgr_sub_add_ex) of the publisher's Id, or
                                   RGR_ID_NONE when it holds no subscription            */
    uint32_t qos_retain;        /* bits 0-1 publish qos, bit 2 publish retain            */
} rgr_publish_attr;

/* Host-side result of rgr_match_batch (arrays owned by the library until
 * rgr_result_free). */
typedef struct rgr_result {
    uint32_t n_topics;
    uint64_t n_hits;
    int32_t* status;            /* [n_topics] RGR_TOPIC_*                               */
    uint64_t* hit_offsets;      /* [n_topics+1] CSR offsets into tuples                 */
    rgr_tuple* tuples;          /* [n_hits] topic-major, filter order, sub_id ascending */
    void* _owner;
} rgr_result;

/* Node directory of a delivery result whose tuples were partitioned by node on the device (rgr_match_batch_deliver_grouped):
 * SubRelationsMap is keyed by node (types.rs:486-497; router.rs:258-261), so per publish the host wants one contiguous slice of
 * delivery tuples per node.  Inside every topic the tuples are ordered by node index (ascending; stable: TopicTree::matches filter
 * order and sub-id order are kept inside a node).  Topic t owns groups [group_offsets[t], group_offsets[t+1]); group g holds the
 * tuples [group_begin[g], group_begin[g+1]) of rgr_result.tuples, all of node index group_node[g] (= delivery word >> 16). */
typedef struct rgr_node_groups {
    uint64_t n_groups;
    uint64_t* group_offsets;    /* [n_topics+1]                                         */
    uint32_t* group_node;       /* [n_groups]                                           */
    uint64_t* group_begin;      /* [n_groups+1]                                         */
} rgr_node_groups;

/* Matched-filter view (TopicTree::matches items without the relation expansion). */
typedef struct rgr_filters_result {
    uint32_t n_topics;
    uint64_t n_pairs;
    int32_t* status;            /* [n_topics]                                           */
    uint64_t* pair_offsets;     /* [n_topics+1]                                         */
    uint32_t* filter_ids;       /* [n_pairs] in iteration order (duplicates preserved)  */
    void* _owner;
} rgr_filters_result;

/* One expansion window of a device-resident batch. Pointers are DEVICE pointers valid
 * until the next rgr_batch_next_window / rgr_batch_begin on the same batch.
 * Hits of topic i (topic_begin <= i < topic_end) are
 *   d_tuples[d_hit_offsets[i - topic_begin] - offsets_bias ..
 *            d_hit_offsets[i - topic_begin + 1] - offsets_bias). */
typedef struct rgr_window {
    uint32_t topic_begin, topic_end;  /* topics [begin,end) of the batch                */
    uint64_t n_hits;
    uint64_t hit_base;                /* hits emitted by earlier windows of this pass   */
    const rgr_tuple* d_tuples;        /* [n_hits]; NULL in the compact formats          */
    const uint64_t* d_hit_offsets;    /* [topic_end-topic_begin+1]                      */
    uint64_t offsets_bias;
    /* compact result formats (rgr_batch_set_format), NULL in RGR_FORMAT_TUPLE */
    const uint32_t* d_sub_ids;        /* [n_hits] sub_id (RGR_FORMAT_SOA) or sub_id | qos << 30 (RGR_FORMAT_PACKED) */
    const uint8_t* d_qos;             /* [n_hits] RGR_FORMAT_SOA only: bits 0-1 qos, bits 2-7 = RGR_SUB_* flag bits 0-5 */
    /* RGR_FORMAT_RUNS: the window's hit list as runs, nothing materialised per hit.  Run r covers positions
     * [d_run_off[r] - offsets_bias, d_run_off[r + 1] - offsets_bias) of the window; its k-th hit is the subscriber entry
     * d_subs[d_run_src[r] + k] = { sub_id, qos | flags << 8 | node_idx << 16 } of topic d_run_topic[r].  Runs are in
     * hit order (topic-major, TopicTree::matches filter order); d_subs stays valid for the pass (the epoch is pinned). */
    uint64_t n_runs;
    const uint32_t* d_run_src;        /* [n_runs]                                        */
    const uint32_t* d_run_topic;      /* [n_runs] topic index (or rgr_batch_set_topic_ids' id) */
    const uint64_t* d_run_off;        /* [n_runs + 1]                                    */
    const uint64_t* d_subs;           /* the epoch's subscriber entries, 8 bytes each    */
    const uint8_t* d_ids24;           /* [3 * n_hits] RGR_FORMAT_IDS24: sub ids as 3 little-endian bytes each; NULL otherwise */
    const rgr_hit8* d_hits8;          /* [n_hits] RGR_FORMAT_DELIVER8 (delivery passes): {sub_id, delivery word}; NULL otherwise */
    const uint32_t* d_topic_order;    /* batches in walk order (rgr_batch_set_order): [topic_end - topic_begin] the BATCH index of the window's k-th
                                         topic — topic_begin / topic_end then count walk positions, d_hit_offsets[k] belongs to batch topic
                                         d_topic_order[k]; tuples name the batch index as ever.  NULL: k-th topic = topic_begin + k */
} rgr_window;

/* Result format of a device-resident batch.  The 12-byte tuple is BASELINE.json's (topic_idx, subscriber_id,
 * qos) as written; its topic_idx column is redundant with d_hit_offsets, and at config-3 fan-out (14.8 k hits
 * per publish) the tuple bytes ARE the cost of a publish.  The compact formats (the SoA result of SURVEY.md
 * §8(b)) keep the same hits in the same order and leave the topic to the CSR offsets. */
enum {
    RGR_FORMAT_TUPLE = 0,             /* rgr_tuple[n_hits], 12 B/hit (default)                                   */
    RGR_FORMAT_SOA = 1,               /* d_sub_ids u32[n_hits] + d_qos u8[n_hits], 5 B/hit                        */
    RGR_FORMAT_PACKED = 2,            /* d_sub_ids u32[n_hits] = sub_id | qos << 30, 4 B/hit; needs sub ids < 2^30
                                         (rgr_batch_begin fails with RGR_ECAPACITY otherwise)                  */
    RGR_FORMAT_RUNS = 3,              /* run descriptors only (d_run_*): a hit list is the concatenation of subscriber
                                         runs that already sit in HBM, so a device-side consumer (a fan-out kernel) can
                                         read them in place — 16 B per (topic, matched filter) instead of bytes per hit */
    RGR_FORMAT_DELIVER8 = 5,          /* delivery passes only (rgr_batch_set_publish_attrs first): d_hits8 rgr_hit8[n_hits] = {sub_id, delivery word},
                                         8 B/hit — the 12-byte tuple without its topic column, which d_hit_offsets implies.  The delivery stage's
                                         cost is its stores (1.78 TB per 10 M publishes at config 3 as tuples): a third fewer bytes, same hits, same
                                         order, same words (v5 dedup included). */
    RGR_FORMAT_IDS24 = 4              /* d_ids24 u8[3 * n_hits]: the sub id of every hit as 3 little-endian bytes, 3 B/hit; needs
                                         sub ids < 2^24 (rgr_batch_begin fails with RGR_ECAPACITY otherwise).  The qos is not
                                         carried: it is a property of the subscription, which the consumer indexes by sub id.
                                         At config-3 fan-out the bytes per hit ARE the pass: 592 GB per 10 M publishes at 4 B/hit
                                         cannot leave the chip in under 94 ms at its ~6.3 TB/s store ceiling; 444 GB can. */
};

typedef struct rgr_stats {
    /* table */
    uint64_t n_filters, n_subs, n_nodes, n_edge_slots, n_tokens, epoch;
    uint64_t table_bytes_device;
    /* cumulative since create / rgr_stats_reset */
    uint64_t topics, invalid_topics, levels, pairs, hits, visited_nodes, overflow_topics;
    uint64_t walk_launches, expand_launches;
    double walk_ms, scan_ms, expand_ms, tokenize_ms, h2d_ms, d2h_ms;   /* HIP-event / wall */
    /* algorithmic bytes of the work counted above */
    uint64_t alg_bytes_walk, alg_bytes_expand;
    /* rgr_commit: epochs published by a full image upload vs by patching the delta */
    uint64_t commits_full, commits_delta;
    /* delivery stage (batches with publish attributes): v5 per-client dedup */
    uint64_t dedup_candidates, dedup_launches;
    double dedup_ms;
    /* retained-topic twin: id of the current retain epoch (0 = none; unchanged by a
     * rgr_retain_commit that found nothing to do) and its topic count */
    uint64_t retain_epoch, retain_topics;
    /* two-tier mode: topics in the delta tier, dead base entries, full compiles so far */
    uint64_t retain_delta_topics, retain_dead, retain_merges;
} rgr_stats;

/* ---- lifecycle ----------------------------------------------------------------- */
int32_t rgr_create(const rgr_config* cfg, rgr_handle** out);
void rgr_destroy(rgr_handle* h);
const char* rgr_last_error(void);
const char* rgr_version(void);

/* ---- subscription table (host-owned, mirrored to HBM at commit) ------------------- */
/* Insert a filter path into the trie (idempotent): *filter_id receives the id of the
 * (new or existing) filter.  RGR_EINVAL_TOPIC if Topic::from_str would fail. */
int32_t rgr_filter_add(rgr_handle* h, const char* filter, uint32_t len, uint32_t* filter_id);
/* Look up without inserting: RGR_ENOENT if absent. */
int32_t rgr_filter_find(rgr_handle* h, const char* filter, uint32_t len, uint32_t* filter_id);
/* Remove the filter (must have no subscriptions left) and prune empty trie nodes. */
int32_t rgr_filter_remove(rgr_handle* h, uint32_t filter_id);
int32_t rgr_sub_add(rgr_handle* h, uint32_t filter_id, uint32_t sub_id, uint8_t qos, uint8_t flags);
/* rgr_sub_add plus what the delivery stage needs to know about the subscriber: its node
 * (dense index of Id::node_id, returned in bits 16-31 of every tuple so the consumer can group
 * by node like SubscriptioRelationsCollectorMap, router.rs:176), a dense id of its `Id` object
 * (No Local compares whole Ids, router.rs:198) and a dense id of its ClientId (the v5
 * collector is keyed by client, types.rs:524). */
int32_t rgr_sub_add_ex(rgr_handle* h, uint32_t filter_id, uint32_t sub_id, uint8_t qos, uint8_t flags,
                       uint16_t node_idx, uint32_t owner_id, uint32_t client_idx);
/* Bulk form for subscriptions added with rgr_subscribe_bulk: owner / client ids per sub_id. */
int32_t rgr_sub_attrs_bulk(rgr_handle* h, const uint32_t* sub_ids, const uint32_t* owner_ids,
                           const uint32_t* client_idx, uint64_t n);
int32_t rgr_sub_remove(rgr_handle* h, uint32_t filter_id, uint32_t sub_id);
/* Restore/bulk path: for i in [0,n): filter_add(filter_i) + sub_add(fid, sub_ids ?
 * sub_ids[i] : i, qos[i], flags ? flags[i] : 0).  filter_ids_out (optional, [n]).
 * Returns RGR_OK and stores the number of rejected filters in *n_rejected. */
int32_t rgr_subscribe_bulk(rgr_handle* h, const uint8_t* blob, const uint64_t* offsets, uint64_t n,
                           const uint32_t* sub_ids, const uint8_t* qos, const uint8_t* flags,
                           uint32_t* filter_ids_out, uint64_t* n_rejected);
/* Snapshot file of the compiled host table (dictionary, trie, edge table, subscriber runs,
 * delivery attributes; flat arrays + checksum): a cold start becomes one read + one full
 * upload instead of re-inserting every subscription as the reference's restore does
 * (rmqtt-cluster-raft/src/router.rs:557-566).  filter_id / sub_id values are preserved.
 * load replaces the host table (nothing changes on a failed load) and must be followed by
 * rgr_commit; batches created earlier stay usable (they are re-tokenised). */
int32_t rgr_snapshot_save(rgr_handle* h, const char* path);
int32_t rgr_snapshot_load(rgr_handle* h, const char* path);
/* Publish the current host table as a new immutable device epoch. */
int32_t rgr_commit(rgr_handle* h);

/* ---- matching, host buffers in / host buffers out ------------------------------------ */
int32_t rgr_match_batch(rgr_handle* h, const uint8_t* topics_blob, const uint64_t* topic_offsets, uint32_t n,
                        rgr_result* out);
/* Same, with the delivery stage: tuples carry delivery words (RGR_HIT_*). attrs: [n]. */
int32_t rgr_match_batch_deliver(rgr_handle* h, const uint8_t* topics_blob, const uint64_t* topic_offsets, uint32_t n,
                                const rgr_publish_attr* attrs, rgr_result* out);
/* rgr_match_batch_deliver with the per-publish grouping by node done on the device ("grouping by NodeId ... as a
 * second kernel"): a stable radix partition of every topic's tuples by the node index of the delivery word, plus the directory.
 * `groups` points into memory owned by `out` (released by rgr_result_free(out)). */
int32_t rgr_match_batch_deliver_grouped(rgr_handle* h, const uint8_t* topics_blob, const uint64_t* topic_offsets, uint32_t n,
                                        const rgr_publish_attr* attrs, rgr_result* out, rgr_node_groups* groups);
void rgr_result_free(rgr_result* r);
int32_t rgr_match_filters(rgr_handle* h, const uint8_t* topics_blob, const uint64_t* topic_offsets, uint32_t n,
                          rgr_filters_result* out);
/* The same walk, but filter_ids[k] holds the SUB ID of the k-th matched filter's first subscriber (RGR_ID_NONE when the filter
 * has no subscriber) instead of the library's filter id.  This is the entry point for Router::matches (router.rs:174-265) on a
 * host that keeps the reference's relations map (AllRelationsMap, types.rs:476): sub id -> its relation -> the filter string ->
 * `relations.get(filter)`, then the per-client loop of router.rs:194-231 runs on the host's own map.  4 bytes per matched
 * FILTER cross PCIe instead of 12 per HIT, and no filter-id bookkeeping is needed (ids are per handle; sub ids are the caller's). */
int32_t rgr_match_filter_subs(rgr_handle* h, const uint8_t* topics_blob, const uint64_t* topic_offsets, uint32_t n,
                              rgr_filters_result* out);
void rgr_filters_result_free(rgr_filters_result* r);

/* ---- matching, device-resident batches (bench / multi-GPU / streaming consumers) ------ */
/* Parse + tokenise the topics against the current dictionary and leave the token
 * arrays resident in HBM. */
int32_t rgr_batch_create(rgr_handle* h, const uint8_t* topics_blob, const uint64_t* topic_offsets, uint32_t n,
                         rgr_batch** out);
/* PUBLISH-packet form of rgr_batch_create (the step right before the path): the batch is built
 * from raw MQTT PUBLISH packets, each entry exactly one frame as the codec delimits it (first byte, remaining
 * length, body: rmqtt-codec/src/v3/codec.rs:63-97).  The topic-name field, QoS, RETAIN, DUP and the packet id are
 * extracted on the device exactly as v3/decode.rs:110-128 (version 3 / 4) and v5/packet/publish.rs:31-101 (version
 * 5, including the property block's validation) do, and the topics go straight to the device tokeniser.
 * Per-packet status: RGR_TOPIC_OK, RGR_TOPIC_INVALID (decoded fine, Topic::from_str fails) or
 * RGR_PACKET_MALFORMED (the codec's DecodeError: such a publish never reaches the router).
 * from_ids != NULL ([n] owner ids, RGR_ID_NONE allowed): the batch also carries publish attributes built from
 * the packets' own QoS / RETAIN bits, i.e. every pass runs the delivery stage.  A v5 publish that uses a topic
 * alias arrives with an empty topic name: resolve aliases before (session state, rmqtt/src/session.rs). */
typedef struct rgr_publish_info {
    uint64_t topic_off;         /* offset of the topic-name bytes inside the packets blob                 */
    uint32_t topic_len;
    uint32_t payload_off;       /* offset of the payload inside the packet                                */
    uint16_t packet_id;         /* 0 = none (QoS 0)                                                       */
    uint8_t qos, retain, dup;
    uint8_t error;              /* 0 ok; 1 not a PUBLISH frame, 2 InvalidLength, 3 MalformedPacket, 4 Utf8Error */
    uint8_t _pad[2];
} rgr_publish_info;
int32_t rgr_batch_create_from_publish(rgr_handle* h, const uint8_t* packets, const uint64_t* packet_offsets, uint32_t n, uint32_t version,
                                      const uint32_t* from_ids, rgr_batch** out);
const rgr_publish_info* rgr_batch_publish_info(const rgr_batch* b);      /* [n], host pointer, batch lifetime */
void rgr_batch_destroy(rgr_batch* b);
/* [n] statuses computed by the tokeniser (host pointer, valid for the batch lifetime) */
const int32_t* rgr_batch_status(const rgr_batch* b);
/* Attach publish attributes ([n], host memory, copied) to a router batch: every later pass
 * runs the delivery stage and emits delivery words (RGR_HIT_*) in rgr_tuple.qos_flags.
 * NULL detaches them.  RGR_ESTATE inside a pass or on a retain batch. */
int32_t rgr_batch_set_publish_attrs(rgr_batch* b, const rgr_publish_attr* attrs);
/* Make later passes write ids[i] instead of i into rgr_tuple.topic_idx (ids: [n], host memory, copied; NULL
 * restores the batch index).  This is how a shard of a multi-GPU router reports the caller's GLOBAL publish
 * index (rgr_group_*), and how a broker's micro-batcher can tag hits with its own publish sequence numbers.
 * Costs nothing per hit (the id is attached per (topic, filter) pair at compaction).  RGR_ESTATE inside a pass
 * or together with publish attributes. */
int32_t rgr_batch_set_topic_ids(rgr_batch* b, const uint32_t* ids);
/* Order in which later passes walk the batch's topics.  RGR_ORDER_WALK: the library sorts the topics by their leading level tokens (three device
 * radix sorts per batch, repeated when a grown dictionary re-tokenises it): neighbouring lanes of the walk then share their upper trie levels
 * and hot subscriber runs (walk -16..-24 %, profiles/r06g_*).  Nothing changes per topic — same hits, same order inside a topic, rgr_tuple.topic_idx
 * still the batch index (or rgr_batch_set_topic_ids' id) — but windows enumerate topics in walk order: see rgr_window.d_topic_order;
 * rgr_batch_topic_order returns the whole permutation ([n], host pointer valid until the next rgr_batch_set_order / rgr_batch_begin / destroy;
 * NULL in caller order).  Device-resident publish batches (RGR_ESTATE on a retain batch or inside a pass).  With publish attributes attached the
 * delivery stage indexes them by walk position (the library gathers them) and answers in RGR_FORMAT_DELIVER8 only — the 12-byte delivery
 * tuple's topic column would name walk positions: rgr_batch_begin refuses that combination. */
enum { RGR_ORDER_CALLER = 0, RGR_ORDER_WALK = 1 };
int32_t rgr_batch_set_order(rgr_batch* b, uint32_t order);
const uint32_t* rgr_batch_topic_order(const rgr_batch* b);
/* Choose the result format of later passes (RGR_FORMAT_*).  RGR_ESTATE inside a pass; with publish attributes attached only
 * RGR_FORMAT_TUPLE and RGR_FORMAT_DELIVER8 carry the delivery word (RGR_FORMAT_DELIVER8 without them is RGR_ESTATE as well;
 * detaching the attributes returns such a batch to RGR_FORMAT_TUPLE). */
int32_t rgr_batch_set_format(rgr_batch* b, uint32_t format);
/* Start a pass over the batch (binds the current epoch, rewinds the window cursor). */
int32_t rgr_batch_begin(rgr_batch* b);
/* Walk (as needed) and expand the next window of topics.  RGR_EOF after the last. */
int32_t rgr_batch_next_window(rgr_batch* b, rgr_window* w);
/* Copy a window's tuples to host memory (blocking). */
int32_t rgr_window_to_host(rgr_batch* b, const rgr_window* w, rgr_tuple* host_tuples, uint64_t* host_hit_offsets);
/* One full pass: begin + every window (tuples stay on device, each window overwrites the
 * previous one).  *n_hits / *n_windows are optional. */
int32_t rgr_batch_run(rgr_batch* b, uint64_t* n_hits, uint32_t* n_windows);

/* One full pass whose windows are streamed to the HOST: every window is expanded into one of two
 * device buffers and copied (asynchronously, overlapping the next window's expansion) into a
 * library-owned pinned staging ring; `consume` (optional) is called with each window's host tuples
 * before its staging slot is reused.  This is the PCIe-inclusive form of rgr_batch_run. */
typedef void (*rgr_window_consumer)(void* user, uint32_t topic_begin, uint32_t topic_end, const rgr_tuple* host_tuples,
                                    uint64_t n_hits);
int32_t rgr_batch_run_to_host(rgr_batch* b, rgr_window_consumer consume, void* user, uint64_t* n_hits, uint32_t* n_windows);

/* rgr_tuple.qos_flags bit of a retained-path hit whose topic was removed after the base tier was
 * compiled (two-tier mode, device-resident windows only; rgr_retain_match_batch drops such hits) */
#define RGR_RETAIN_HIT_DEAD 1u

/* ---- retained-message twin (RetainTree) ----------------------------------------------- */
/* Insert / replace a retained topic carrying caller value `topic_id`. */
int32_t rgr_retain_topic_add(rgr_handle* h, const char* topic, uint32_t len, uint32_t topic_id);
int32_t rgr_retain_topic_remove(rgr_handle* h, const char* topic, uint32_t len);
/* Device-resident batch against ONE tier (two-tier mode): 0 = base, 1 = delta (RGR_ENOENT when the
 * delta tier is empty).  The complete answer of a filter is its base hits without RGR_RETAIN_HIT_DEAD
 * ones plus its delta hits.  rgr_retain_batch_create is tier 0. */
int32_t rgr_retain_batch_create_tier(rgr_handle* h, const uint8_t* filters_blob, const uint64_t* filter_offsets, uint32_t n,
                                     uint32_t tier, rgr_batch** out);
int32_t rgr_retain_add_bulk(rgr_handle* h, const uint8_t* blob, const uint64_t* offsets, uint64_t n,
                            const uint32_t* topic_ids, uint64_t* n_rejected);
int32_t rgr_retain_commit(rgr_handle* h);
typedef struct rgr_retain_result {
    uint32_t n_filters;
    uint64_t n_hits;
    int32_t* status;            /* [n_filters]                                          */
    uint64_t* hit_offsets;      /* [n_filters+1]                                        */
    uint32_t* topic_ids;        /* [n_hits] per filter, deterministic (trie preorder); the
                                   reference's own order is hash-map order: compare as sets */
    void* _owner;
} rgr_retain_result;
int32_t rgr_retain_match_batch(rgr_handle* h, const uint8_t* filters_blob, const uint64_t* filter_offsets, uint32_t n,
                               rgr_retain_result* out);
/* Device-resident form: a batch of SUBSCRIBE filters; drive it with rgr_batch_begin /
 * rgr_batch_next_window / rgr_batch_run exactly like a publish batch.  In its windows
 * rgr_tuple.topic_idx is the filter index and rgr_tuple.sub_id the retained topic_id. */
int32_t rgr_retain_batch_create(rgr_handle* h, const uint8_t* filters_blob, const uint64_t* filter_offsets, uint32_t n,
                                rgr_batch** out);
void rgr_retain_result_free(rgr_retain_result* r);

/* ---- the retained path's DENSE answer: ranges of the preorder value array ----------------------------------------------------------
 * A consumer of RetainStorage::get (rmqtt-retainer/src/storage.rs:604-611; DefaultRetainStorage::get_message retain.rs:250-267)
 * needs the matched TOPIC IDS, nothing per hit from the device.  The compiled image lays the retained topics out in trie preorder
 * (DESIGN §3), so `a/#` is ONE contiguous range of the value array and a `+` level a handful: the answer of a filter is a short list of
 * ranges into an array the host already mirrors (8 bytes per retained topic, refreshed at rgr_retain_commit).  What crosses PCIe is
 * 16 bytes per RANGE instead of 12 per hit — at BASELINE configs[4] 25.9 k hits per filter travel as ~10 ranges.
 *   hits of filter i  =  for r in ranges[range_offsets[i] .. range_offsets[i+1]):  vals[tier(r)][r.begin .. r.begin + len(r))
 * in the order rgr_retain_match_batch returns them; tier(r) = r.len >> 31 (1 = the delta tier of the two-tier mode), len(r) =
 * r.len & 0x7fffffff.  Entries whose flags carry RGR_RETAIN_HIT_DEAD were removed / replaced after the base tier was compiled: skip
 * them (rgr_retain_match_batch does the same).  `vals` stay valid until rgr_retain_ranges_free, across later commits. */
typedef struct rgr_id_range { uint32_t begin, len; } rgr_id_range;
typedef struct rgr_retain_val { uint32_t topic_id, flags; } rgr_retain_val;
typedef struct rgr_retain_ranges {
    uint32_t n_filters;
    uint64_t n_ranges;
    uint64_t n_entries;             /* sum of the ranges' lengths (dead entries included)   */
    int32_t* status;                /* [n_filters]                                          */
    uint64_t* range_offsets;        /* [n_filters+1]                                        */
    rgr_id_range* ranges;           /* [n_ranges]                                           */
    const rgr_retain_val* vals[2];  /* host mirrors of the value arrays: [0] base, [1] delta tier (NULL without one) */
    uint64_t n_vals[2];
    void* _owner;
} rgr_retain_ranges;
int32_t rgr_retain_match_ranges(rgr_handle* h, const uint8_t* filters_blob, const uint64_t* filter_offsets, uint32_t n,
                                rgr_retain_ranges* out);
void rgr_retain_ranges_free(rgr_retain_ranges* r);

/* Device-resident retain batches: answer with POSITIONS.  Later passes write, into rgr_tuple.sub_id, the position of the hit in the epoch's
 * preorder value array instead of the topic id stored there: topic id = vals[position].topic_id, with `vals` the host mirror
 * rgr_batch_retain_vals returns (the array rgr_retain_ranges.vals points at).  The expansion then reads nothing per hit — a trailing '#' is a
 * contiguous range of that array (retain.rs:502-524), its hits are consecutive numbers.  rgr_tuple.qos_flags is 0 in this form: in two-tier
 * mode the RGR_RETAIN_HIT_DEAD bit of a hit is vals[position].flags.  RGR_ESTATE inside a pass or on a publish batch. */
int32_t rgr_batch_set_retain_positions(rgr_batch* b, int32_t on);
/* Host mirror of the value array of the epoch (and tier) the batch's last rgr_batch_begin bound; valid until the next rgr_batch_begin /
 * rgr_batch_destroy of this batch. */
int32_t rgr_batch_retain_vals(const rgr_batch* b, const rgr_retain_val** vals, uint64_t* n);

/* ---- multi-GPU sharding rule (host-side helper, no device work) -------------------------
 * Table and publishes shard by a hash of the first `key_levels` topic levels (SURVEY.md §8(e):
 * a first-level-only hash is far too skewed under Zipf level-0 tokens; 3 levels keep the
 * hottest shard key near 1 % of the hits):
 *   topic  -> shard = H(level0 .. level_{k-1} | shorter topics hash what they have) mod n_shards
 *   filter -> the same, unless one of its first `key_levels` levels is a wildcard: then -1 =
 *             replicate on every shard.  A topic's complete match set then lives on its owner
 *             shard, so the data path needs no collective.
 * out[i] in [0,n_shards) or -1; invalid topics/filters get a shard too (they match nothing).
 * key_levels: 1..8 (0 = default 3). */
int32_t rgr_shard_assign(const uint8_t* blob, const uint64_t* offsets, uint64_t n, uint32_t n_shards, int32_t is_filter,
                         uint32_t key_levels, int32_t* out);

/* ---- multi-GPU: communicators and single-process shard groups ----------------
 * With the rule above the DATA path needs no collective (a topic's whole hit list comes from its owner
 * GPU); the exchange step is every rank learning every rank's hits, over RCCL / xGMI:
 *   counts  one ncclAllGather per step                                       rgr_comm_allgather_u64
 *   tuples  all-gatherv = counts + ncclGroupStart { ncclSend / ncclRecv per peer, exact sizes } ncclGroupEnd
 *           (point-to-point on every xGMI link at once — a ring would be per-link bound), overlapped with
 *           the next window's expansion                                      rgr_comm_gather_pass
 * A communicator belongs to one handle (= one device).  One rank per process: rank 0 calls
 * rgr_comm_unique_id, ships the 128 bytes to the other ranks by any means (the launcher's rendezvous,
 * torch.distributed, a file) and every rank calls rgr_comm_create (ncclCommInitRank).  One process with
 * several devices: rgr_group_create builds a handle + communicator per device (ncclCommInitAll) and
 * shards tables and batches across them.  RCCL is dlopen'ed at first use. */
#define RGR_COMM_ID_BYTES 128
typedef struct rgr_comm rgr_comm;
int32_t rgr_comm_unique_id(uint8_t* id /* [RGR_COMM_ID_BYTES] */);
int32_t rgr_comm_create(rgr_handle* h, const uint8_t* id, uint32_t rank, uint32_t world, rgr_comm** out);
void rgr_comm_destroy(rgr_comm* c);
/* What the transport itself reports about this communicator (a launcher's evidence that N ranks really formed ONE RCCL world):
 * ranks = ncclCommCount, rank = ncclCommUserRank, device = the handle's HIP ordinal; transport 1 = RCCL (xGMI / PCIe between
 * devices), 0 = device copies between shards that share one GPU (single-process groups on a test rig). */
typedef struct rgr_comm_info_t { uint32_t ranks, rank; int32_t device; int32_t transport; } rgr_comm_info_t;
int32_t rgr_comm_info(rgr_comm* c, rgr_comm_info_t* out);
/* every rank's value on every rank: all[world] */
int32_t rgr_comm_allgather_u64(rgr_comm* c, uint64_t mine, uint64_t* all);
/* One pass of batch `b` (created on the communicator's handle) whose windows are all-gathered: in every round
 * each rank contributes its next window (or nothing once it ran out) and receives all ranks' tuples in rank
 * order.  consume (optional) sees each round's gathered DEVICE buffer — valid during the call only — with the
 * per-rank tuple counts.  Collective: every rank must call it.  Tuples carry rgr_batch_set_topic_ids' ids. */
typedef void (*rgr_gather_consumer)(void* user, const rgr_tuple* d_tuples, const uint64_t* counts, uint32_t world, uint64_t n_total);
int32_t rgr_comm_gather_pass(rgr_comm* c, rgr_batch* b, rgr_gather_consumer consume, void* user, uint64_t* my_hits, uint64_t* all_hits);

/* ---- the exchange step with RUN DESCRIPTORS instead of tuples (BASELINE configs[3] at full fan-out) -------------------------
 * All-gathering every 12-byte tuple to every rank moves 8 x 1.78 TB per pass at config-3 fan-out.  What a rank needs in order to
 * KNOW every rank's hits is much smaller: the subscriber entries are table data (8 bytes per subscription: 80 MB at 10 M), so
 * every rank keeps a replica of every rank's subs[] (rgr_comm_replicate_subs: one all-gatherv after a commit), and a pass then
 * all-gathers only the RUN descriptors — 16 bytes per (topic, matched filter with subscribers) pair.  The hits of descriptor d
 * are peer_subs(d.shard)[d.src .. d.src + d.len), in order, for topic d.topic (the caller's id when rgr_batch_set_topic_ids was
 * used), exactly the tuples rgr_comm_gather_pass would have delivered for that run. */
typedef struct rgr_run { uint32_t shard, src, len, topic; } rgr_run;
/* Replicate the current epoch's subscriber entries of every rank on every rank.  Collective: every rank calls it, after its
 * rgr_commit and before rgr_comm_gather_runs_pass (which refuses a batch whose epoch differs from the replicated one). */
int32_t rgr_comm_replicate_subs(rgr_comm* c);
/* Device pointer to this rank's replica of rank `rank`'s subs[] ({ sub_id, qos | flags << 8 | node_idx << 16 }, 8 bytes each). */
int32_t rgr_comm_peer_subs(rgr_comm* c, uint32_t rank, const uint64_t** d_subs, uint64_t* n_entries);
/* Called once per round with the descriptors of every rank (rank p's counts[p] descriptors start at sum(counts[0..p))). */
typedef void (*rgr_runs_consumer)(void* user, const rgr_run* d_runs, const uint64_t* counts, uint32_t world, uint64_t n_total);
int32_t rgr_comm_gather_runs_pass(rgr_comm* c, rgr_batch* b, rgr_runs_consumer consume, void* user, uint64_t* my_runs, uint64_t* all_runs,
                                  uint64_t* all_hits);

typedef struct rgr_group rgr_group;
typedef struct rgr_group_batch rgr_group_batch;
/* One handle per entry of devices[] (cfg->device is ignored).  Distinct ordinals: RCCL communicators over
 * xGMI.  Repeated ordinals (several shards on one GPU — a test rig): the shards exchange through device copies
 * with the same protocol; rgr_group_uses_rccl tells which. */
int32_t rgr_group_create(const rgr_config* cfg, const int32_t* devices, uint32_t n_devices, rgr_group** out);
void rgr_group_destroy(rgr_group* g);
uint32_t rgr_group_size(const rgr_group* g);
/* Leading topic levels hashed into the shard key (default 3; 1 = SURVEY 8(e)'s first-level rule).  Only while the group is empty. */
int32_t rgr_group_set_key_levels(rgr_group* g, uint32_t key_levels);
rgr_handle* rgr_group_handle(rgr_group* g, uint32_t shard);          /* per-shard stats, snapshots, retain twin ... */
rgr_comm* rgr_group_comm(rgr_group* g, uint32_t shard);
int32_t rgr_group_uses_rccl(const rgr_group* g);
/* Router::add / remove / the restore loop, routed by rgr_shard_assign: a filter goes to its owner shard, or
 * to every shard when a wildcard sits in its key levels.  sub ids stay the caller's. */
int32_t rgr_group_subscribe_bulk(rgr_group* g, const uint8_t* blob, const uint64_t* offsets, uint64_t n, const uint32_t* sub_ids,
                                 const uint8_t* qos, const uint8_t* flags, uint64_t* n_rejected);
/* delivery-stage attributes of bulk-loaded subscriptions (rgr_sub_attrs_bulk), per sub id */
int32_t rgr_group_sub_attrs_bulk(rgr_group* g, const uint32_t* sub_ids, const uint32_t* owner_ids, const uint32_t* client_idx, uint64_t n);
int32_t rgr_group_subscribe(rgr_group* g, const char* filter, uint32_t len, uint32_t sub_id, uint8_t qos, uint8_t flags);
/* with the delivery-stage attributes of rgr_sub_add_ex */
int32_t rgr_group_subscribe_ex(rgr_group* g, const char* filter, uint32_t len, uint32_t sub_id, uint8_t qos, uint8_t flags,
                               uint16_t node_idx, uint32_t owner_id, uint32_t client_idx);
/* last_of_filter != 0: the caller's relations map for this filter became empty (router.rs:484-490): prune it */
int32_t rgr_group_unsubscribe(rgr_group* g, const char* filter, uint32_t len, uint32_t sub_id, int32_t last_of_filter);
int32_t rgr_group_commit(rgr_group* g);
/* Router::matches over the group, host buffers in / out: identical to rgr_match_batch on one handle holding
 * the whole table (same order, topic_idx = the caller's index). */
int32_t rgr_group_match_batch(rgr_group* g, const uint8_t* topics_blob, const uint64_t* topic_offsets, uint32_t n, rgr_result* out);
int32_t rgr_group_match_batch_deliver(rgr_group* g, const uint8_t* topics_blob, const uint64_t* topic_offsets, uint32_t n,
                                      const rgr_publish_attr* attrs, rgr_result* out);
int32_t rgr_group_match_batch_deliver_grouped(rgr_group* g, const uint8_t* topics_blob, const uint64_t* topic_offsets, uint32_t n,
                                              const rgr_publish_attr* attrs, rgr_result* out, rgr_node_groups* groups);
/* rgr_match_filter_subs over the group (topics routed to their owner shard, lists stitched back into the caller's order). */
int32_t rgr_group_match_filter_subs(rgr_group* g, const uint8_t* topics_blob, const uint64_t* topic_offsets, uint32_t n, rgr_filters_result* out);
/* Device-resident form: the batch is split by owner shard; tuples carry the caller's topic index. */
int32_t rgr_group_batch_create(rgr_group* g, const uint8_t* topics_blob, const uint64_t* topic_offsets, uint32_t n, rgr_group_batch** out);
void rgr_group_batch_destroy(rgr_group_batch* gb);
rgr_batch* rgr_group_batch_shard(rgr_group_batch* gb, uint32_t shard);
/* one pass on all shards at once, tuples left on their GPUs; the per-shard hit counts are all-gathered */
int32_t rgr_group_batch_run(rgr_group_batch* gb, uint64_t* shard_hits /* [size], optional */, uint64_t* total_hits);
/* one pass with every window all-gathered to every shard; consume runs on shard `consumer_shard`'s thread */
int32_t rgr_group_batch_gather(rgr_group_batch* gb, uint32_t consumer_shard, rgr_gather_consumer consume, void* user, uint64_t* total_hits);
/* The run-descriptor form of rgr_group_batch_gather (rgr_comm_replicate_subs on every shard first when its table changed). */
int32_t rgr_group_batch_gather_runs(rgr_group_batch* gb, uint32_t consumer_shard, rgr_runs_consumer consume, void* user, uint64_t* total_runs,
                                    uint64_t* total_hits);
/* Device pointer to shard `holder`'s replica of shard `of`'s subs[] (valid until the next rgr_group_batch_gather_runs after a commit). */
int32_t rgr_group_peer_subs(rgr_group* g, uint32_t holder, uint32_t of, const uint64_t** d_subs, uint64_t* n_entries);

/* The retained-message twin over the group: retained topics live on the shard their key levels hash to; a SUBSCRIBE filter whose
 * key levels are literal asks that one shard, a filter with a wildcard among them asks every shard and the answers are concatenated (shard order).
 * Same meaning as the single-handle calls (RetainTree insert / remove / matches, rmqtt/src/retain.rs:373-413, 450-526). */
int32_t rgr_group_retain_topic_add(rgr_group* g, const char* topic, uint32_t len, uint32_t topic_id);
int32_t rgr_group_retain_topic_remove(rgr_group* g, const char* topic, uint32_t len);
int32_t rgr_group_retain_add_bulk(rgr_group* g, const uint8_t* blob, const uint64_t* offsets, uint64_t n, const uint32_t* topic_ids, uint64_t* n_rejected);
int32_t rgr_group_retain_commit(rgr_group* g);
int32_t rgr_group_retain_match_batch(rgr_group* g, const uint8_t* filters_blob, const uint64_t* filter_offsets, uint32_t n, rgr_retain_result* out);

/* ---- observability ------------------------------------------------------------------------ */
int32_t rgr_stats_get(rgr_handle* h, rgr_stats* out);
int32_t rgr_stats_reset(rgr_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* RMQTT_GPU_ROUTER_H */
