//! Raw bindings of include/rmqtt_gpu_router.h (hand-written; identical to what bindgen emits).
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_void};

pub const RGR_OK: i32 = 0;
pub const RGR_EOF: i32 = 1;
pub const RGR_TOPIC_OK: i32 = 0;
pub const RGR_SUB_V5: u8 = 1;
pub const RGR_SUB_NO_LOCAL: u8 = 2;
pub const RGR_SUB_SHARED: u8 = 4;
pub const RGR_SUB_RAP: u8 = 8;
// delivery word bits (rgr_tuple.qos_flags of batches that carry publish attributes)
pub const RGR_HIT_QOS_MASK: u32 = 3;
pub const RGR_HIT_RETAIN: u32 = 1 << 2;
pub const RGR_HIT_NO_LOCAL: u32 = 1 << 3;
pub const RGR_HIT_V5_DUP: u32 = 1 << 4;
pub const RGR_ID_NONE: u32 = 0xFFFF_FFFF;

#[repr(C)]
pub struct rgr_handle { _p: [u8; 0] }

#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct rgr_config {
    pub device: i32,
    pub slot_cap: u32,
    pub window_hits: u64,
    pub chunk_topics: u32,
    pub host_threads: u32,
    pub collect_walk_stats: u32,
    pub host_tokenize: u32,
    pub retain_delta_max: u32,
    pub _reserved0: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rgr_tuple { pub topic_idx: u32, pub sub_id: u32, pub qos_flags: u32 }
#[repr(C)]
#[derive(Clone, Copy)]
pub struct rgr_publish_attr { pub from_id: u32, pub qos_retain: u32 }

#[repr(C)]
pub struct rgr_result {
    pub n_topics: u32,
    pub n_hits: u64,
    pub status: *mut i32,
    pub hit_offsets: *mut u64,
    pub tuples: *mut rgr_tuple,
    pub _owner: *mut c_void,
}

#[repr(C)]
pub struct rgr_retain_result {
    pub n_filters: u32,
    pub n_hits: u64,
    pub status: *mut i32,
    pub hit_offsets: *mut u64,
    pub topic_ids: *mut u32,
    pub _owner: *mut c_void,
}

extern "C" {
    pub fn rgr_create(cfg: *const rgr_config, out: *mut *mut rgr_handle) -> i32;
    pub fn rgr_destroy(h: *mut rgr_handle);
    pub fn rgr_last_error() -> *const c_char;
    pub fn rgr_filter_add(h: *mut rgr_handle, filter: *const c_char, len: u32, filter_id: *mut u32) -> i32;
    pub fn rgr_filter_remove(h: *mut rgr_handle, filter_id: u32) -> i32;
    pub fn rgr_sub_add(h: *mut rgr_handle, filter_id: u32, sub_id: u32, qos: u8, flags: u8) -> i32;
    pub fn rgr_sub_add_ex(h: *mut rgr_handle, filter_id: u32, sub_id: u32, qos: u8, flags: u8, node_idx: u16, owner_id: u32,
                          client_idx: u32) -> i32;
    pub fn rgr_sub_remove(h: *mut rgr_handle, filter_id: u32, sub_id: u32) -> i32;
    pub fn rgr_commit(h: *mut rgr_handle) -> i32;
    pub fn rgr_match_batch(h: *mut rgr_handle, blob: *const u8, offsets: *const u64, n: u32, out: *mut rgr_result) -> i32;
    pub fn rgr_match_batch_deliver(h: *mut rgr_handle, blob: *const u8, offsets: *const u64, n: u32, attrs: *const rgr_publish_attr,
                                   out: *mut rgr_result) -> i32;
    pub fn rgr_result_free(r: *mut rgr_result);
    pub fn rgr_retain_topic_add(h: *mut rgr_handle, topic: *const c_char, len: u32, topic_id: u32) -> i32;
    pub fn rgr_retain_topic_remove(h: *mut rgr_handle, topic: *const c_char, len: u32) -> i32;
    pub fn rgr_retain_commit(h: *mut rgr_handle) -> i32;
    pub fn rgr_retain_match_batch(h: *mut rgr_handle, blob: *const u8, offsets: *const u64, n: u32, out: *mut rgr_retain_result) -> i32;
    pub fn rgr_retain_result_free(r: *mut rgr_retain_result);
    pub fn rgr_subscribe_bulk(h: *mut rgr_handle, blob: *const u8, offsets: *const u64, n: u64, sub_ids: *const u32,
                              qos: *const u8, flags: *const u8, filter_ids_out: *mut u32, n_rejected: *mut u64) -> i32;
    pub fn rgr_shard_assign(blob: *const u8, offsets: *const u64, n: u64, n_shards: u32, is_filter: i32, key_levels: u32,
                            out: *mut i32) -> i32;
}

// ---- multi-GPU group (one handle per device; include/rmqtt_gpu_router.h "multi-GPU") ----------------
#[repr(C)]
pub struct rgr_group { _p: [u8; 0] }

extern "C" {
    pub fn rgr_group_create(cfg: *const rgr_config, devices: *const i32, n_devices: u32, out: *mut *mut rgr_group) -> i32;
    pub fn rgr_group_destroy(g: *mut rgr_group);
    pub fn rgr_group_size(g: *const rgr_group) -> u32;
    pub fn rgr_group_handle(g: *mut rgr_group, shard: u32) -> *mut rgr_handle;
    pub fn rgr_group_subscribe_ex(g: *mut rgr_group, filter: *const c_char, len: u32, sub_id: u32, qos: u8, flags: u8, node_idx: u16,
                                  owner_id: u32, client_idx: u32) -> i32;
    pub fn rgr_group_unsubscribe(g: *mut rgr_group, filter: *const c_char, len: u32, sub_id: u32, last_of_filter: i32) -> i32;
    pub fn rgr_group_subscribe_bulk(g: *mut rgr_group, blob: *const u8, offsets: *const u64, n: u64, sub_ids: *const u32, qos: *const u8,
                                    flags: *const u8, n_rejected: *mut u64) -> i32;
    pub fn rgr_group_sub_attrs_bulk(g: *mut rgr_group, sub_ids: *const u32, owner_ids: *const u32, client_idx: *const u32, n: u64) -> i32;
    pub fn rgr_group_commit(g: *mut rgr_group) -> i32;
    pub fn rgr_group_retain_topic_add(g: *mut rgr_group, topic: *const c_char, len: u32, topic_id: u32) -> i32;
    pub fn rgr_group_retain_topic_remove(g: *mut rgr_group, topic: *const c_char, len: u32) -> i32;
    pub fn rgr_group_retain_add_bulk(g: *mut rgr_group, blob: *const u8, offsets: *const u64, n: u64, topic_ids: *const u32, n_rejected: *mut u64) -> i32;
    pub fn rgr_group_retain_commit(g: *mut rgr_group) -> i32;
    pub fn rgr_group_retain_match_batch(g: *mut rgr_group, filters_blob: *const u8, filter_offsets: *const u64, n: u32, out: *mut rgr_retain_result) -> i32;
    pub fn rgr_group_match_batch(g: *mut rgr_group, blob: *const u8, offsets: *const u64, n: u32, out: *mut rgr_result) -> i32;
    pub fn rgr_group_match_batch_deliver(g: *mut rgr_group, blob: *const u8, offsets: *const u64, n: u32, attrs: *const rgr_publish_attr,
                                         out: *mut rgr_result) -> i32;
    pub fn rgr_group_match_filter_subs(g: *mut rgr_group, blob: *const u8, offsets: *const u64, n: u32, out: *mut rgr_filters_result) -> i32;
}

// ---- device-resident batches, result formats, PUBLISH-packet batches, communicators (not used by the plugin itself;
// declared for consumers that keep the hits on the device) -----------------------------------------------------------
#[repr(C)]
pub struct rgr_batch { _p: [u8; 0] }
#[repr(C)]
pub struct rgr_comm { _p: [u8; 0] }

pub const RGR_FORMAT_TUPLE: u32 = 0;
pub const RGR_FORMAT_SOA: u32 = 1;
pub const RGR_FORMAT_PACKED: u32 = 2;
pub const RGR_FORMAT_RUNS: u32 = 3;
pub const RGR_FORMAT_IDS24: u32 = 4;
pub const RGR_FORMAT_DELIVER8: u32 = 5;
pub const RGR_ORDER_CALLER: u32 = 0;
pub const RGR_ORDER_WALK: u32 = 1;
pub const RGR_TOPIC_INVALID: i32 = -2;
pub const RGR_PACKET_MALFORMED: i32 = -8;
pub const RGR_COMM_ID_BYTES: usize = 128;

#[repr(C)]
pub struct rgr_window {
    pub topic_begin: u32,
    pub topic_end: u32,
    pub n_hits: u64,
    pub hit_base: u64,
    pub d_tuples: *const rgr_tuple,
    pub d_hit_offsets: *const u64,
    pub offsets_bias: u64,
    pub d_sub_ids: *const u32,
    pub d_qos: *const u8,
    pub n_runs: u64,
    pub d_run_src: *const u32,
    pub d_run_topic: *const u32,
    pub d_run_off: *const u64,
    pub d_subs: *const u64,
    pub d_ids24: *const u8,
    pub d_hits8: *const rgr_hit8,
    pub d_topic_order: *const u32,
}

/// One hit of a delivery pass in RGR_FORMAT_DELIVER8: relation + delivery word (the topic is implied by the CSR offsets).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rgr_hit8 {
    pub sub_id: u32,
    pub word: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rgr_publish_info {
    pub topic_off: u64,
    pub topic_len: u32,
    pub payload_off: u32,
    pub packet_id: u16,
    pub qos: u8,
    pub retain: u8,
    pub dup: u8,
    pub error: u8,
    pub _pad: [u8; 2],
}

extern "C" {
    pub fn rgr_batch_create(h: *mut rgr_handle, blob: *const u8, offsets: *const u64, n: u32, out: *mut *mut rgr_batch) -> i32;
    pub fn rgr_batch_create_from_publish(h: *mut rgr_handle, packets: *const u8, packet_offsets: *const u64, n: u32, version: u32,
                                         from_ids: *const u32, out: *mut *mut rgr_batch) -> i32;
    pub fn rgr_batch_publish_info(b: *const rgr_batch) -> *const rgr_publish_info;
    pub fn rgr_batch_status(b: *const rgr_batch) -> *const i32;
    pub fn rgr_batch_destroy(b: *mut rgr_batch);
    pub fn rgr_batch_set_publish_attrs(b: *mut rgr_batch, attrs: *const rgr_publish_attr) -> i32;
    pub fn rgr_batch_set_topic_ids(b: *mut rgr_batch, ids: *const u32) -> i32;
    pub fn rgr_batch_set_format(b: *mut rgr_batch, format: u32) -> i32;
    pub fn rgr_batch_set_order(b: *mut rgr_batch, order: u32) -> i32;
    pub fn rgr_batch_topic_order(b: *const rgr_batch) -> *const u32;
    pub fn rgr_batch_set_retain_positions(b: *mut rgr_batch, on: i32) -> i32;
    pub fn rgr_batch_retain_vals(b: *const rgr_batch, vals: *mut *const rgr_retain_val, n: *mut u64) -> i32;
    pub fn rgr_batch_begin(b: *mut rgr_batch) -> i32;
    pub fn rgr_batch_next_window(b: *mut rgr_batch, w: *mut rgr_window) -> i32;
    pub fn rgr_batch_run(b: *mut rgr_batch, n_hits: *mut u64, n_windows: *mut u32) -> i32;
    pub fn rgr_match_filters(h: *mut rgr_handle, blob: *const u8, offsets: *const u64, n: u32, out: *mut rgr_filters_result) -> i32;
    pub fn rgr_match_filter_subs(h: *mut rgr_handle, blob: *const u8, offsets: *const u64, n: u32, out: *mut rgr_filters_result) -> i32;
    pub fn rgr_filters_result_free(r: *mut rgr_filters_result);
    pub fn rgr_comm_unique_id(id: *mut u8) -> i32;
    pub fn rgr_comm_create(h: *mut rgr_handle, id: *const u8, rank: u32, world: u32, out: *mut *mut rgr_comm) -> i32;
    pub fn rgr_comm_destroy(c: *mut rgr_comm);
    pub fn rgr_comm_allgather_u64(c: *mut rgr_comm, mine: u64, all: *mut u64) -> i32;
    pub fn rgr_comm_info(c: *mut rgr_comm, out: *mut rgr_comm_info_t) -> i32;
    pub fn rgr_retain_match_ranges(h: *mut rgr_handle, blob: *const u8, offsets: *const u64, n: u32, out: *mut rgr_retain_ranges) -> i32;
    pub fn rgr_retain_ranges_free(r: *mut rgr_retain_ranges);
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rgr_id_range {
    pub begin: u32,
    pub len: u32, // bit 31: the delta tier's value array
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rgr_retain_val {
    pub topic_id: u32,
    pub flags: u32, // RGR_RETAIN_HIT_DEAD: removed / replaced since the base tier was compiled
}
pub const RGR_RETAIN_HIT_DEAD: u32 = 1;

#[repr(C)]
pub struct rgr_retain_ranges {
    pub n_filters: u32,
    pub n_ranges: u64,
    pub n_entries: u64,
    pub status: *mut i32,
    pub range_offsets: *mut u64,
    pub ranges: *mut rgr_id_range,
    pub vals: [*const rgr_retain_val; 2],
    pub n_vals: [u64; 2],
    pub _owner: *mut c_void,
}

#[repr(C)]
pub struct rgr_comm_info_t {
    pub ranks: u32,
    pub rank: u32,
    pub device: i32,
    pub transport: i32,
}

#[repr(C)]
pub struct rgr_filters_result {
    pub n_topics: u32,
    pub n_pairs: u64,
    pub status: *mut i32,
    pub pair_offsets: *mut u64,
    pub filter_ids: *mut u32,
    pub _owner: *mut c_void,
}
