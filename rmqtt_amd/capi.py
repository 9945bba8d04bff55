"""ctypes view of the C ABI in include/rmqtt_gpu_router.h (tests, bench, smoke).

No compute happens in Python: every call goes straight to librmqtt_gpu_router.so.
Importing this module does not need a GPU; ``Router()`` (rgr_create) does and raises
``RgrError`` with RGR_EDEVICE when none is usable — there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

RGR_OK, RGR_EOF = 0, 1
RGR_EINVAL, RGR_EINVAL_TOPIC, RGR_ENOMEM, RGR_EDEVICE, RGR_ECAPACITY, RGR_ENOENT, RGR_ESTATE = -1, -2, -3, -4, -5, -6, -7
RGR_TOPIC_OK, RGR_TOPIC_INVALID, RGR_PACKET_MALFORMED = 0, -2, -8
PUBLISH_INFO_DTYPE = np.dtype([("topic_off", np.uint64), ("topic_len", np.uint32), ("payload_off", np.uint32), ("packet_id", np.uint16),
                               ("qos", np.uint8), ("retain", np.uint8), ("dup", np.uint8), ("error", np.uint8), ("_pad", np.uint8, 2)])
RGR_SUB_V5, RGR_SUB_NO_LOCAL, RGR_SUB_SHARED, RGR_SUB_RAP = 1, 2, 4, 8
RGR_HIT_QOS_MASK, RGR_HIT_RETAIN, RGR_HIT_NO_LOCAL, RGR_HIT_V5_DUP = 3, 4, 8, 16
ID_NONE = 0xFFFFFFFF
RGR_SUB_TABLE_SHIFT, RGR_SUB_TABLE_MASK = 4, 0xF0     # flags bits 4-7: caller-defined table id
RGR_FORMAT_TUPLE, RGR_FORMAT_SOA, RGR_FORMAT_PACKED, RGR_FORMAT_RUNS, RGR_FORMAT_IDS24, RGR_FORMAT_DELIVER8 = 0, 1, 2, 3, 4, 5

TUPLE_DTYPE = np.dtype([("topic_idx", np.uint32), ("sub_id", np.uint32), ("qos_flags", np.uint32)])
PUBLISH_ATTR_DTYPE = np.dtype([("from_id", np.uint32), ("qos_retain", np.uint32)])

# every symbol include/rmqtt_gpu_router.h declares
SYMBOLS = [
    "rgr_create", "rgr_destroy", "rgr_last_error", "rgr_version",
    "rgr_filter_add", "rgr_filter_find", "rgr_filter_remove", "rgr_sub_add", "rgr_sub_add_ex", "rgr_sub_attrs_bulk",
    "rgr_sub_remove", "rgr_subscribe_bulk", "rgr_snapshot_save", "rgr_snapshot_load", "rgr_commit",
    "rgr_match_batch", "rgr_match_batch_deliver", "rgr_match_batch_deliver_grouped", "rgr_group_match_batch_deliver_grouped", "rgr_result_free", "rgr_match_filters", "rgr_match_filter_subs", "rgr_filters_result_free",
    "rgr_batch_create", "rgr_batch_create_from_publish", "rgr_batch_publish_info", "rgr_batch_destroy", "rgr_batch_status", "rgr_batch_set_publish_attrs",
    "rgr_batch_set_format", "rgr_batch_set_topic_ids", "rgr_batch_set_retain_positions", "rgr_batch_retain_vals", "rgr_batch_set_order", "rgr_batch_topic_order", "rgr_batch_begin", "rgr_batch_next_window",
    "rgr_window_to_host", "rgr_batch_run", "rgr_batch_run_to_host",
    "rgr_retain_topic_add", "rgr_retain_topic_remove", "rgr_retain_add_bulk", "rgr_retain_commit",
    "rgr_retain_match_batch", "rgr_retain_result_free", "rgr_retain_match_ranges", "rgr_retain_ranges_free", "rgr_retain_batch_create", "rgr_retain_batch_create_tier",
    "rgr_shard_assign", "rgr_stats_get", "rgr_stats_reset",
    "rgr_comm_unique_id", "rgr_comm_create", "rgr_comm_destroy", "rgr_comm_info", "rgr_comm_allgather_u64", "rgr_comm_gather_pass",
    "rgr_comm_replicate_subs", "rgr_comm_peer_subs", "rgr_comm_gather_runs_pass", "rgr_group_batch_gather_runs", "rgr_group_peer_subs",
    "rgr_group_create", "rgr_group_destroy", "rgr_group_size", "rgr_group_set_key_levels", "rgr_group_handle", "rgr_group_comm", "rgr_group_uses_rccl",
    "rgr_group_subscribe_bulk", "rgr_group_sub_attrs_bulk", "rgr_group_subscribe", "rgr_group_subscribe_ex", "rgr_group_unsubscribe", "rgr_group_commit",
    "rgr_group_match_batch", "rgr_group_match_batch_deliver", "rgr_group_match_filter_subs",
    "rgr_group_retain_topic_add", "rgr_group_retain_topic_remove", "rgr_group_retain_add_bulk", "rgr_group_retain_commit", "rgr_group_retain_match_batch",
    "rgr_group_batch_create", "rgr_group_batch_destroy", "rgr_group_batch_shard", "rgr_group_batch_run", "rgr_group_batch_gather",
]
GATHER_CONSUMER = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.c_uint64)
RUN_DTYPE = np.dtype([("shard", np.uint32), ("src", np.uint32), ("len", np.uint32), ("topic", np.uint32)])      # == rgr_run
SUB_ENTRY_DTYPE = np.dtype([("sub_id", np.uint32), ("qos_flags", np.uint32)])                                      # one subs[] entry
RGR_COMM_ID_BYTES = 128


class CommInfo(C.Structure):      # == rgr_comm_info_t
    _fields_ = [("ranks", C.c_uint32), ("rank", C.c_uint32), ("device", C.c_int32), ("transport", C.c_int32)]


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("slot_cap", C.c_uint32), ("window_hits", C.c_uint64),
                ("chunk_topics", C.c_uint32), ("host_threads", C.c_uint32), ("collect_walk_stats", C.c_uint32),
                ("host_tokenize", C.c_uint32), ("retain_delta_max", C.c_uint32), ("_reserved0", C.c_uint32)]


class Result(C.Structure):
    _fields_ = [("n_topics", C.c_uint32), ("n_hits", C.c_uint64), ("status", C.c_void_p),
                ("hit_offsets", C.c_void_p), ("tuples", C.c_void_p), ("_owner", C.c_void_p)]


class NodeGroups(C.Structure):
    _fields_ = [("n_groups", C.c_uint64), ("group_offsets", C.c_void_p), ("group_node", C.c_void_p), ("group_begin", C.c_void_p)]


class FiltersResult(C.Structure):
    _fields_ = [("n_topics", C.c_uint32), ("n_pairs", C.c_uint64), ("status", C.c_void_p),
                ("pair_offsets", C.c_void_p), ("filter_ids", C.c_void_p), ("_owner", C.c_void_p)]


class RetainResult(C.Structure):
    _fields_ = [("n_filters", C.c_uint32), ("n_hits", C.c_uint64), ("status", C.c_void_p),
                ("hit_offsets", C.c_void_p), ("topic_ids", C.c_void_p), ("_owner", C.c_void_p)]


class RetainRanges(C.Structure):      # == rgr_retain_ranges
    _fields_ = [("n_filters", C.c_uint32), ("n_ranges", C.c_uint64), ("n_entries", C.c_uint64), ("status", C.c_void_p),
                ("range_offsets", C.c_void_p), ("ranges", C.c_void_p), ("vals", C.c_void_p * 2), ("n_vals", C.c_uint64 * 2), ("_owner", C.c_void_p)]


RANGE_DTYPE = np.dtype([("begin", np.uint32), ("len", np.uint32)])               # == rgr_id_range
RETAIN_VAL_DTYPE = np.dtype([("topic_id", np.uint32), ("flags", np.uint32)])     # == rgr_retain_val
RGR_RETAIN_HIT_DEAD = 1
RGR_ORDER_CALLER, RGR_ORDER_WALK = 0, 1


class Window(C.Structure):
    _fields_ = [("topic_begin", C.c_uint32), ("topic_end", C.c_uint32), ("n_hits", C.c_uint64),
                ("hit_base", C.c_uint64), ("d_tuples", C.c_void_p), ("d_hit_offsets", C.c_void_p),
                ("offsets_bias", C.c_uint64), ("d_sub_ids", C.c_void_p), ("d_qos", C.c_void_p),
                ("n_runs", C.c_uint64), ("d_run_src", C.c_void_p), ("d_run_topic", C.c_void_p), ("d_run_off", C.c_void_p), ("d_subs", C.c_void_p),
                ("d_ids24", C.c_void_p), ("d_hits8", C.c_void_p), ("d_topic_order", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = ([(n, C.c_uint64) for n in ("n_filters", "n_subs", "n_nodes", "n_edge_slots", "n_tokens", "epoch",
                                           "table_bytes_device", "topics", "invalid_topics", "levels", "pairs", "hits",
                                           "visited_nodes", "overflow_topics", "walk_launches", "expand_launches")] +
                [(n, C.c_double) for n in ("walk_ms", "scan_ms", "expand_ms", "tokenize_ms", "h2d_ms", "d2h_ms")] +
                [(n, C.c_uint64) for n in ("alg_bytes_walk", "alg_bytes_expand", "commits_full", "commits_delta",
                                           "dedup_candidates", "dedup_launches")] +
                [("dedup_ms", C.c_double)] +
                [(n, C.c_uint64) for n in ("retain_epoch", "retain_topics", "retain_delta_topics", "retain_dead", "retain_merges")])

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class RgrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rgr error {code}: {msg}")
        self.code = code


_LIB = None


def lib():
    """Load (building in-tree if stale) librmqtt_gpu_router.so.  Fails loudly if it cannot."""
    global _LIB
    if _LIB is None:
        path = _build.build_gpu()
        L = C.CDLL(path, mode=C.RTLD_GLOBAL)
        vp, u32, u64, i32, u8 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_uint8
        for name in SYMBOLS:
            getattr(L, name)   # AttributeError if the .so misses a declared symbol
        L.rgr_last_error.restype = C.c_char_p
        L.rgr_version.restype = C.c_char_p
        L.rgr_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
        L.rgr_destroy.argtypes = [vp]; L.rgr_destroy.restype = None
        L.rgr_filter_add.argtypes = [vp, C.c_char_p, u32, C.POINTER(u32)]
        L.rgr_filter_find.argtypes = [vp, C.c_char_p, u32, C.POINTER(u32)]
        L.rgr_filter_remove.argtypes = [vp, u32]
        L.rgr_sub_add.argtypes = [vp, u32, u32, u8, u8]
        L.rgr_sub_add_ex.argtypes = [vp, u32, u32, u8, u8, C.c_uint16, u32, u32]
        L.rgr_sub_attrs_bulk.argtypes = [vp, vp, vp, vp, u64]
        L.rgr_match_batch_deliver.argtypes = [vp, vp, vp, u32, vp, C.POINTER(Result)]
        L.rgr_batch_set_publish_attrs.argtypes = [vp, vp]
        L.rgr_sub_remove.argtypes = [vp, u32, u32]
        L.rgr_subscribe_bulk.argtypes = [vp, vp, vp, u64, vp, vp, vp, vp, C.POINTER(u64)]
        L.rgr_commit.argtypes = [vp]
        L.rgr_snapshot_save.argtypes = [vp, C.c_char_p]
        L.rgr_snapshot_load.argtypes = [vp, C.c_char_p]
        L.rgr_match_batch.argtypes = [vp, vp, vp, u32, C.POINTER(Result)]
        L.rgr_result_free.argtypes = [C.POINTER(Result)]; L.rgr_result_free.restype = None
        L.rgr_match_filters.argtypes = [vp, vp, vp, u32, C.POINTER(FiltersResult)]
        L.rgr_filters_result_free.argtypes = [C.POINTER(FiltersResult)]; L.rgr_filters_result_free.restype = None
        L.rgr_batch_create.argtypes = [vp, vp, vp, u32, C.POINTER(vp)]
        L.rgr_batch_destroy.argtypes = [vp]; L.rgr_batch_destroy.restype = None
        L.rgr_batch_create_from_publish.argtypes = [vp, vp, vp, u32, u32, vp, C.POINTER(vp)]
        L.rgr_batch_publish_info.argtypes = [vp]; L.rgr_batch_publish_info.restype = vp
        L.rgr_batch_status.argtypes = [vp]; L.rgr_batch_status.restype = vp
        L.rgr_batch_begin.argtypes = [vp]
        L.rgr_batch_set_format.argtypes = [vp, u32]
        L.rgr_batch_set_topic_ids.argtypes = [vp, vp]
        L.rgr_batch_set_order.argtypes = [vp, u32]
        L.rgr_batch_topic_order.argtypes = [vp]; L.rgr_batch_topic_order.restype = vp
        L.rgr_batch_set_retain_positions.argtypes = [vp, C.c_int32]
        L.rgr_batch_retain_vals.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
        L.rgr_batch_next_window.argtypes = [vp, C.POINTER(Window)]
        L.rgr_window_to_host.argtypes = [vp, C.POINTER(Window), vp, vp]
        L.rgr_batch_run.argtypes = [vp, C.POINTER(u64), C.POINTER(u32)]
        L.rgr_batch_run_to_host.argtypes = [vp, vp, vp, C.POINTER(u64), C.POINTER(u32)]
        L.rgr_retain_topic_add.argtypes = [vp, C.c_char_p, u32, u32]
        L.rgr_retain_topic_remove.argtypes = [vp, C.c_char_p, u32]
        L.rgr_retain_add_bulk.argtypes = [vp, vp, vp, u64, vp, C.POINTER(u64)]
        L.rgr_retain_commit.argtypes = [vp]
        L.rgr_retain_match_batch.argtypes = [vp, vp, vp, u32, C.POINTER(RetainResult)]
        L.rgr_retain_result_free.argtypes = [C.POINTER(RetainResult)]; L.rgr_retain_result_free.restype = None
        L.rgr_retain_match_ranges.argtypes = [vp, vp, vp, u32, C.POINTER(RetainRanges)]
        L.rgr_retain_ranges_free.argtypes = [C.POINTER(RetainRanges)]; L.rgr_retain_ranges_free.restype = None
        L.rgr_retain_batch_create.argtypes = [vp, vp, vp, u32, C.POINTER(vp)]
        L.rgr_retain_batch_create_tier.argtypes = [vp, vp, vp, u32, u32, C.POINTER(vp)]
        L.rgr_shard_assign.argtypes = [vp, vp, u64, u32, i32, u32, vp]
        L.rgr_stats_get.argtypes = [vp, C.POINTER(Stats)]
        L.rgr_stats_reset.argtypes = [vp]
        L.rgr_comm_unique_id.argtypes = [vp]
        L.rgr_comm_create.argtypes = [vp, vp, u32, u32, C.POINTER(vp)]
        L.rgr_comm_destroy.argtypes = [vp]; L.rgr_comm_destroy.restype = None
        L.rgr_comm_allgather_u64.argtypes = [vp, u64, vp]
        L.rgr_comm_info.argtypes = [vp, C.POINTER(CommInfo)]
        L.rgr_comm_gather_pass.argtypes = [vp, vp, GATHER_CONSUMER, vp, C.POINTER(u64), C.POINTER(u64)]
        L.rgr_group_create.argtypes = [C.POINTER(Config), vp, u32, C.POINTER(vp)]
        L.rgr_group_destroy.argtypes = [vp]; L.rgr_group_destroy.restype = None
        L.rgr_group_size.argtypes = [vp]; L.rgr_group_size.restype = u32
        L.rgr_group_handle.argtypes = [vp, u32]; L.rgr_group_handle.restype = vp
        L.rgr_group_comm.argtypes = [vp, u32]; L.rgr_group_comm.restype = vp
        L.rgr_group_uses_rccl.argtypes = [vp]
        L.rgr_group_subscribe_bulk.argtypes = [vp, vp, vp, u64, vp, vp, vp, C.POINTER(u64)]
        L.rgr_group_subscribe.argtypes = [vp, C.c_char_p, u32, u32, u8, u8]
        L.rgr_group_unsubscribe.argtypes = [vp, C.c_char_p, u32, u32, i32]
        L.rgr_group_commit.argtypes = [vp]
        L.rgr_group_retain_topic_add.argtypes = [vp, C.c_char_p, u32, u32]
        L.rgr_group_retain_topic_remove.argtypes = [vp, C.c_char_p, u32]
        L.rgr_group_retain_add_bulk.argtypes = [vp, vp, vp, u64, vp, C.POINTER(u64)]
        L.rgr_group_retain_commit.argtypes = [vp]
        L.rgr_group_retain_match_batch.argtypes = [vp, vp, vp, u32, C.POINTER(RetainResult)]
        L.rgr_group_match_batch.argtypes = [vp, vp, vp, u32, C.POINTER(Result)]
        L.rgr_group_batch_create.argtypes = [vp, vp, vp, u32, C.POINTER(vp)]
        L.rgr_group_batch_destroy.argtypes = [vp]; L.rgr_group_batch_destroy.restype = None
        L.rgr_group_batch_shard.argtypes = [vp, u32]; L.rgr_group_batch_shard.restype = vp
        L.rgr_group_batch_run.argtypes = [vp, vp, C.POINTER(u64)]
        L.rgr_group_batch_gather.argtypes = [vp, u32, GATHER_CONSUMER, vp, C.POINTER(u64)]
        # round 3
        L.rgr_match_filter_subs.argtypes = [vp, vp, vp, u32, C.POINTER(FiltersResult)]
        L.rgr_group_match_filter_subs.argtypes = [vp, vp, vp, u32, C.POINTER(FiltersResult)]
        L.rgr_match_batch_deliver_grouped.argtypes = [vp, vp, vp, u32, vp, C.POINTER(Result), C.POINTER(NodeGroups)]
        L.rgr_group_match_batch_deliver.argtypes = [vp, vp, vp, u32, vp, C.POINTER(Result)]
        L.rgr_group_match_batch_deliver_grouped.argtypes = [vp, vp, vp, u32, vp, C.POINTER(Result), C.POINTER(NodeGroups)]
        L.rgr_group_set_key_levels.argtypes = [vp, u32]
        L.rgr_comm_replicate_subs.argtypes = [vp]
        L.rgr_comm_peer_subs.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(u64)]
        L.rgr_comm_gather_runs_pass.argtypes = [vp, vp, GATHER_CONSUMER, vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
        L.rgr_group_batch_gather_runs.argtypes = [vp, u32, GATHER_CONSUMER, vp, C.POINTER(u64), C.POINTER(u64)]
        L.rgr_group_peer_subs.argtypes = [vp, u32, u32, C.POINTER(vp), C.POINTER(u64)]
        L.rgr_group_subscribe_ex.argtypes = [vp, C.c_char_p, u32, u32, u8, u8, C.c_uint16, u32, u32]
        L.rgr_group_sub_attrs_bulk.argtypes = [vp, vp, vp, vp, u64]
        for name in SYMBOLS:            # a symbol without argtypes would take 64-bit pointers as C ints (r3d: a truncated pointer, SIGSEGV)
            fn = getattr(L, name)
            if fn.argtypes is None and name not in ("rgr_last_error", "rgr_version"):
                raise RuntimeError(f"capi: {name} has no argtypes")
        _LIB = L
    return _LIB


def _check(rc):
    if rc < 0:
        raise RgrError(rc, lib().rgr_last_error().decode())
    return rc


def _b(s):
    return s if isinstance(s, (bytes, bytearray)) else s.encode()


def _blob_ptr(blob):
    if isinstance(blob, np.ndarray):
        return C.c_void_p(blob.ctypes.data), blob
    b = bytes(blob)
    return C.cast(C.c_char_p(b), C.c_void_p), b


def _copy(ptr, n, dtype):
    if not n:
        return np.zeros(0, dtype=dtype)
    out = np.empty(int(n), dtype=dtype)
    C.memmove(out.ctypes.data, ptr, out.nbytes)      # (string_at is limited to 2 GiB)
    return out


def pack(strs):
    bs = [_b(s) for s in strs]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    return np.frombuffer(b"".join(bs), dtype=np.uint8), offs


class Router:
    """One rgr_handle.  Thin: argument marshalling only."""

    def __init__(self, device=0, slot_cap=0, window_hits=0, chunk_topics=0, host_threads=0, collect_walk_stats=True,
                 host_tokenize=False, retain_delta_max=0):
        self._h = C.c_void_p()
        cfg = Config(device, slot_cap, window_hits, chunk_topics, host_threads, int(collect_walk_stats), int(host_tokenize),
                     int(retain_delta_max), 0)
        _check(lib().rgr_create(C.byref(cfg), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().rgr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- table
    def filter_add(self, f):
        f = _b(f); fid = C.c_uint32()
        _check(lib().rgr_filter_add(self._h, f, len(f), C.byref(fid)))
        return fid.value

    def filter_find(self, f):
        f = _b(f); fid = C.c_uint32()
        rc = lib().rgr_filter_find(self._h, f, len(f), C.byref(fid))
        return fid.value if rc == RGR_OK else None

    def filter_remove(self, fid):
        return lib().rgr_filter_remove(self._h, fid)

    def sub_add(self, fid, sub_id, qos=0, flags=0):
        _check(lib().rgr_sub_add(self._h, fid, sub_id, qos, flags))

    def sub_add_ex(self, fid, sub_id, qos=0, flags=0, node_idx=0, owner_id=ID_NONE, client_idx=ID_NONE):
        _check(lib().rgr_sub_add_ex(self._h, fid, sub_id, qos, flags, node_idx, owner_id, client_idx))

    def sub_attrs_bulk(self, owner_ids, client_idx, sub_ids=None):
        o = np.ascontiguousarray(owner_ids, dtype=np.uint32)
        c = np.ascontiguousarray(client_idx, dtype=np.uint32)
        s_ = None if sub_ids is None else np.ascontiguousarray(sub_ids, dtype=np.uint32)
        _check(lib().rgr_sub_attrs_bulk(self._h, None if s_ is None else s_.ctypes.data, o.ctypes.data, c.ctypes.data, len(o)))

    def sub_remove(self, fid, sub_id):
        return lib().rgr_sub_remove(self._h, fid, sub_id)

    def subscribe_bulk(self, blob, offsets, sub_ids=None, qos=None, flags=None, want_filter_ids=False):
        n = len(offsets) - 1
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        keep = [offsets]
        def p(a, dt):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=dt); keep.append(a)
            return C.c_void_p(a.ctypes.data)
        fids = np.zeros(n, dtype=np.uint32) if want_filter_ids else None
        rej = C.c_uint64(0)
        bp, bk = _blob_ptr(blob)
        _check(lib().rgr_subscribe_bulk(self._h, bp, offsets.ctypes.data, n, p(sub_ids, np.uint32), p(qos, np.uint8),
                                        p(flags, np.uint8), C.c_void_p(fids.ctypes.data) if want_filter_ids else None,
                                        C.byref(rej)))
        return (int(rej.value), fids) if want_filter_ids else int(rej.value)

    def commit(self):
        _check(lib().rgr_commit(self._h))

    def snapshot_save(self, path):
        _check(lib().rgr_snapshot_save(self._h, os.fsencode(path)))

    def snapshot_load(self, path):
        """Replace the host table by a snapshot file; raises RgrError (table untouched) on a bad file."""
        _check(lib().rgr_snapshot_load(self._h, os.fsencode(path)))

    # ---- matching
    def match_batch(self, blob, offsets):
        """-> dict(status int32[n], hit_offsets uint64[n+1], tuples TUPLE_DTYPE[n_hits])"""
        n = len(offsets) - 1
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        r = Result()
        bp, bk = _blob_ptr(blob)
        _check(lib().rgr_match_batch(self._h, bp, offsets.ctypes.data, n, C.byref(r)))
        try:
            return dict(status=_copy(r.status, n, np.int32), hit_offsets=_copy(r.hit_offsets, n + 1, np.uint64),
                        tuples=_copy(r.tuples, r.n_hits, TUPLE_DTYPE))
        finally:
            lib().rgr_result_free(C.byref(r))

    def match_batch_deliver(self, blob, offsets, publish_attrs, grouped=False):
        """match_batch with the delivery stage: tuples carry delivery words (RGR_HIT_*).
        publish_attrs: PUBLISH_ATTR_DTYPE[n] (from_id, qos_retain).  grouped: every topic's tuples partitioned by node on the
        device (rgr_match_batch_deliver_grouped) + the node directory (group_offsets / group_node / group_begin)."""
        n = len(offsets) - 1
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        pa = np.ascontiguousarray(publish_attrs, dtype=PUBLISH_ATTR_DTYPE)
        assert len(pa) == n
        r = Result()
        g = NodeGroups()
        bp, bk = _blob_ptr(blob)
        if grouped:
            _check(lib().rgr_match_batch_deliver_grouped(self._h, bp, offsets.ctypes.data, n, pa.ctypes.data, C.byref(r), C.byref(g)))
        else:
            _check(lib().rgr_match_batch_deliver(self._h, bp, offsets.ctypes.data, n, pa.ctypes.data, C.byref(r)))
        try:
            out = dict(status=_copy(r.status, n, np.int32), hit_offsets=_copy(r.hit_offsets, n + 1, np.uint64),
                       tuples=_copy(r.tuples, r.n_hits, TUPLE_DTYPE))
            if grouped:
                out.update(group_offsets=_copy(g.group_offsets, n + 1, np.uint64), group_node=_copy(g.group_node, g.n_groups, np.uint32),
                           group_begin=_copy(g.group_begin, g.n_groups + 1, np.uint64))
            return out
        finally:
            lib().rgr_result_free(C.byref(r))

    def match_filters(self, blob, offsets, first_subs=False):
        """Matched filters per topic, in TopicTree::matches order: the library's filter ids, or (first_subs) the sub id of
        each filter's first subscriber (ID_NONE when it has none) — rgr_match_filter_subs."""
        n = len(offsets) - 1
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        r = FiltersResult()
        bp, bk = _blob_ptr(blob)
        _check((lib().rgr_match_filter_subs if first_subs else lib().rgr_match_filters)(self._h, bp, offsets.ctypes.data, n, C.byref(r)))
        try:
            return dict(status=_copy(r.status, n, np.int32), pair_offsets=_copy(r.pair_offsets, n + 1, np.uint64),
                        filter_ids=_copy(r.filter_ids, r.n_pairs, np.uint32))
        finally:
            lib().rgr_filters_result_free(C.byref(r))

    def batch(self, blob, offsets):
        return Batch(self, blob, offsets)

    def publish_batch(self, packets, offsets, version=4, from_ids=None):
        """Batch built from raw MQTT PUBLISH packets (one frame per entry)."""
        return Batch(self, packets, offsets, publish_version=version, from_ids=from_ids)

    def retain_batch(self, blob, offsets, tier=None):
        return Batch(self, blob, offsets, retain=True, tier=tier)

    # ---- retain twin
    def retain_add(self, topic, topic_id):
        t = _b(topic)
        return lib().rgr_retain_topic_add(self._h, t, len(t), topic_id)

    def retain_remove(self, topic):
        t = _b(topic)
        return lib().rgr_retain_topic_remove(self._h, t, len(t))

    def retain_add_bulk(self, blob, offsets, topic_ids=None):
        n = len(offsets) - 1
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        ids = None if topic_ids is None else np.ascontiguousarray(topic_ids, dtype=np.uint32)
        rej = C.c_uint64(0)
        bp, bk = _blob_ptr(blob)
        _check(lib().rgr_retain_add_bulk(self._h, bp, offsets.ctypes.data, n,
                                         None if ids is None else C.c_void_p(ids.ctypes.data), C.byref(rej)))
        return int(rej.value)

    def retain_commit(self):
        _check(lib().rgr_retain_commit(self._h))

    def retain_match_batch(self, blob, offsets):
        n = len(offsets) - 1
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        r = RetainResult()
        bp, bk = _blob_ptr(blob)
        _check(lib().rgr_retain_match_batch(self._h, bp, offsets.ctypes.data, n, C.byref(r)))
        try:
            return dict(status=_copy(r.status, n, np.int32), hit_offsets=_copy(r.hit_offsets, n + 1, np.uint64),
                        topic_ids=_copy(r.topic_ids, r.n_hits, np.uint32))
        finally:
            lib().rgr_retain_result_free(C.byref(r))

    def retain_match_ranges(self, blob, offsets, flatten=True):
        """rgr_retain_match_ranges: per filter the matched retained topics as ranges of the host-mirrored value array.
        -> dict(status, range_offsets, ranges, n_entries[, hit_offsets, topic_ids: the ranges resolved through the mirror, dead entries
        dropped — what rgr_retain_match_batch returns])."""
        n = len(offsets) - 1
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        r = RetainRanges()
        bp, bk = _blob_ptr(blob)
        _check(lib().rgr_retain_match_ranges(self._h, bp, offsets.ctypes.data, n, C.byref(r)))
        try:
            out = dict(status=_copy(r.status, n, np.int32), range_offsets=_copy(r.range_offsets, n + 1, np.uint64),
                       ranges=_copy(r.ranges, r.n_ranges, RANGE_DTYPE), n_entries=int(r.n_entries), n_ranges=int(r.n_ranges))
            if flatten:
                vals = [np.ctypeslib.as_array(C.cast(r.vals[t], C.POINTER(C.c_uint32)), shape=(int(r.n_vals[t]) * 2,)).reshape(-1, 2)
                        if r.vals[t] else np.zeros((0, 2), dtype=np.uint32) for t in range(2)]
                rg = out["ranges"]
                ln = (rg["len"] & 0x7FFFFFFF).astype(np.int64)
                tier = (rg["len"] >> 31).astype(np.int64)
                ids, cnt = [], np.zeros(n, dtype=np.int64)
                ro = out["range_offsets"].astype(np.int64)
                for i in range(n):
                    for k in range(ro[i], ro[i + 1]):
                        v = vals[tier[k]][int(rg["begin"][k]):int(rg["begin"][k]) + int(ln[k])]
                        live = v[(v[:, 1] & RGR_RETAIN_HIT_DEAD) == 0, 0]
                        ids.append(live.copy()); cnt[i] += len(live)
                out["topic_ids"] = np.concatenate(ids) if ids else np.zeros(0, dtype=np.uint32)
                out["hit_offsets"] = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
            return out
        finally:
            lib().rgr_retain_ranges_free(C.byref(r))

    # ---- stats
    def stats(self):
        s = Stats()
        _check(lib().rgr_stats_get(self._h, C.byref(s)))
        return s.as_dict()

    def stats_reset(self):
        _check(lib().rgr_stats_reset(self._h))


class Batch:
    """Device-resident tokenised batch (rgr_batch_*)."""

    def __init__(self, router, blob, offsets, retain=False, tier=None, publish_version=None, from_ids=None):
        self.router = router
        self.n = len(offsets) - 1
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._b = C.c_void_p()
        bp, bk = _blob_ptr(blob)
        if publish_version is not None:      # blob = raw MQTT PUBLISH packets (rgr_batch_create_from_publish)
            f = None if from_ids is None else np.ascontiguousarray(from_ids, dtype=np.uint32)
            _check(lib().rgr_batch_create_from_publish(router._h, bp, offsets.ctypes.data, self.n, publish_version,
                                                       None if f is None else f.ctypes.data, C.byref(self._b)))
            return
        if retain and tier is not None:
            _check(lib().rgr_retain_batch_create_tier(router._h, bp, offsets.ctypes.data, self.n, tier, C.byref(self._b)))
            return
        create = lib().rgr_retain_batch_create if retain else lib().rgr_batch_create
        _check(create(router._h, bp, offsets.ctypes.data, self.n, C.byref(self._b)))

    def close(self):
        if self._b:
            lib().rgr_batch_destroy(self._b)
            self._b = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def status(self):
        return _copy(lib().rgr_batch_status(self._b), self.n, np.int32)

    def publish_info(self):
        p = lib().rgr_batch_publish_info(self._b)
        return _copy(p, self.n, PUBLISH_INFO_DTYPE) if p else None

    def set_publish_attrs(self, publish_attrs):
        if publish_attrs is None:
            _check(lib().rgr_batch_set_publish_attrs(self._b, None))
            return
        pa = np.ascontiguousarray(publish_attrs, dtype=PUBLISH_ATTR_DTYPE)
        assert len(pa) == self.n
        _check(lib().rgr_batch_set_publish_attrs(self._b, pa.ctypes.data))

    def set_topic_ids(self, ids):
        """Tuples carry ids[i] instead of the batch index i (None restores it)."""
        if ids is None:
            _check(lib().rgr_batch_set_topic_ids(self._b, None)); return
        a = np.ascontiguousarray(ids, dtype=np.uint32)
        assert len(a) == self.n
        _check(lib().rgr_batch_set_topic_ids(self._b, a.ctypes.data))

    def set_format(self, fmt):
        """RGR_FORMAT_TUPLE (12 B/hit) | RGR_FORMAT_SOA (sub ids + qos bytes, 5 B/hit) | RGR_FORMAT_PACKED (4 B/hit) | RGR_FORMAT_RUNS |
        RGR_FORMAT_IDS24 (3-byte sub ids, 3 B/hit) | RGR_FORMAT_DELIVER8 (delivery passes: {sub_id, delivery word}, 8 B/hit)."""
        _check(lib().rgr_batch_set_format(self._b, fmt))

    def set_order(self, walk=True):
        """RGR_ORDER_WALK: later passes walk the topics sorted by their leading tokens (windows then enumerate walk positions: Window.d_topic_order)."""
        _check(lib().rgr_batch_set_order(self._b, RGR_ORDER_WALK if walk else RGR_ORDER_CALLER))

    def topic_order(self):
        """The permutation of a batch in walk order: order[k] = batch index of the topic walked k-th; None in caller order."""
        p = lib().rgr_batch_topic_order(self._b)
        return _copy(p, self.n, np.uint32) if p else None

    def set_retain_positions(self, on=True):
        """Retain batches: tuples carry positions in the epoch's preorder value array instead of topic ids (see retain_vals)."""
        _check(lib().rgr_batch_set_retain_positions(self._b, 1 if on else 0))

    def retain_vals(self):
        """Host mirror of the value array of the epoch the last begin() bound: structured array (topic_id, flags), a copy."""
        p, n = C.c_void_p(), C.c_uint64(0)
        _check(lib().rgr_batch_retain_vals(self._b, C.byref(p), C.byref(n)))
        return _copy(p, int(n.value), RETAIN_VAL_DTYPE) if n.value else np.zeros(0, dtype=RETAIN_VAL_DTYPE)

    def run(self):
        """One full pass; tuples stay on the device.  -> (n_hits, n_windows)"""
        h, w = C.c_uint64(0), C.c_uint32(0)
        _check(lib().rgr_batch_run(self._b, C.byref(h), C.byref(w)))
        return int(h.value), int(w.value)

    def run_to_host(self):
        """One full pass streaming every window into pinned host staging (PCIe-inclusive).  -> (n_hits, n_windows)"""
        h, w = C.c_uint64(0), C.c_uint32(0)
        _check(lib().rgr_batch_run_to_host(self._b, None, None, C.byref(h), C.byref(w)))
        return int(h.value), int(w.value)

    def begin(self):
        _check(lib().rgr_batch_begin(self._b))

    def next_window(self):
        w = Window()
        rc = _check(lib().rgr_batch_next_window(self._b, C.byref(w)))
        return None if rc == RGR_EOF else w

    def window_to_host(self, w):
        tuples = np.zeros(int(w.n_hits), dtype=TUPLE_DTYPE)
        offs = np.zeros(w.topic_end - w.topic_begin + 1, dtype=np.uint64)
        _check(lib().rgr_window_to_host(self._b, C.byref(w), tuples.ctypes.data if w.n_hits else None, offs.ctypes.data))
        return tuples, offs


_HIP = None


def device_to_host(ptr, nbytes):
    """Copy library-owned device memory to a numpy uint8 array (tests / bench checks only)."""
    global _HIP
    if _HIP is None:
        _HIP = C.CDLL("libamdhip64.so")
        _HIP.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    out = np.empty(int(nbytes), dtype=np.uint8)
    if nbytes:
        rc = _HIP.hipMemcpy(out.ctypes.data, C.c_void_p(int(ptr)), int(nbytes), 2)     # hipMemcpyDeviceToHost
        if rc != 0:
            raise RgrError(RGR_EDEVICE, f"hipMemcpy D2H failed ({rc})")
    return out


class Comm:
    """One rank's RCCL communicator on a Router's device (rgr_comm_*): one rank per process."""

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * RGR_COMM_ID_BYTES)()
        _check(lib().rgr_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, router, uid, rank, world):
        self.router, self.rank, self.world = router, rank, world
        self._c = C.c_void_p()
        uid = bytes(uid)
        assert len(uid) == RGR_COMM_ID_BYTES
        _check(lib().rgr_comm_create(router._h, uid, rank, world, C.byref(self._c)))

    def close(self):
        if self._c:
            lib().rgr_comm_destroy(self._c)
            self._c = C.c_void_p()

    def info(self):
        """What the transport reports: {"ranks": ncclCommCount, "rank": ncclCommUserRank, "device", "transport": "rccl" | "device copies"}."""
        ci = CommInfo()
        _check(lib().rgr_comm_info(self._c, C.byref(ci)))
        return {"ranks": int(ci.ranks), "rank": int(ci.rank), "device": int(ci.device), "transport": "rccl" if ci.transport == 1 else "device copies"}

    def allgather_u64(self, mine):
        out = np.zeros(self.world, dtype=np.uint64)
        _check(lib().rgr_comm_allgather_u64(self._c, int(mine), out.ctypes.data))
        return out

    def gather_pass(self, batch, collect=False):
        """All-gathered pass of `batch`.  -> (my_hits, all_hits, tuples | None): with collect, every round's
        gathered device buffer is copied to the host and concatenated (tests)."""
        parts = []

        def cb(user, d_tuples, counts, world, n_total):
            if collect and n_total:
                parts.append(device_to_host(d_tuples, int(n_total) * 12).view(TUPLE_DTYPE))
        fn = GATHER_CONSUMER(cb)
        mine, allh = C.c_uint64(0), C.c_uint64(0)
        _check(lib().rgr_comm_gather_pass(self._c, batch._b, fn, None, C.byref(mine), C.byref(allh)))
        tup = (np.concatenate(parts) if parts else np.zeros(0, dtype=TUPLE_DTYPE)) if collect else None
        return int(mine.value), int(allh.value), tup

    def replicate_subs(self):
        """Collective: every rank's subs[] on every rank (call after commit, before gather_runs_pass)."""
        _check(lib().rgr_comm_replicate_subs(self._c))

    def peer_subs(self, rank):
        p, n = C.c_void_p(), C.c_uint64(0)
        _check(lib().rgr_comm_peer_subs(self._c, rank, C.byref(p), C.byref(n)))
        return device_to_host(p.value, int(n.value) * 8).view(SUB_ENTRY_DTYPE) if n.value else np.zeros(0, dtype=SUB_ENTRY_DTYPE)

    def gather_runs_pass(self, batch, collect=False):
        """All-gathered pass with run descriptors as the payload.  -> (my_runs, all_runs, all_hits, descriptors | None)"""
        parts = []

        def cb(user, d_runs, counts, world, n_total):
            if collect and n_total:
                parts.append(device_to_host(d_runs, int(n_total) * 16).view(RUN_DTYPE))
        fn = GATHER_CONSUMER(cb)
        mine, allr, allh = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _check(lib().rgr_comm_gather_runs_pass(self._c, batch._b, fn, None, C.byref(mine), C.byref(allr), C.byref(allh)))
        got = (np.concatenate(parts) if parts else np.zeros(0, dtype=RUN_DTYPE)) if collect else None
        return int(mine.value), int(allr.value), int(allh.value), got


class Group:
    """Single-process multi-device router (rgr_group_*): one shard per entry of `devices`."""

    def __init__(self, devices, slot_cap=0, window_hits=0, chunk_topics=0, host_threads=0):
        cfg = Config(0, slot_cap, window_hits, chunk_topics, host_threads, 1, 0, 0, 0)
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        self._g = C.c_void_p()
        _check(lib().rgr_group_create(C.byref(cfg), devs.ctypes.data, len(devs), C.byref(self._g)))
        self.size = len(devs)

    def close(self):
        if self._g:
            lib().rgr_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def uses_rccl(self):
        return bool(lib().rgr_group_uses_rccl(self._g))

    def set_key_levels(self, k):
        """Leading topic levels hashed into the shard key (default 3); only while the group is empty."""
        _check(lib().rgr_group_set_key_levels(self._g, k))

    def shard_stats(self, shard):
        s = Stats()
        _check(lib().rgr_stats_get(lib().rgr_group_handle(self._g, shard), C.byref(s)))
        return s.as_dict()

    def subscribe_bulk(self, blob, offsets, sub_ids=None, qos=None, flags=None):
        n = len(offsets) - 1
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        keep = []

        def p(a, dt):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=dt); keep.append(a)
            return C.c_void_p(a.ctypes.data)
        rej = C.c_uint64(0)
        bp, bk = _blob_ptr(blob)
        _check(lib().rgr_group_subscribe_bulk(self._g, bp, offsets.ctypes.data, n, p(sub_ids, np.uint32), p(qos, np.uint8), p(flags, np.uint8), C.byref(rej)))
        return int(rej.value)

    def subscribe(self, f, sub_id, qos=0, flags=0):
        f = _b(f)
        _check(lib().rgr_group_subscribe(self._g, f, len(f), sub_id, qos, flags))

    def unsubscribe(self, f, sub_id, last_of_filter=False):
        f = _b(f)
        _check(lib().rgr_group_unsubscribe(self._g, f, len(f), sub_id, int(last_of_filter)))

    def commit(self):
        _check(lib().rgr_group_commit(self._g))

    # ---- the retained-message twin over the group (rgr_group_retain_*)
    def retain_add(self, topic, topic_id):
        t = _b(topic)
        return lib().rgr_group_retain_topic_add(self._g, t, len(t), topic_id)

    def retain_remove(self, topic):
        t = _b(topic)
        return lib().rgr_group_retain_topic_remove(self._g, t, len(t))

    def retain_add_bulk(self, blob, offsets, topic_ids=None):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        ids = None if topic_ids is None else np.ascontiguousarray(topic_ids, dtype=np.uint32)
        rej = C.c_uint64(0)
        bp, bk = _blob_ptr(blob)
        _check(lib().rgr_group_retain_add_bulk(self._g, bp, offsets.ctypes.data, len(offsets) - 1, None if ids is None else ids.ctypes.data, C.byref(rej)))
        return int(rej.value)

    def retain_commit(self):
        _check(lib().rgr_group_retain_commit(self._g))

    def retain_match_batch(self, blob, offsets):
        n = len(offsets) - 1
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        r = RetainResult()
        bp, bk = _blob_ptr(blob)
        _check(lib().rgr_group_retain_match_batch(self._g, bp, offsets.ctypes.data, n, C.byref(r)))
        try:
            return dict(status=_copy(r.status, n, np.int32), hit_offsets=_copy(r.hit_offsets, n + 1, np.uint64),
                        topic_ids=_copy(r.topic_ids, r.n_hits, np.uint32))
        finally:
            lib().rgr_retain_result_free(C.byref(r))

    def match_batch(self, blob, offsets):
        n = len(offsets) - 1
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        r = Result()
        bp, bk = _blob_ptr(blob)
        _check(lib().rgr_group_match_batch(self._g, bp, offsets.ctypes.data, n, C.byref(r)))
        try:
            return dict(status=_copy(r.status, n, np.int32), hit_offsets=_copy(r.hit_offsets, n + 1, np.uint64),
                        tuples=_copy(r.tuples, r.n_hits, TUPLE_DTYPE))
        finally:
            lib().rgr_result_free(C.byref(r))

    def peer_subs(self, holder, of):
        """Shard `holder`'s replica of shard `of`'s subscriber entries (after a gather_runs pass), copied to the host."""
        p, n = C.c_void_p(), C.c_uint64(0)
        _check(lib().rgr_group_peer_subs(self._g, holder, of, C.byref(p), C.byref(n)))
        return device_to_host(p.value, int(n.value) * 8).view(SUB_ENTRY_DTYPE) if n.value else np.zeros(0, dtype=SUB_ENTRY_DTYPE)

    def match_filter_subs(self, blob, offsets):
        """rgr_group_match_filter_subs: per topic, the sub id of each matched filter's first subscriber, in the caller's topic order."""
        n = len(offsets) - 1
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        r = FiltersResult()
        bp, bk = _blob_ptr(blob)
        _check(lib().rgr_group_match_filter_subs(self._g, bp, offsets.ctypes.data, n, C.byref(r)))
        try:
            return dict(status=_copy(r.status, n, np.int32), pair_offsets=_copy(r.pair_offsets, n + 1, np.uint64),
                        filter_ids=_copy(r.filter_ids, r.n_pairs, np.uint32))
        finally:
            lib().rgr_filters_result_free(C.byref(r))

    def batch(self, blob, offsets):
        return GroupBatch(self, blob, offsets)


class GroupBatch:
    def __init__(self, group, blob, offsets):
        self.group = group
        self.n = len(offsets) - 1
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._b = C.c_void_p()
        bp, bk = _blob_ptr(blob)
        _check(lib().rgr_group_batch_create(group._g, bp, offsets.ctypes.data, self.n, C.byref(self._b)))

    def close(self):
        if self._b:
            lib().rgr_group_batch_destroy(self._b)
            self._b = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self):
        """-> (per-shard hits uint64[size], total)"""
        sh = np.zeros(self.group.size, dtype=np.uint64)
        tot = C.c_uint64(0)
        _check(lib().rgr_group_batch_run(self._b, sh.ctypes.data, C.byref(tot)))
        return sh, int(tot.value)

    def gather(self, consumer_shard=0, collect=False):
        """All-gathered pass.  -> (total hits, tuples gathered on `consumer_shard` | None)"""
        parts = []

        def cb(user, d_tuples, counts, world, n_total):
            if collect and n_total:
                parts.append(device_to_host(d_tuples, int(n_total) * 12).view(TUPLE_DTYPE))
        fn = GATHER_CONSUMER(cb)
        tot = C.c_uint64(0)
        _check(lib().rgr_group_batch_gather(self._b, consumer_shard, fn, None, C.byref(tot)))
        tup = (np.concatenate(parts) if parts else np.zeros(0, dtype=TUPLE_DTYPE)) if collect else None
        return int(tot.value), tup

    def gather_runs(self, consumer_shard=0, collect=False):
        """All-gathered pass with RUN DESCRIPTORS as the payload (rgr_group_batch_gather_runs).
        -> (total runs, total hits, descriptors gathered on `consumer_shard` (RUN_DTYPE) | None)"""
        parts = []

        def cb(user, d_runs, counts, world, n_total):
            if collect and n_total:
                parts.append(device_to_host(d_runs, int(n_total) * 16).view(RUN_DTYPE))
        fn = GATHER_CONSUMER(cb)
        runs, hits = C.c_uint64(0), C.c_uint64(0)
        _check(lib().rgr_group_batch_gather_runs(self._b, consumer_shard, fn, None, C.byref(runs), C.byref(hits)))
        got = (np.concatenate(parts) if parts else np.zeros(0, dtype=RUN_DTYPE)) if collect else None
        return int(runs.value), int(hits.value), got
