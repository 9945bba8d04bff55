"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY — see oracle.hpp).

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may
import this module.  The product package ``rmqtt_amd`` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("oracle.cpp", "oracle.hpp")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return so


class OrcId(C.Structure):
    _fields_ = [("node_id", C.c_uint64), ("client_id", C.c_char_p), ("client_len", C.c_uint32),
                ("create_time", C.c_int64), ("lid", C.c_uint16)]


class OrcOpts(C.Structure):
    _fields_ = [("v5", C.c_uint8), ("qos", C.c_uint8), ("no_local", C.c_uint8), ("retain_as_published", C.c_uint8),
                ("retain_handling", C.c_uint8), ("sub_ident", C.c_uint32), ("shared_group", C.c_char_p), ("shared_group_len", C.c_uint32)]


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("levels", "visited", "matched", "hits", "invalid")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, cp, u64, i64 = C.c_void_p, C.c_char_p, C.c_uint64, C.c_int64
        L.orc_free.argtypes = [vp]
        L.orc_parse_topic.argtypes = [cp, u64, C.POINTER(C.c_uint8), C.c_int]
        L.orc_tree_new.restype = vp
        L.orc_tree_free.argtypes = [vp]
        for f in (L.orc_tree_insert, L.orc_tree_remove):
            f.argtypes = [vp, cp, u64, u64]
        L.orc_tree_values_size.argtypes = [vp]; L.orc_tree_values_size.restype = u64
        L.orc_tree_nodes_size.argtypes = [vp]; L.orc_tree_nodes_size.restype = u64
        L.orc_tree_matches.argtypes = [vp, cp, u64]; L.orc_tree_matches.restype = vp
        L.orc_tree_is_match.argtypes = [vp, cp, u64]
        L.orc_retain_new.restype = vp
        L.orc_retain_free.argtypes = [vp]
        L.orc_retain_insert.argtypes = [vp, cp, u64, i64]
        L.orc_retain_remove.argtypes = [vp, cp, u64, C.POINTER(i64)]
        L.orc_retain_retain_ge.argtypes = [vp, u64, i64]; L.orc_retain_retain_ge.restype = u64
        L.orc_retain_values_size.argtypes = [vp]; L.orc_retain_values_size.restype = u64
        L.orc_retain_nodes_size.argtypes = [vp]; L.orc_retain_nodes_size.restype = u64
        L.orc_retain_matches.argtypes = [vp, cp, u64]; L.orc_retain_matches.restype = vp
        L.orc_retain_match_batch.argtypes = [vp, vp, vp, u64, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]
        L.orc_retain_match_batch.restype = u64
        L.orc_retain_insert_bulk.argtypes = [vp, vp, vp, u64, vp]; L.orc_retain_insert_bulk.restype = u64
        L.orc_retain_match_timed.argtypes = [vp, vp, vp, u64, C.c_int, C.POINTER(u64), C.POINTER(u64)]
        L.orc_retain_match_timed.restype = C.c_double
        L.orc_router_new.restype = vp
        L.orc_router_set_shared_policy.argtypes = [vp, C.c_int]; L.orc_router_set_shared_policy.restype = None
        L.orc_router_free.argtypes = [vp]
        L.orc_router_add.argtypes = [vp, cp, u64, C.POINTER(OrcId), C.POINTER(OrcOpts), C.c_uint32]
        L.orc_router_remove.argtypes = [vp, cp, u64, C.POINTER(OrcId)]
        L.orc_router_topics.argtypes = [vp]; L.orc_router_topics.restype = i64
        L.orc_router_routes.argtypes = [vp]; L.orc_router_routes.restype = i64
        L.orc_router_topics_tree.argtypes = [vp]; L.orc_router_topics_tree.restype = u64
        L.orc_router_filter_id.argtypes = [vp, cp, u64]; L.orc_router_filter_id.restype = C.c_uint32
        L.orc_router_add_bulk.argtypes = [vp, vp, vp, u64, vp, vp]
        L.orc_router_matches.argtypes = [vp, C.POINTER(OrcId), cp, u64]; L.orc_router_matches.restype = vp
        L.orc_router_forwards.argtypes = [vp, C.POINTER(OrcId), cp, u64, C.c_uint8, C.c_uint8]; L.orc_router_forwards.restype = vp
        L.orc_router_match_flat.argtypes = [vp, vp, vp, u64, vp] + [C.POINTER(vp)] * 5 + [C.POINTER(OrcStats)]
        L.orc_router_match_flat.restype = u64
        L.orc_router_match_timed.argtypes = [vp, vp, vp, u64, C.c_int, C.POINTER(OrcStats)]
        L.orc_router_match_timed.restype = C.c_double
        L.orc_router_matches_timed.argtypes = [vp, vp, vp, u64, C.c_int, C.c_int, C.POINTER(OrcStats)]
        L.orc_router_matches_timed.restype = C.c_double
        L.orc_retain_match_timed_dyn.argtypes = [vp, vp, vp, u64, C.c_int, C.POINTER(u64), C.POINTER(u64)]
        L.orc_retain_match_timed_dyn.restype = C.c_double
        L.orc_router_match_digest.argtypes = [vp, vp, vp, u64, C.c_int, vp, vp]; L.orc_router_match_digest.restype = None
        L.orc_retain_match_digest.argtypes = [vp, vp, vp, u64, C.c_int, vp, vp]; L.orc_retain_match_digest.restype = None
        L.orc_router_add_bulk_ex.argtypes = [vp, vp, vp, u64, vp, vp, vp]
        L.orc_router_deliver_digest.argtypes = [vp, vp, vp, u64, vp, vp, C.c_int, vp, vp]; L.orc_router_deliver_digest.restype = None
        L.orc_router_forwards_timed.argtypes = [vp, vp, vp, u64, vp, vp, C.c_int, C.POINTER(OrcStats), C.POINTER(u64)]
        L.orc_router_forwards_timed.restype = C.c_double
        L.orc_router_match_digest_fast.argtypes = [vp, vp, vp, u64, C.c_int, vp, vp]; L.orc_router_match_digest_fast.restype = None
        L.orc_retain_match_digest_fast.argtypes = [vp, vp, vp, u64, C.c_int, vp, vp]; L.orc_retain_match_digest_fast.restype = None
        _LIB = L
    return _LIB


def _b(s):
    return s if isinstance(s, bytes) else s.encode()


def _take_str(p):
    if not p:
        return None
    s = C.string_at(p).decode()
    lib().orc_free(p)
    return s


def _take_arr(p, n, dtype):
    a = np.ctypeslib.as_array(C.cast(p, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(max(n, 1),))[:n].copy()
    lib().orc_free(p)
    return a


def pack_strings(strs):
    """list[str|bytes] -> (blob bytes, offsets uint64[n+1])"""
    bs = [_b(s) for s in strs]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    return b"".join(bs), offs


def parse_topic(s):
    s = _b(s)
    kinds = (C.c_uint8 * 64)()
    n = lib().orc_parse_topic(s, len(s), kinds, 64)
    return None if n < 0 else [int(kinds[i]) for i in range(min(n, 64))]


class TopicTree:
    """TopicTree<u64> (rmqtt/src/trie.rs) — golden-vector harness."""

    def __init__(self):
        self._h = lib().orc_tree_new()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_tree_free(self._h); self._h = None

    def insert(self, f, v):
        f = _b(f); return lib().orc_tree_insert(self._h, f, len(f), v)

    def remove(self, f, v):
        f = _b(f); return lib().orc_tree_remove(self._h, f, len(f), v)

    def values_size(self):
        return int(lib().orc_tree_values_size(self._h))

    def nodes_size(self):
        return int(lib().orc_tree_nodes_size(self._h))

    def is_match(self, t):
        t = _b(t); return lib().orc_tree_is_match(self._h, t, len(t))

    def matches(self, t):
        """-> list[(filter, [values])] in iterator order, or None on Err."""
        t = _b(t)
        s = _take_str(lib().orc_tree_matches(self._h, t, len(t)))
        if s is None:
            return None
        out = []
        for line in s.split("\n")[:-1]:
            f, vs = line.split("\t")
            out.append((f, [int(x) for x in vs.split(",")] if vs else []))
        return out


class RetainTree:
    """RetainTree<i64> (rmqtt/src/retain.rs)."""

    def __init__(self):
        self._h = lib().orc_retain_new()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_retain_free(self._h); self._h = None

    def insert(self, t, v):
        t = _b(t); return lib().orc_retain_insert(self._h, t, len(t), v)

    def remove(self, t):
        t = _b(t); v = C.c_int64(0)
        rc = lib().orc_retain_remove(self._h, t, len(t), C.byref(v))
        return (rc, int(v.value) if rc == 1 else None)

    def retain_ge(self, keep_from, max_limit=2**64 - 1):
        return int(lib().orc_retain_retain_ge(self._h, max_limit, keep_from))

    def values_size(self):
        return int(lib().orc_retain_values_size(self._h))

    def nodes_size(self):
        return int(lib().orc_retain_nodes_size(self._h))

    def matches(self, f):
        """-> sorted list[(topic, value)] or None on Err."""
        f = _b(f)
        s = _take_str(lib().orc_retain_matches(self._h, f, len(f)))
        if s is None:
            return None
        out = []
        for line in s.split("\n")[:-1]:
            t, v = line.rsplit("\t", 1)
            out.append((t, int(v)))
        return out

    def insert_bulk(self, blob, offsets, ids=None):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        ids = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint32)
        return int(lib().orc_retain_insert_bulk(self._h, _ptr(blob), offsets.ctypes.data, len(offsets) - 1,
                                                None if ids is None else C.c_void_p(ids.ctypes.data)))

    def match_timed(self, blob, offsets, threads=1, dynamic=True):
        """Timed RetainTree::matches over a batch on `threads` threads (cpu_baseline); `dynamic`: filters are
        handed out one at a time from an atomic cursor instead of a static partition."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        h, v = C.c_uint64(0), C.c_uint64(0)
        fn = lib().orc_retain_match_timed_dyn if dynamic else lib().orc_retain_match_timed
        sec = fn(self._h, _ptr(blob), offsets.ctypes.data, len(offsets) - 1, threads, C.byref(h), C.byref(v))
        return float(sec), dict(hits=int(h.value), visited=int(v.value))

    def match_digest(self, blob, offsets, threads=1, fast=False):
        """-> (status int32[n], digest uint64[n,3] = hits, sum of ids, sum of id^2 per filter).  fast: the same digests through
        bottom-up subtree aggregates for the '#' step (orc_retain_match_digest_fast) — O(nodes walked outside a '#') per filter."""
        n = len(offsets) - 1
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        status = np.zeros(n, dtype=np.int32)
        out = np.zeros((n, 3), dtype=np.uint64)
        (lib().orc_retain_match_digest_fast if fast else lib().orc_retain_match_digest)(self._h, _ptr(blob), offsets.ctypes.data, n, threads, status.ctypes.data, out.ctypes.data)
        return status, out

    def match_batch(self, blob, offsets):
        n = len(offsets) - 1
        status = np.zeros(n, dtype=np.int32)
        po, pv = C.c_void_p(), C.c_void_p()
        visited = C.c_uint64(0)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        tot = lib().orc_retain_match_batch(self._h, _ptr(blob), offsets.ctypes.data, n, status.ctypes.data,
                                           C.byref(po), C.byref(pv), C.byref(visited))
        return status, _take_arr(po, n + 1, np.uint64), _take_arr(pv, tot, np.int64), int(visited.value)


def mk_id(node_id=1, client_id="c", create_time=0, lid=0):
    c = _b(client_id)
    i = OrcId(node_id, c, len(c), create_time, lid)
    i._keep = c
    return i


def mk_opts(qos=0, v5=False, no_local=False, sub_ident=0, rap=False, rh=0, shared_group=None):
    g = _b(shared_group) if shared_group else None
    o = OrcOpts(int(v5), qos, int(no_local), int(rap), rh, sub_ident, g, len(g) if g else 0)
    o._keep = g
    return o


class DefaultRouter:
    """Restated DefaultRouter (rmqtt/src/router.rs:121-265, 434-496)."""

    def __init__(self):
        self._h = lib().orc_router_new()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_router_free(self._h); self._h = None

    def add(self, f, id_, opts, rel_id=0):
        f = _b(f); return lib().orc_router_add(self._h, f, len(f), C.byref(id_), C.byref(opts), rel_id)

    def set_shared_policy(self, policy):
        """SharedSubscription::choice stand-in: 0 = nobody (the reference's default), 1 = smallest client id,
        2 = largest rel_id of the group."""
        lib().orc_router_set_shared_policy(self._h, policy)

    def remove(self, f, id_):
        f = _b(f); return lib().orc_router_remove(self._h, f, len(f), C.byref(id_))

    def topics(self):
        return int(lib().orc_router_topics(self._h))

    def routes(self):
        return int(lib().orc_router_routes(self._h))

    def topics_tree(self):
        return int(lib().orc_router_topics_tree(self._h))

    def filter_id(self, f):
        f = _b(f); return int(lib().orc_router_filter_id(self._h, f, len(f)))

    def add_bulk(self, blob, offsets, client, qos):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        client = np.ascontiguousarray(client, dtype=np.uint32)
        qos = np.ascontiguousarray(qos, dtype=np.uint8)
        return lib().orc_router_add_bulk(self._h, _ptr(blob), offsets.ctypes.data, len(offsets) - 1,
                                         client.ctypes.data, qos.ctypes.data)

    def matches(self, this_id, topic):
        """Canonical text dump of the SubRelationsMap, or None on Err."""
        t = _b(topic)
        return _take_str(lib().orc_router_matches(self._h, C.byref(this_id), t, len(t)))

    def forwards(self, this_id, topic, pub_qos=0, pub_retain=False):
        """What forwards_to would send per node (text dump), or None on Err."""
        t = _b(topic)
        return _take_str(lib().orc_router_forwards(self._h, C.byref(this_id), t, len(t), pub_qos, 1 if pub_retain else 0))

    def match_flat(self, blob, offsets):
        n = len(offsets) - 1
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        status = np.zeros(n, dtype=np.int32)
        ps = [C.c_void_p() for _ in range(5)]
        st = OrcStats()
        tot = lib().orc_router_match_flat(self._h, _ptr(blob), offsets.ctypes.data, n, status.ctypes.data,
                                          *[C.byref(p) for p in ps], C.byref(st))
        return dict(status=status, hit_offsets=_take_arr(ps[0], n + 1, np.uint64),
                    filter_ids=_take_arr(ps[1], tot, np.uint32), sub_ids=_take_arr(ps[2], tot, np.uint32),
                    qos=_take_arr(ps[3], tot, np.uint8), flags=_take_arr(ps[4], tot, np.uint8), stats=st.as_dict())

    def match_timed(self, blob, offsets, threads=1):
        """Timed match_flat (the CHECKER's canonical form: sorts every relation list), static partition."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        st = OrcStats()
        sec = lib().orc_router_match_timed(self._h, _ptr(blob), offsets.ctypes.data, len(offsets) - 1, threads, C.byref(st))
        return float(sec), st.as_dict()

    def matches_timed(self, blob, offsets, threads=1, refcounted=True):
        """Timed DefaultRouter::_matches-shaped pass (router.rs:174-265: no canonicalising sort, per-hit
        ref-counted clones, dynamic chunks): the cpu_baseline figure.  refcounted=False: plain pointer copies
        instead of the ByteString-style atomic refcount bumps."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        st = OrcStats()
        sec = lib().orc_router_matches_timed(self._h, _ptr(blob), offsets.ctypes.data, len(offsets) - 1, threads, int(refcounted), C.byref(st))
        return float(sec), st.as_dict()

    def add_bulk_ex(self, blob, offsets, client, qos, flags):
        """add_bulk with per-subscription RGR_SUB_* flag bits (v5 = 1, No Local = 2, Retain As Published = 8)."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        client = np.ascontiguousarray(client, dtype=np.uint32); qos = np.ascontiguousarray(qos, dtype=np.uint8)
        flags = np.ascontiguousarray(flags, dtype=np.uint8)
        return int(lib().orc_router_add_bulk_ex(self._h, _ptr(blob), offsets.ctypes.data, len(offsets) - 1, client.ctypes.data, qos.ctypes.data, flags.ctypes.data))

    def deliver_digest(self, blob, offsets, pub_client, pub_qos_retain, threads=1):
        """-> (status, uint64[n,4]): per publish the digest of the delivery verdicts of its hits (DefaultRouter::deliver_digest, oracle.hpp)."""
        n = len(offsets) - 1
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        pc = np.ascontiguousarray(pub_client, dtype=np.uint32); pq = np.ascontiguousarray(pub_qos_retain, dtype=np.uint8)
        status = np.zeros(n, dtype=np.int32)
        out = np.zeros((n, 4), dtype=np.uint64)
        lib().orc_router_deliver_digest(self._h, _ptr(blob), offsets.ctypes.data, n, pc.ctypes.data, pq.ctypes.data, threads, status.ctypes.data, out.ctypes.data)
        return status, out

    def forwards_timed(self, blob, offsets, pub_client, pub_qos_retain, threads=1):
        """cpu_baseline of the delivery stage: _matches + collector + forwards_to's per-recipient transform.  -> (seconds, stats + rows)"""
        n = len(offsets) - 1
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        pc = np.ascontiguousarray(pub_client, dtype=np.uint32); pq = np.ascontiguousarray(pub_qos_retain, dtype=np.uint8)
        st = OrcStats(); rows = C.c_uint64(0)
        sec = lib().orc_router_forwards_timed(self._h, _ptr(blob), offsets.ctypes.data, n, pc.ctypes.data, pq.ctypes.data, threads, C.byref(st), C.byref(rows))
        d = st.as_dict(); d["rows"] = int(rows.value)
        return float(sec), d

    def match_digest(self, blob, offsets, threads=1, fast=False):
        """-> (status int32[n], digest uint64[n,4]) — see orc_router_match_digest (oracle.cpp).  fast: the same digests composed
        from per-filter pre-reduced digests (orc_router_match_digest_fast) — O(matched filters) per topic instead of O(hits)."""
        n = len(offsets) - 1
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        status = np.zeros(n, dtype=np.int32)
        out = np.zeros((n, 4), dtype=np.uint64)
        (lib().orc_router_match_digest_fast if fast else lib().orc_router_match_digest)(self._h, _ptr(blob), offsets.ctypes.data, n, threads, status.ctypes.data, out.ctypes.data)
        return status, out


def _ptr(buf):
    """bytes | numpy uint8 array | int address -> void* value"""
    if isinstance(buf, (bytes, bytearray)):
        return C.cast(C.c_char_p(bytes(buf)) if isinstance(buf, bytearray) else C.c_char_p(buf), C.c_void_p)
    if isinstance(buf, np.ndarray):
        return C.c_void_p(buf.ctypes.data)
    return C.c_void_p(int(buf))
