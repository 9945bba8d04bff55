"""ctypes binding of tests/emu/emu.cpp — TEST INFRASTRUCTURE ONLY (host emulation of the
HIP pipeline over the product's own table compiler + match_core.hpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
_LIB = None

TUPLE_DTYPE = np.dtype([("topic_idx", np.uint32), ("sub_id", np.uint32), ("qos_flags", np.uint32)])


def build():
    so = os.path.join(HERE, "libemu.so")
    csrc = os.path.join(ROOT, "rmqtt_amd", "csrc")
    deps = [os.path.join(HERE, "emu.cpp")] + [os.path.join(csrc, f) for f in
                                               ("table.cpp", "table.hpp", "retain.cpp", "retain.hpp", "match_core.hpp", "kernels.hpp", "topic.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), "-I", csrc,
                               os.path.join(HERE, "emu.cpp"), os.path.join(csrc, "table.cpp"), os.path.join(csrc, "retain.cpp"), "-o", so])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, u32, u64, u8 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint8
        L.emu_new.argtypes = [u32, u32, u64, u32, u32]; L.emu_new.restype = vp
        L.emu_free.argtypes = [vp]; L.emu_free_buf.argtypes = [vp]
        L.emu_filter_add.argtypes = [vp, C.c_char_p, u32, C.POINTER(u32)]
        L.emu_filter_find.argtypes = [vp, C.c_char_p, u32, C.POINTER(u32)]
        L.emu_filter_remove.argtypes = [vp, u32]
        L.emu_sub_add.argtypes = [vp, u32, u32, u8, u8]
        L.emu_sub_add_ex.argtypes = [vp, u32, u32, u8, u8, C.c_uint16, u32, u32]
        L.emu_match_deliver.argtypes = [vp, vp, vp, u32, vp, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64), C.POINTER(vp), C.POINTER(vp)]
        L.emu_sub_remove.argtypes = [vp, u32, u32]
        L.emu_snapshot_save.argtypes = [vp, C.c_char_p]
        L.emu_snapshot_load.argtypes = [vp, C.c_char_p]
        for f in ("emu_n_nodes", "emu_n_filters", "emu_n_subs", "emu_visited", "emu_overflow_topics", "emu_windows"):
            getattr(L, f).argtypes = [vp]; getattr(L, f).restype = u64
        L.emu_subscribe_bulk.argtypes = [vp, vp, vp, u64, vp, vp, vp, C.POINTER(u64)]
        L.emu_match.argtypes = [vp, vp, vp, u32, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64), C.POINTER(vp), C.POINTER(vp)]
        L.emu_retain_add.argtypes = [vp, C.c_char_p, u32, u32]
        L.emu_retain_remove.argtypes = [vp, C.c_char_p, u32]
        L.emu_retain_topics.argtypes = [vp]; L.emu_retain_topics.restype = u64
        L.emu_retain_version.argtypes = [vp]; L.emu_retain_version.restype = u64
        L.emu_tier_add.argtypes = [vp, C.c_char_p, u32, u32]
        L.emu_tier_remove.argtypes = [vp, C.c_char_p, u32]
        L.emu_tier_counter.argtypes = [vp, C.c_int]; L.emu_tier_counter.restype = u64
        L.emu_tier_commit.argtypes = [vp, u64]
        L.emu_tier_match.argtypes = [vp, vp, vp, u32, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]
        L.emu_retain_nodes.argtypes = [vp]; L.emu_retain_nodes.restype = u64
        L.emu_retain_add_bulk.argtypes = [vp, vp, vp, u64, vp, C.POINTER(u64)]
        L.emu_retain_match.argtypes = [vp, vp, vp, u32, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]
        L.emu_publish_scan.argtypes = [vp, vp, u32, C.c_int, vp]; L.emu_publish_scan.restype = None
        _LIB = L
    return _LIB


def _b(s):
    return s if isinstance(s, (bytes, bytearray)) else s.encode()


def _take(p, n, dtype):
    if n:
        a = np.frombuffer(C.string_at(p, int(n) * np.dtype(dtype).itemsize), dtype=dtype).copy()
    else:
        a = np.zeros(0, dtype=dtype)
    lib().emu_free_buf(p)
    return a


class EmuRouter:
    """Same surface as rmqtt_amd.capi.Router for the calls the parity tests use."""

    def __init__(self, slot_cap=0, chunk_topics=0, window_hits=0, lds_window=2560, tile=0):
        self._h = lib().emu_new(slot_cap, chunk_topics, window_hits, lds_window, tile)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().emu_free(self._h); self._h = None

    def filter_add(self, f):
        f = _b(f); fid = C.c_uint32()
        rc = lib().emu_filter_add(self._h, f, len(f), C.byref(fid))
        if rc != 0:
            raise ValueError(rc)
        return fid.value

    def filter_find(self, f):
        f = _b(f); fid = C.c_uint32()
        return fid.value if lib().emu_filter_find(self._h, f, len(f), C.byref(fid)) == 0 else None

    def filter_remove(self, fid):
        return lib().emu_filter_remove(self._h, fid)

    def sub_add(self, fid, sub_id, qos=0, flags=0):
        assert lib().emu_sub_add(self._h, fid, sub_id, qos, flags) == 0

    def sub_add_ex(self, fid, sub_id, qos=0, flags=0, node_idx=0, owner_id=0xFFFFFFFF, client_idx=0xFFFFFFFF):
        assert lib().emu_sub_add_ex(self._h, fid, sub_id, qos, flags, node_idx, owner_id, client_idx) == 0

    def sub_remove(self, fid, sub_id):
        return lib().emu_sub_remove(self._h, fid, sub_id)

    def subscribe_bulk(self, blob, offsets, sub_ids=None, qos=None, flags=None):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        blob = np.ascontiguousarray(np.frombuffer(bytes(blob), dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob)
        keep = []
        def p(a, dt):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=dt); keep.append(a)
            return C.c_void_p(a.ctypes.data)
        rej = C.c_uint64(0)
        lib().emu_subscribe_bulk(self._h, blob.ctypes.data, offsets.ctypes.data, len(offsets) - 1, p(sub_ids, np.uint32),
                                 p(qos, np.uint8), p(flags, np.uint8), C.byref(rej))
        return int(rej.value)

    def commit(self):
        pass

    def snapshot_save(self, path):
        assert lib().emu_snapshot_save(self._h, os.fsencode(path)) == 0

    def snapshot_load(self, path):
        if lib().emu_snapshot_load(self._h, os.fsencode(path)) != 0:
            raise ValueError("bad snapshot")

    def _match(self, blob, offsets, publish_attrs=None):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        blob = np.ascontiguousarray(np.frombuffer(bytes(blob), dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob)
        n = len(offsets) - 1
        status = np.zeros(n, dtype=np.int32)
        ho, tp, po, pf = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        nh = C.c_uint64(0)
        pa = None
        if publish_attrs is not None:
            pa = np.ascontiguousarray(publish_attrs, dtype=np.dtype([("from_id", np.uint32), ("qos_retain", np.uint32)]))
            assert len(pa) == n
        rc = lib().emu_match_deliver(self._h, blob.ctypes.data if len(blob) else None, offsets.ctypes.data, n,
                                     None if pa is None else pa.ctypes.data, status.ctypes.data,
                                     C.byref(ho), C.byref(tp), C.byref(nh), C.byref(po), C.byref(pf))
        assert rc == 0, rc
        hit_offsets = _take(ho, n + 1, np.uint64)
        tuples = _take(tp, nh.value, TUPLE_DTYPE)
        pair_offsets = _take(po, n + 1, np.uint64)
        fids = _take(pf, int(pair_offsets[-1]), np.uint32)
        return status, hit_offsets, tuples, pair_offsets, fids

    def match_batch(self, blob, offsets):
        s, ho, tp, _, _ = self._match(blob, offsets)
        return dict(status=s, hit_offsets=ho, tuples=tp)

    def match_batch_deliver(self, blob, offsets, publish_attrs, grouped=False):
        s, ho, tp, _, _ = self._match(blob, offsets, publish_attrs)
        if not grouped:
            return dict(status=s, hit_offsets=ho, tuples=tp)
        # rgr_match_batch_deliver_grouped, stated independently of the device's radix passes: per topic a stable sort by node index
        tp = tp.copy()
        go, gn, gb = [0], [], []
        for t in range(len(ho) - 1):
            a, e = int(ho[t]), int(ho[t + 1])
            seg = tp[a:e]
            node = seg["qos_flags"] >> 16
            seg = seg[np.argsort(node, kind="stable")]
            tp[a:e] = seg
            node = seg["qos_flags"] >> 16
            starts = np.nonzero(np.r_[True, node[1:] != node[:-1]])[0] if e > a else np.zeros(0, dtype=np.int64)
            gn.extend(int(x) for x in node[starts])
            gb.extend(a + int(x) for x in starts)
            go.append(len(gn))
        gb.append(len(tp))
        return dict(status=s, hit_offsets=ho, tuples=tp, group_offsets=np.array(go, dtype=np.uint64), group_node=np.array(gn, dtype=np.uint32),
                    group_begin=np.array(gb, dtype=np.uint64))

    def match_filters(self, blob, offsets):
        s, _, _, po, pf = self._match(blob, offsets)
        return dict(status=s, pair_offsets=po, filter_ids=pf)

    # ---- retain twin (same surface as capi.Router)
    def retain_add(self, topic, topic_id):
        t = _b(topic); return lib().emu_retain_add(self._h, t, len(t), topic_id)

    # ---- two-tier retained set (what rgr_config.retain_delta_max > 0 turns on in the product)
    def tier_add(self, topic, topic_id):
        t = _b(topic)
        return lib().emu_tier_add(self._h, t, len(t), topic_id)

    def tier_remove(self, topic):
        t = _b(topic)
        return lib().emu_tier_remove(self._h, t, len(t))

    def tier_commit(self, delta_max):
        rc = lib().emu_tier_commit(self._h, delta_max)
        assert rc == 0, rc

    def tier_counters(self):
        names = ["n_topics", "n_delta", "n_dead", "merges", "delta_compiles"]
        return {k: int(lib().emu_tier_counter(self._h, i)) for i, k in enumerate(names)}

    def tier_match_batch(self, blob, offsets):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        blob = np.ascontiguousarray(np.frombuffer(bytes(blob), dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob)
        n = len(offsets) - 1
        status = np.zeros(n, dtype=np.int32)
        ho, ids = C.c_void_p(), C.c_void_p()
        nh = C.c_uint64(0)
        rc = lib().emu_tier_match(self._h, blob.ctypes.data if len(blob) else None, offsets.ctypes.data, n, status.ctypes.data,
                                  C.byref(ho), C.byref(ids), C.byref(nh))
        assert rc == 0, rc
        return dict(status=status, hit_offsets=_take(ho, n + 1, np.uint64), topic_ids=_take(ids, nh.value, np.uint32))

    def retain_version(self):
        return int(lib().emu_retain_version(self._h))

    def retain_remove(self, topic):
        t = _b(topic); return lib().emu_retain_remove(self._h, t, len(t))

    def retain_add_bulk(self, blob, offsets, topic_ids=None):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        blob = np.ascontiguousarray(np.frombuffer(bytes(blob), dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob)
        ids = None if topic_ids is None else np.ascontiguousarray(topic_ids, dtype=np.uint32)
        rej = C.c_uint64(0)
        lib().emu_retain_add_bulk(self._h, blob.ctypes.data, offsets.ctypes.data, len(offsets) - 1,
                                  None if ids is None else C.c_void_p(ids.ctypes.data), C.byref(rej))
        return int(rej.value)

    def retain_commit(self):
        pass

    def retain_match_batch(self, blob, offsets):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        blob = np.ascontiguousarray(np.frombuffer(bytes(blob), dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob)
        n = len(offsets) - 1
        status = np.zeros(n, dtype=np.int32)
        ho, tp = C.c_void_p(), C.c_void_p()
        nh = C.c_uint64(0)
        rc = lib().emu_retain_match(self._h, blob.ctypes.data if len(blob) else None, offsets.ctypes.data, n, status.ctypes.data,
                                    C.byref(ho), C.byref(tp), C.byref(nh))
        assert rc == 0, rc
        hit_offsets = _take(ho, n + 1, np.uint64)
        tuples = _take(tp, nh.value, TUPLE_DTYPE)
        return dict(status=status, hit_offsets=hit_offsets, topic_ids=tuples["sub_id"].copy(), filter_idx=tuples["topic_idx"].copy())

    def counters(self):
        return {k: int(getattr(lib(), "emu_" + k)(self._h)) for k in ("n_nodes", "n_filters", "n_subs", "visited", "overflow_topics", "windows")}


PUBLISH_INFO_DTYPE = np.dtype([("topic_off", np.uint64), ("topic_len", np.uint32), ("payload_off", np.uint32), ("packet_id", np.uint16),
                               ("qos", np.uint8), ("retain", np.uint8), ("dup", np.uint8), ("error", np.uint8), ("_pad", np.uint8, 2)])


def publish_scan(blob, offsets, version):
    """The device's per-packet PUBLISH scan (match_core.hpp publish_scan) executed on the host."""
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    out = np.zeros(n, dtype=PUBLISH_INFO_DTYPE)
    lib().emu_publish_scan(blob.ctypes.data, offsets.ctypes.data, n, version, out.ctypes.data)
    return out
