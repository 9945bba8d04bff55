"""ctypes binding of tests/hipsim — TEST INFRASTRUCTURE ONLY (kernel SOURCES of rmqtt_amd/csrc run on the host, one OS thread per GPU
thread; see hipsim.hpp).  Needs the ROCm clang (host compilation of HIP-style vector types and builtins)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
_LIB = None
SUB_DTYPE = np.dtype([("sub_id", np.uint32), ("qos_flags", np.uint32)])
FMT_SOA, FMT_PACKED, FMT_IDS24 = 1, 2, 4


def clang():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("amdclang++"), shutil.which("clang++")):
        if c and os.path.exists(c):
            return c
    return None


def build(extra=(), name="libhipsim_expand.so"):
    so = os.path.join(HERE, name)
    csrc = os.path.join(ROOT, "rmqtt_amd", "csrc")
    deps = [os.path.join(HERE, f) for f in ("sim_expand_compact.cpp", "hipsim.hpp")] + \
           [os.path.join(csrc, f) for f in ("expand_compact.inc", "match_core.hpp", "kernels.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        cc = clang()
        if cc is None:
            raise RuntimeError("hipsim needs clang++ (vector extensions of the kernel sources)")
        tmp = f"{so}.tmp{os.getpid()}"
        subprocess.check_call([cc, "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", *extra, "-I", os.path.join(ROOT, "include"), "-I", csrc,
                               os.path.join(HERE, "sim_expand_compact.cpp"), "-o", tmp])
        os.replace(tmp, so)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, i32, u64 = C.c_void_p, C.c_int32, C.c_uint64
        L.sim_expand_compact.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, u64, u64, vp, vp]
        L.sim_expand_compact.restype = i32
        L.sim_expand_ids24_fused.argtypes = [vp, vp, vp, vp, vp, u64, u64, u64, vp]
        L.sim_expand_ids24_fused.restype = C.c_int64
        _LIB = L
    return _LIB


def expand_compact(variant, fmt, tiles_per_block, subs, pair_src, pair_off, pair_lo, pair_hi, use_packed=True, guard=4096):
    """Run one window through an expansion kernel of expand_compact.inc.  Returns (ids bytes or u32 array, qos array or None), and checks
    that nothing outside the window's output was written."""
    subs = np.ascontiguousarray(subs, dtype=SUB_DTYPE)
    pair_src = np.ascontiguousarray(pair_src, dtype=np.uint32)
    pair_off = np.ascontiguousarray(pair_off, dtype=np.uint64)
    pair_topic = np.zeros(len(pair_src), dtype=np.uint32)
    # what pack_subs_kernel leaves (+ the padding the lane-held kernel's dummy loads rely on)
    packed = np.zeros(len(subs) + 16, dtype=np.uint32)
    packed[:len(subs)] = subs["sub_id"] | ((subs["qos_flags"] & 3) << 30)
    hits = int(pair_off[pair_hi] - pair_off[pair_lo])
    bph = 3 if fmt == FMT_IDS24 else 4
    out = np.full(hits * bph + 2 * guard, 0xA5, dtype=np.uint8)
    qos = np.full(hits + 2 * guard, 0xA5, dtype=np.uint8)
    rc = lib().sim_expand_compact(variant, fmt, tiles_per_block, subs.ctypes.data, packed.ctypes.data if use_packed else None,
                                  pair_src.ctypes.data, pair_topic.ctypes.data, pair_off.ctypes.data, pair_lo, pair_hi,
                                  out.ctypes.data + guard, qos.ctypes.data + guard)
    if rc == -2:
        raise AssertionError("hipsim: threads of a wave / block diverged around a convergent operation (__syncthreads, cross-lane read): "
                             "on the device a cross-lane read returns 0 for lanes that are switched off")
    if rc != 0:
        raise ValueError(f"sim_expand_compact: unknown combination variant={variant} fmt={fmt} T={tiles_per_block}")
    assert (out[:guard] == 0xA5).all() and (out[guard + hits * bph:] == 0xA5).all(), "write outside the window's ids"
    body = out[guard:guard + hits * bph]
    if fmt == FMT_SOA:
        assert (qos[:guard] == 0xA5).all() and (qos[guard + hits:] == 0xA5).all(), "write outside the window's qos bytes"
        return body.view(np.uint32).copy(), qos[guard:guard + hits].copy()
    assert (qos == 0xA5).all(), "qos bytes written by a format that has none"
    return (body.copy() if fmt == FMT_IDS24 else body.view(np.uint32).copy()), None


def expand_ids24_fused(subs, pair_src, pair_off, pair_lo, pair_mid, pair_hi):
    """expand_ids24_lp_tiles_kernel (RGR_TILES_FUSED): the window [pair_lo, pair_mid) expanded to 3-byte ids while the tail blocks of
    the same grid write the tile records of [pair_mid, pair_hi).  -> (ids bytes, differing records of the next window)."""
    subs = np.ascontiguousarray(subs, dtype=SUB_DTYPE)
    pair_src = np.ascontiguousarray(pair_src, dtype=np.uint32)
    pair_off = np.ascontiguousarray(pair_off, dtype=np.uint64)
    pair_topic = np.arange(len(pair_src), dtype=np.uint32)
    packed = np.zeros(len(subs) + 16, dtype=np.uint32)
    packed[:len(subs)] = subs["sub_id"] | ((subs["qos_flags"] & 3) << 30)
    hits = int(pair_off[pair_mid] - pair_off[pair_lo])
    out = np.full(hits * 3 + 64, 0xA5, dtype=np.uint8)
    d = lib().sim_expand_ids24_fused(subs.ctypes.data, packed.ctypes.data, pair_src.ctypes.data, pair_topic.ctypes.data, pair_off.ctypes.data,
                                     pair_lo, pair_mid, pair_hi, out.ctypes.data)
    if d == -2:
        raise AssertionError("hipsim: divergence in expand_ids24_lp_tiles_kernel")
    assert (out[hits * 3:] == 0xA5).all(), "write past the window's ids"
    return out[:hits * 3].copy(), int(d)


# ---------------------------------------------------------------------------------------------- v5 per-client dedup (dedup.inc)
CAND_DTYPE = np.dtype([("pos", np.uint32), ("client_idx", np.uint32)])
TUPLE_DTYPE = np.dtype([("topic_idx", np.uint32), ("sub_id", np.uint32), ("qos_flags", np.uint32)])
HIT8_DTYPE = np.dtype([("sub_id", np.uint32), ("word", np.uint32)])
_DLIB = None


def dedup_lib():
    global _DLIB
    if _DLIB is None:
        so = os.path.join(HERE, "libhipsim_dedup.so")
        csrc = os.path.join(ROOT, "rmqtt_amd", "csrc")
        deps = [os.path.join(HERE, f) for f in ("sim_dedup.cpp", "hipsim.hpp")] + [os.path.join(csrc, f) for f in ("dedup.inc", "match_core.hpp", "kernels.hpp")]
        if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            tmp = f"{so}.tmp{os.getpid()}"
            subprocess.check_call([clang(), "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", csrc,
                                   os.path.join(HERE, "sim_dedup.cpp"), "-o", tmp])
            os.replace(tmp, so)
        L = C.CDLL(so)
        vp, i32, u32, u64 = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64
        L.sim_dedup.argtypes = [i32, u32, u32, vp, vp, vp, u32, vp, u32, vp, u64, vp]
        L.sim_dedup.restype = i32
        _DLIB = L
    return _DLIB


def dedup(variant, hit_off, cand_pos, cand_client, tile=2048, grid_topic=1024, max_slots=4096):
    """One window through dedup_tile / dedup_classify / dedup_topic[_pipe].  hit_off: the window's per-topic offsets (n + 1, absolute);
    candidates: window-relative positions + client indices (any order).  Lays the candidates out as the expansion does (tile i owns
    cand[i * tile ...], its count carries bit 31 when a whole topic with candidates may lie inside it, tile_trange bounds its topics) and
    returns (the positions the kernels flagged kHitV5Dup, number of topic-pass items)."""
    hit_off = np.ascontiguousarray(hit_off, dtype=np.uint64)
    nt = len(hit_off) - 1
    hit_lo = int(hit_off[0])
    nh = int(hit_off[-1]) - hit_lo
    ntiles = (nh + tile - 1) // tile
    pos = np.asarray(cand_pos, dtype=np.int64)
    cl = np.asarray(cand_client, dtype=np.uint32)
    cand = np.zeros(ntiles * tile, dtype=CAND_DTYPE)
    ncand = np.zeros(ntiles, dtype=np.uint32)
    trange = np.zeros(2 * ntiles, dtype=np.uint32)
    rel = hit_off.astype(np.int64) - hit_lo
    t_of = np.searchsorted(rel, pos, side="right") - 1
    order = np.argsort(pos // tile, kind="stable")
    for i in order:
        tl = int(pos[i]) // tile
        cand[tl * tile + int(ncand[tl])] = (pos[i], cl[i])
        ncand[tl] += 1
    for tl in range(ntiles):
        lo, hi = tl * tile, min(nh, (tl + 1) * tile)
        t_first = int(np.searchsorted(rel, lo, side="right") - 1)
        t_last = int(np.searchsorted(rel, hi - 1, side="right") - 1)
        # the expansion's flag (expand_tuple.inc tile_whole_topic_flag): two topics meeting in the tile count only when one of them is whole
        first_whole, last_whole = rel[t_first] >= lo, rel[t_last + 1] <= lo + tile
        whole = (first_whole and last_whole) if t_first == t_last else (first_whole or last_whole) if t_last - t_first == 1 else True
        if ncand[tl] >= 2 and whole:
            ncand[tl] |= 1 << 31
            trange[2 * tl], trange[2 * tl + 1] = t_first, t_last
    tuples = np.zeros(nh, dtype=TUPLE_DTYPE)
    n_items = C.c_uint32(0)
    rc = dedup_lib().sim_dedup(variant, grid_topic, max_slots, cand.ctypes.data, ncand.ctypes.data, trange.ctypes.data, ntiles, tuples.ctypes.data, nt,
                               hit_off.ctypes.data, hit_lo, C.byref(n_items))
    if rc == -2:
        raise AssertionError("hipsim: threads diverged around a convergent operation in the dedup kernels")
    assert rc == 0
    assert not (tuples["qos_flags"] & ~np.uint32(16)).any() and not tuples["topic_idx"].any() and not tuples["sub_id"].any()
    return np.flatnonzero(tuples["qos_flags"] & 16), int(n_items.value), t_of


# ---------------------------------------------------------------------------------------------- tuple expansions (expand_tuple.inc)
ATTR_DTYPE = np.dtype([("owner_id", np.uint32), ("client_idx", np.uint32)])
PUB_DTYPE = np.dtype([("from_id", np.uint32), ("qos_retain", np.uint32)])
_TLIB = None


def tuple_lib():
    global _TLIB
    if _TLIB is None:
        so = os.path.join(HERE, "libhipsim_expand_tuple.so")
        csrc = os.path.join(ROOT, "rmqtt_amd", "csrc")
        deps = [os.path.join(HERE, f) for f in ("sim_expand_tuple.cpp", "hipsim.hpp")] + [os.path.join(csrc, f) for f in ("expand_tuple.inc", "match_core.hpp", "kernels.hpp")]
        if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            tmp = f"{so}.tmp{os.getpid()}"
            subprocess.check_call([clang(), "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", csrc,
                                   os.path.join(HERE, "sim_expand_tuple.cpp"), "-o", tmp])
            os.replace(tmp, so)
        L = C.CDLL(so)
        vp, i32, u32, u64 = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64
        L.sim_expand_tuple.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, u64, u64, u32, vp, vp, vp, vp]
        L.sim_expand_tuple.restype = i32
        _TLIB = L
    return _TLIB


def expand_tuple(variant, subs, attrs, pub, pair_src, pair_topic, pair_off, pair_qr, pair_lo, pair_hi, topic_lo, want_cand=True, tile=2048):
    """One window through a tuple expansion kernel.  -> (tuples, per-tile candidate lists [(pos, client) sorted], tile_ncand raw words,
    tile_trange) — the last three None for variant 0 / want_cand False."""
    subs = np.ascontiguousarray(subs, dtype=SUB_DTYPE)
    attrs = None if attrs is None else np.ascontiguousarray(attrs, dtype=ATTR_DTYPE)
    pub = np.ascontiguousarray(pub, dtype=PUB_DTYPE)
    pair_src = np.ascontiguousarray(pair_src, dtype=np.uint32)
    pair_topic = np.ascontiguousarray(pair_topic, dtype=np.uint32)
    pair_off = np.ascontiguousarray(pair_off, dtype=np.uint64)
    pair_qr = np.ascontiguousarray(pair_qr, dtype=np.uint8)
    nh = int(pair_off[pair_hi] - pair_off[pair_lo])
    ntiles = (nh + tile - 1) // tile
    hits8 = variant in (5, 6)                       # 8-byte hits {sub_id, word}: the buffer is sized for them and guarded right behind
    out = np.zeros(nh + 1, dtype=HIT8_DTYPE if hits8 else TUPLE_DTYPE)
    out[nh] = (0xA5A5A5A5, 0xA5A5A5A5) if hits8 else (0xA5A5A5A5, 0xA5A5A5A5, 0xA5A5A5A5)
    deliver = variant != 0
    cand = np.full(ntiles * tile, 0xFFFFFFFF, dtype=np.uint64).view(CAND_DTYPE) if deliver and want_cand else None
    ncand = np.full(ntiles, 0xDEADBEEF, dtype=np.uint32) if deliver and want_cand else None
    trange = np.full(2 * ntiles, 0xDEADBEEF, dtype=np.uint32) if deliver and want_cand else None
    p = lambda a: None if a is None else a.ctypes.data
    rc = tuple_lib().sim_expand_tuple(variant, subs.ctypes.data, p(attrs), pub.ctypes.data, pair_src.ctypes.data, pair_topic.ctypes.data, pair_off.ctypes.data,
                                      pair_qr.ctypes.data, pair_lo, pair_hi, topic_lo, out.ctypes.data, p(cand), p(ncand), p(trange))
    if rc == -2:
        raise AssertionError("hipsim: threads diverged around a convergent operation in the tuple expansion")
    assert rc == 0
    assert all(int(x) == 0xA5A5A5A5 for x in out[nh]), "write past the window's tuples"
    lists = None
    if cand is not None:
        lists = []
        for t in range(ntiles):
            n = int(ncand[t] & 0x7FFFFFFF)
            sl = cand[t * tile:t * tile + n]
            lists.append(sorted(zip(sl["pos"].tolist(), sl["client_idx"].tolist())))
    return out[:nh].copy(), lists, ncand, trange


# ---------------------------------------------------------------------------------------------- count / compact, batched (prep_batched.inc)
DESC_DTYPE = np.dtype([("begin", np.uint32), ("count", np.uint32)])
_PLIB = None


def prep_lib():
    global _PLIB
    if _PLIB is None:
        so = os.path.join(HERE, "libhipsim_prep.so")
        csrc = os.path.join(ROOT, "rmqtt_amd", "csrc")
        deps = [os.path.join(HERE, f) for f in ("sim_prep.cpp", "hipsim.hpp")] + [os.path.join(csrc, f) for f in ("prep_batched.inc", "match_core.hpp", "kernels.hpp")]
        if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            tmp = f"{so}.tmp{os.getpid()}"
            subprocess.check_call([clang(), "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", csrc,
                                   os.path.join(HERE, "sim_prep.cpp"), "-o", tmp])
            os.replace(tmp, so)
        L = C.CDLL(so)
        vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
        L.sim_prep.argtypes = [u32, u32, vp, vp, vp, vp, u64, vp, u32, vp, vp, vp]
        L.sim_prep.restype = C.c_int64
        _PLIB = L
    return _PLIB


def prep(lists, filt, slot_cap, topic_base=0, pub=None):
    """lists: per topic the matched filter ids in order.  Lays them out as the walk does (the first slot_cap-sized lists j-major in the
    slot array, longer ones in the overflow arena) and runs the batched count / compact kernels beside the shared per-topic functions.
    -> (differing words, pairs, hits)."""
    n = len(lists)
    cnt = np.array([len(x) for x in lists], dtype=np.uint32)
    slots = np.full(max(1, slot_cap) * n, 0xDEADBEEF, dtype=np.uint32)
    ovf_base = np.zeros(n, dtype=np.uint64)
    arena = []
    for t, ids in enumerate(lists):
        if len(ids) <= slot_cap:
            for j, f in enumerate(ids):
                slots[j * n + t] = f
        else:
            ovf_base[t] = len(arena)
            arena += list(ids)
    arena = np.asarray(arena + [0xDEADBEEF], dtype=np.uint32)
    filt = np.ascontiguousarray(filt, dtype=DESC_DTYPE)
    pub_arr = None if pub is None else np.ascontiguousarray(pub, dtype=PUB_DTYPE)
    npairs, nhits = C.c_uint64(0), C.c_uint64(0)
    d = prep_lib().sim_prep(n, slot_cap, slots.ctypes.data, cnt.ctypes.data, ovf_base.ctypes.data, arena.ctypes.data, len(arena), filt.ctypes.data, topic_base,
                            None if pub_arr is None else pub_arr.ctypes.data, C.byref(npairs), C.byref(nhits))
    if d == -2:
        raise AssertionError("hipsim: threads diverged around a barrier in the batched count / compact kernels")
    return int(d), int(npairs.value), int(nhits.value)
