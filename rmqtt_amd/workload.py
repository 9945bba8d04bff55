"""Seeded synthetic workloads of SURVEY.md §8(d) / BASELINE.json ``configs``.

Thin ctypes wrapper over csrc/workload.cpp so the oracle, the GPU path and the
bench all see byte-identical inputs.  Strings are handed around as
(blob: uint8[...], offsets: uint64[n+1]).
"""
import ctypes as C

import numpy as np

from . import build as _build

_LIB = None


class _Params(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n", C.c_uint64), ("p_plus", C.c_double), ("p_hash", C.c_double),
                ("p_sys", C.c_double), ("p_blank", C.c_double), ("n_clients", C.c_uint64),
                ("fixed_depth", C.c_int32), ("force_wildcard", C.c_int32), ("distinct", C.c_int32), ("reserved", C.c_int32)]


def _lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(_build.build_workload())
        vp = C.c_void_p
        L.wl_gen_subs.argtypes = [C.POINTER(_Params)] + [C.POINTER(vp)] * 4
        L.wl_gen_topics.argtypes = [C.POINTER(_Params)] + [C.POINTER(vp)] * 2
        L.wl_take.argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(vp), C.POINTER(vp)]
        L.wl_free.argtypes = [vp]
        _LIB = L
    return _LIB


def _take(p, n, dtype):
    ct = np.ctypeslib.as_ctypes_type(dtype)
    a = np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(max(int(n), 1),))[: int(n)].copy()
    _lib().wl_free(p)
    return a


# BASELINE.json configs 1..5 (index 0 unused).  p_plus = 0.028/level gives ≈20 % of
# filters with at least one '+' at the mean depth of 8.
CONFIGS = {
    1: dict(n_sub=10_000, n_pub=100_000, p_plus=0.0, p_hash=0.0, p_sys=0.0, p_blank=0.0, fixed_depth=4),
    2: dict(n_sub=1_000_000, n_pub=1_000_000, p_plus=0.028, p_hash=0.0, p_sys=0.005, p_blank=0.01, fixed_depth=0),
    3: dict(n_sub=10_000_000, n_pub=10_000_000, p_plus=0.028, p_hash=0.10, p_sys=0.005, p_blank=0.01, fixed_depth=0),
    4: dict(n_sub=10_000_000, n_pub=10_000_000, p_plus=0.028, p_hash=0.10, p_sys=0.005, p_blank=0.01, fixed_depth=0),
    5: dict(n_sub=5_000_000, n_pub=1_000_000, p_plus=0.028, p_hash=0.10, p_sys=0.005, p_blank=0.01, fixed_depth=0),
}
SUB_SEED = 0x5EED0000
PUB_SEED = 0x9B1C0000


def gen_subs(n, seed, p_plus=0.028, p_hash=0.0, p_sys=0.0, n_clients=0, fixed_depth=0, force_wildcard=False):
    """-> (blob uint8[], offsets uint64[n+1], client uint32[n], qos uint8[n])"""
    p = _Params(seed, n, p_plus, p_hash, p_sys, 0.0, n_clients, fixed_depth, int(force_wildcard), 0, 0)
    ps = [C.c_void_p() for _ in range(4)]
    rc = _lib().wl_gen_subs(C.byref(p), *[C.byref(x) for x in ps])
    assert rc == 0
    offs = _take(ps[1], n + 1, np.uint64)
    blob = _take(ps[0], offs[-1], np.uint8)
    return blob, offs, _take(ps[2], n, np.uint32), _take(ps[3], n, np.uint8)


def gen_topics(n, seed, p_sys=0.01, p_blank=0.01, fixed_depth=0, distinct=False):
    """-> (blob uint8[], offsets uint64[n+1])"""
    p = _Params(seed, n, 0.0, 0.0, p_sys, p_blank, 0, fixed_depth, 0, int(distinct), 0)
    ps = [C.c_void_p() for _ in range(2)]
    rc = _lib().wl_gen_topics(C.byref(p), *[C.byref(x) for x in ps])
    assert rc == 0
    offs = _take(ps[1], n + 1, np.uint64)
    blob = _take(ps[0], offs[-1], np.uint8)
    return blob, offs


def take(blob, offsets, idx):
    """Gather strings idx of (blob, offsets) -> (blob uint8[], offsets uint64[m+1])."""
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    idx = np.ascontiguousarray(idx, dtype=np.uint64)
    pb, po = C.c_void_p(), C.c_void_p()
    _lib().wl_take(blob.ctypes.data, offsets.ctypes.data, idx.ctypes.data, len(idx), C.byref(pb), C.byref(po))
    o = _take(po, len(idx) + 1, np.uint64)
    return _take(pb, o[-1], np.uint8), o


def config_subs(cfg, scale=1.0):
    c = CONFIGS[cfg]
    n = max(1, int(c["n_sub"] * scale))
    return gen_subs(n, SUB_SEED + cfg, c["p_plus"], c["p_hash"], c["p_sys"], 0, c["fixed_depth"])


def config_topics(cfg, scale=1.0):
    c = CONFIGS[cfg]
    n = max(1, int(c["n_pub"] * scale))
    sys_p = 0.01 if cfg != 1 else 0.0
    return gen_topics(n, PUB_SEED + cfg, sys_p, c["p_blank"], c["fixed_depth"])


def strings(blob, offsets, lo=0, hi=None):
    hi = len(offsets) - 1 if hi is None else hi
    b = blob.tobytes() if isinstance(blob, np.ndarray) else bytes(blob)
    return [b[int(offsets[i]):int(offsets[i + 1])].decode() for i in range(lo, hi)]
