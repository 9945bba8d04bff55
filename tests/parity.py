"""Shared parity machinery: build the same subscription table in the oracle and in a
backend (HIP through the C ABI, or the host emulator), match the same topics, and compare
bit-exactly in the canonical form of SURVEY.md App. A.5:
  per topic, the sequence of (sub_id, qos|flags) must be identical — filters in
  TopicTree::matches' iteration order, subscribers ascending by sub_id inside a filter.
"""
import numpy as np

from oracle import oracle as orc
from rmqtt_amd import workload as wl


def make_backend(kind, **kw):
    if kind == "hip":
        from rmqtt_amd import capi
        kw.pop("lds_window", None)
        kw.pop("tile", None)
        return capi.Router(**kw)
    from tests.emu import emu
    return emu.EmuRouter(**kw)


def pack(strs):
    blob, offs = orc.pack_strings(strs)
    return np.frombuffer(blob, dtype=np.uint8), offs


class Pair:
    """Oracle DefaultRouter + backend router holding the same table."""

    def __init__(self, kind, **kw):
        self.oracle = orc.DefaultRouter()
        self.backend = make_backend(kind, **kw)
        self.fids = {}        # filter string -> backend filter id
        self.refs = {}        # filter string -> number of relations

    def add_bulk(self, blob, offs, client, qos):
        assert self.oracle.add_bulk(blob, offs, client, qos) == 0
        assert self.backend.subscribe_bulk(blob, offs, None, qos, None) == 0
        self.backend.commit()

    def add(self, filt, client, sub_id, qos=0, v5=False, no_local=False):
        o = self.oracle.add(filt, orc.mk_id(1, client), orc.mk_opts(qos=qos, v5=v5, no_local=no_local), rel_id=sub_id)
        if o != 0:
            try:
                self.backend.filter_add(filt)
            except Exception:
                return False
            raise AssertionError(f"oracle rejected {filt!r} but backend accepted it")
        fid = self.backend.filter_add(filt)
        self.fids[filt] = fid
        self.refs[filt] = self.refs.get(filt, 0) + 1
        self.backend.sub_add(fid, sub_id, qos, (1 if v5 else 0) | (2 if no_local else 0))
        return True

    def remove(self, filt, client, sub_id):
        assert self.oracle.remove(filt, orc.mk_id(1, client)) == 0
        fid = self.fids[filt]
        assert self.backend.sub_remove(fid, sub_id) == 0
        self.refs[filt] -= 1
        if self.refs[filt] == 0:            # router.rs:484-490: last relation gone => prune filter
            assert self.backend.filter_remove(fid) == 0
            del self.fids[filt], self.refs[filt]

    def commit(self):
        self.backend.commit()

    def check(self, blob, offs, what=""):
        exp = self.oracle.match_flat(blob, offs)
        got = self.backend.match_batch(blob, offs)
        compare_flat(got, exp, what)
        return exp, got


def compare_flat(got, exp, what=""):
    n = len(exp["status"])
    assert np.array_equal(got["status"] < 0, exp["status"] < 0), f"{what}: status differs"
    assert np.array_equal(got["hit_offsets"], exp["hit_offsets"]), f"{what}: hit offsets differ " + _first_diff(got, exp)
    t = got["tuples"]
    assert len(t) == len(exp["sub_ids"])
    assert np.array_equal(t["sub_id"], exp["sub_ids"]), f"{what}: sub_id sequence differs " + _first_diff(got, exp)
    qf = exp["qos"].astype(np.uint32) | (exp["flags"].astype(np.uint32) << 8)
    assert np.array_equal(t["qos_flags"], qf), f"{what}: qos/flags differ"
    # topic_idx column must restate the CSR offsets
    counts = np.diff(exp["hit_offsets"]).astype(np.int64)
    assert np.array_equal(t["topic_idx"], np.repeat(np.arange(n, dtype=np.uint32), counts)), f"{what}: topic_idx wrong"


def _first_diff(got, exp):
    a, b = got["hit_offsets"], exp["hit_offsets"]
    d = np.nonzero(a != b)[0]
    if len(d):
        i = int(d[0]) - 1
        return f"(first at topic {i}: got {int(a[i + 1] - a[i])} hits, expected {int(b[i + 1] - b[i])})"
    s, e = got["tuples"]["sub_id"], exp["sub_ids"]
    d = np.nonzero(s != e)[0]
    return f"(first differing hit {int(d[0])})" if len(d) else ""


def workload(cfg, n_sub, n_pub, seed_off=0):
    c = wl.CONFIGS[cfg]
    subs = wl.gen_subs(n_sub, wl.SUB_SEED + cfg + seed_off, c["p_plus"], c["p_hash"], c["p_sys"], 0, c["fixed_depth"])
    topics = wl.gen_topics(n_pub, wl.PUB_SEED + cfg + seed_off, 0.01 if cfg != 1 else 0.0, c["p_blank"], c["fixed_depth"])
    return subs, topics
