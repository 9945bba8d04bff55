"""Delivery stage vs the oracle.

The oracle side is `DefaultRouter::_matches` (router.rs:174-265: No Local, per-node collectors),
the v3/v5 collector (types.rs:510-540) and forwards_to's per-recipient transform
(shared.rs:886-908: Retain-As-Published, qos downgrade, subscription identifiers), dumped by
oracle `forwards`.  The backend side is rgr_match_batch_deliver: tuples whose third word is the
delivery word; the test folds them back into the same dump using only what the Rust glue would
hold (sub_id -> relation) and compares text for text.
"""
import random

import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as orc
from rmqtt_amd import capi
from rmqtt_amd import workload as wl
from tests.parity import make_backend, pack

BACKENDS = ["emu", pytest.param("hip", marks=pytest.mark.gpu)]

FILTERS = ["a/b", "a/+", "a/#", "#", "+/b", "a/b/c", "a/+/c", "+/+", "+/+/+", "$SYS/#", "$SYS/+", "+/#", "a/b/#", "x", "a//b", "/+", "a/b/+"]
TOPICS = ["a/b", "a/b/c", "a/x", "x", "$SYS/y", "a", "a//b", "+/b", "a/+", "a/#/b", "", "/b", "a/b/c/d", "q/r/s"]


class World:
    """Same subscription table in the oracle and in a backend, plus the id maps the glue keeps."""

    def __init__(self, kind, seed, **kw):
        self.rng = random.Random(seed)
        self.oracle = orc.DefaultRouter()
        self.backend = make_backend(kind, **kw)
        self.nodes = [1, 2, 7]
        self.clients = [f"c{i}" for i in range(24)]
        self.client_node = {c: self.rng.choice(self.nodes) for c in self.clients}
        self.owner_ids = {}            # (node, client, create_time) -> dense owner id
        self.client_idx = {}           # (node, client) -> dense client idx
        self.rels = {}                 # sub_id -> dict(filter, client, node, ident)
        self.sub_of = {}               # (filter, client) -> sub_id
        self.fids, self.refs = {}, {}
        self.next_sub = 0

    def owner(self, node, client, ct):
        return self.owner_ids.setdefault((node, client, ct), len(self.owner_ids))

    def add(self, filt, client, ct, qos, v5, no_local, rap, ident, shared=False):
        node = self.client_node[client]
        o = self.oracle.add(filt, orc.mk_id(node, client, ct), orc.mk_opts(qos=qos, v5=v5, no_local=no_local, sub_ident=ident, rap=rap),
                            rel_id=0)
        assert o == 0
        key = (filt, client)
        if key not in self.sub_of:
            self.sub_of[key] = self.next_sub
            self.next_sub += 1
            self.refs[filt] = self.refs.get(filt, 0) + 1
        sid = self.sub_of[key]
        fid = self.backend.filter_add(filt)
        self.fids[filt] = fid
        flags = (capi.RGR_SUB_V5 if v5 else 0) | (capi.RGR_SUB_NO_LOCAL if v5 and no_local else 0) | (capi.RGR_SUB_RAP if v5 and rap else 0)
        cidx = self.client_idx.setdefault((node, client), len(self.client_idx))
        self.backend.sub_add_ex(fid, sid, qos, flags, self.nodes.index(node), self.owner(node, client, ct), cidx)
        self.rels[sid] = dict(filter=filt, client=client, node=node, ident=ident if v5 else 0, v5=v5)

    def remove(self, filt, client, ct):
        node = self.client_node[client]
        rc = self.oracle.remove(filt, orc.mk_id(node, client, ct))
        assert rc in (0, 1)
        if rc != 0:
            return False                  # stored Id differs (router.rs:460-467): nothing removed
        sid = self.sub_of.pop((filt, client))
        assert self.backend.sub_remove(self.fids[filt], sid) == 0
        del self.rels[sid]
        self.refs[filt] -= 1
        if self.refs[filt] == 0:
            assert self.backend.filter_remove(self.fids[filt]) == 0
            del self.fids[filt], self.refs[filt]
        return True

    def random_sub(self):
        r = self.rng
        v5 = r.random() < 0.55
        return dict(filt=r.choice(FILTERS), client=r.choice(self.clients), ct=r.choice([0, 0, 0, 5]), qos=r.randrange(3), v5=v5,
                    no_local=v5 and r.random() < 0.4, rap=v5 and r.random() < 0.5, ident=r.randrange(1, 90) if v5 and r.random() < 0.6 else 0)

    # ---- one batch of publishes through both sides
    def check(self, n_pub=60):
        r = self.rng
        known = list(self.owner_ids)
        pubs = []
        for _ in range(n_pub):
            topic = r.choice(TOPICS)
            if known and r.random() < 0.7:
                node, client, ct = r.choice(known)
            else:
                node, client, ct = r.choice(self.nodes), "stranger", 0
            pubs.append((topic, node, client, ct, r.randrange(3), r.random() < 0.5))
        blob, offs = pack([p[0] for p in pubs])
        attrs = np.zeros(len(pubs), dtype=capi.PUBLISH_ATTR_DTYPE)
        for i, (_, node, client, ct, q, ret) in enumerate(pubs):
            attrs[i] = (self.owner_ids.get((node, client, ct), capi.ID_NONE), q | (4 if ret else 0))
        self.backend.commit()
        got = self.backend.match_batch_deliver(blob, offs, attrs)
        plain = self.backend.match_batch(blob, offs)
        assert np.array_equal(plain["hit_offsets"], got["hit_offsets"]) and np.array_equal(plain["tuples"]["sub_id"], got["tuples"]["sub_id"])
        for i, (topic, node, client, ct, q, ret) in enumerate(pubs):
            exp = self.oracle.forwards(orc.mk_id(node, client, ct), topic, q, ret)
            lo, hi = int(got["hit_offsets"][i]), int(got["hit_offsets"][i + 1])
            if exp is None:
                assert got["status"][i] < 0 and lo == hi
                continue
            assert got["status"][i] == 0
            assert self.fold(got["tuples"][lo:hi], i) == exp, (topic, client, ct)

    def fold(self, tuples, topic_idx):
        """Delivery words -> the oracle's dump format, as the glue would build its SubRelationsMap."""
        per_node = {}
        v5_rows = {}
        for tp in tuples:
            assert tp["topic_idx"] == topic_idx
            w = int(tp["qos_flags"])
            rel = self.rels[int(tp["sub_id"])]
            assert self.nodes[w >> 16] == rel["node"]
            assert bool((w >> 8) & capi.RGR_SUB_V5) == rel["v5"]
            if w & capi.RGR_HIT_NO_LOCAL:
                continue
            rows = per_node.setdefault(rel["node"], [])
            if not rel["v5"]:
                assert not (w & (capi.RGR_HIT_V5_DUP | capi.RGR_HIT_RETAIN))
                rows.append([rel["client"], rel["filter"], w & 3, 0, None])
            elif w & capi.RGR_HIT_V5_DUP:
                row = v5_rows[(rel["node"], rel["client"])]
                if rel["ident"]:
                    row[4] = (row[4] or []) + [rel["ident"]]
            else:
                assert (rel["node"], rel["client"]) not in v5_rows
                row = [rel["client"], rel["filter"], w & 3, 1 if w & capi.RGR_HIT_RETAIN else 0, [rel["ident"]] if rel["ident"] else None]
                v5_rows[(rel["node"], rel["client"])] = row
                rows.append(row)
        out = ""
        for node in sorted(per_node):
            if not per_node[node]:
                continue
            out += f"N {node}\n"
            out += "".join(sorted(f"{c}\t{f}\t{q}\t{rt}\t{','.join(map(str, ids)) if ids else '-'}\n" for c, f, q, rt, ids in per_node[node]))
        return out


def golden_world(kind, seed=77, n_subs=400, n_pub=120):
    """A fixed world for tests/golden/delivery_small.json: (world, publishes) — the oracle's dumps of
    these publishes are what the fixture hashes (tests/golden/make_golden.py)."""
    w = World(kind, seed)
    for _ in range(n_subs):
        w.add(**w.random_sub())
    r = random.Random(seed + 1)
    known = sorted(w.owner_ids)
    pubs = []
    for _ in range(n_pub):
        node, client, ct = r.choice(known) if r.random() < 0.7 else (r.choice(w.nodes), "stranger", 0)
        pubs.append((r.choice(TOPICS), node, client, ct, r.randrange(3), r.random() < 0.5))
    return w, pubs


def golden_dumps(w, pubs, use_backend):
    """Per publish the forwards dump ("" for an invalid topic): from the oracle, or folded from the backend's delivery words."""
    if not use_backend:
        return [w.oracle.forwards(orc.mk_id(n, c, ct), t, q, ret) or "" for t, n, c, ct, q, ret in pubs]
    blob, offs = pack([p[0] for p in pubs])
    attrs = np.zeros(len(pubs), dtype=capi.PUBLISH_ATTR_DTYPE)
    for i, (_, n, c, ct, q, ret) in enumerate(pubs):
        attrs[i] = (w.owner_ids.get((n, c, ct), capi.ID_NONE), q | (4 if ret else 0))
    w.backend.commit()
    got = w.backend.match_batch_deliver(blob, offs, attrs)
    out = []
    for i in range(len(pubs)):
        lo, hi = int(got["hit_offsets"][i]), int(got["hit_offsets"][i + 1])
        out.append("" if got["status"][i] < 0 else w.fold(got["tuples"][lo:hi], i))
    return out


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("seed,kw", [(1, {}), (2, dict(window_hits=64, chunk_topics=16)), (3, dict(slot_cap=2, window_hits=1))])
def test_delivery_stage_matches_oracle_forwards(kind, seed, kw):
    w = World(kind, seed, **kw)
    for _ in range(160):
        w.add(**w.random_sub())
    w.check()
    # churn: re-subscribes (options replaced, possibly with a new Id), removals, new subscriptions;
    # on the HIP backend these go through the incremental commit with the attribute runs
    for rnd in range(4):
        for _ in range(25):
            if w.sub_of and w.rng.random() < 0.45:
                filt, client = w.rng.choice(sorted(w.sub_of))
                w.remove(filt, client, w.rng.choice([0, 0, 5]))
            else:
                w.add(**w.random_sub())
        w.check(40)


@pytest.mark.parametrize("kind", BACKENDS)
def test_v5_dedup_many_candidates(kind):
    """One topic matched by many overlapping filters, every client v5 on all of them: a few
    thousand dedup candidates in one window, first filter in TopicTree::matches order wins."""
    w = World(kind, 9)
    w.clients = [f"k{i}" for i in range(300)]
    w.client_node = {c: w.nodes[i % 3] for i, c in enumerate(w.clients)}
    filters = ["a/b/c", "a/b/+", "a/+/c", "+/b/c", "a/b/#", "a/#", "#", "+/+/+", "+/#"]
    for i, c in enumerate(w.clients):
        for j, f in enumerate(filters):
            if (i + j) % 4 != 3:
                w.add(f, c, 0, (i + j) % 3, True, (i % 5) == 0, (j % 2) == 0, (i * 9 + j) % 200)
    w.check(8)


@pytest.mark.parametrize("test_slots", [0, 64])
@pytest.mark.parametrize("kind", BACKENDS)
def test_v5_dedup_topics_spanning_tiles_in_parts(kind, test_slots, monkeypatch):
    """Topics with ~10 k v5 candidates each: they span several expansion tiles (block-per-topic LDS tables) and exceed one
    table, so they are split into parts by client; with RGR_DEDUP_TEST_SLOTS=64 every part overflows its (shrunken) table
    and the on-the-fly re-split runs.  Small topics in the same window go through the tile tables."""
    if test_slots:
        monkeypatch.setenv("RGR_DEDUP_TEST_SLOTS", str(test_slots))
    w = World(kind, 21)
    w.clients = [f"k{i}" for i in range(1500)]
    w.client_node = {c: w.nodes[i % 3] for i, c in enumerate(w.clients)}
    filters = ["a/b/c", "a/b/+", "a/+/c", "+/b/c", "a/b/#", "a/#", "#", "+/+/+", "+/#"]
    for i, c in enumerate(w.clients):
        for j, f in enumerate(filters):
            if (i + 2 * j) % 5 != 3:
                w.add(f, c, 0, (i + j) % 3, (i + j) % 7 != 0, (i % 5) == 0, (j % 2) == 0, (i * 9 + j) % 200)
    for i, c in enumerate(w.clients[:40]):          # low-fan-out topics: two overlapping filters, a handful of clients
        w.add("q/r/s", c, 0, 1, True, False, False, 0)
        if i % 2:
            w.add("q/r/+", c, 0, 2, True, False, True, 7)
    w.check(14)


@pytest.mark.parametrize("kind", BACKENDS)
def test_v5_dedup_long_runs_with_churn(kind):
    """Runs of thousands of subscribers (longer than an expansion tile, so a topic's largest run spans tiles on its own): overlapping filters
    with thousands of v5 clients each, publishers that hold No Local subscriptions inside the long runs (a dropped hit neither is nor makes a
    duplicate), subscription churn on the long runs through the incremental commit — everything against the oracle's forwards().  (Written
    for round 6's exempt-run experiment, tools/dropped/r6_exempt_runs.diff; kept as the parity world with the longest runs.)"""
    w = World(kind, 33)
    w.clients = [f"k{i}" for i in range(2600)]
    w.client_node = {c: w.nodes[i % 3] for i, c in enumerate(w.clients)}
    filters = ["a/#", "a/b/c", "+/+/+", "a/b/+", "#"]
    for i, c in enumerate(w.clients):
        for j, f in enumerate(filters):
            if j >= 3 and (i + j) % 6:            # "a/b/+" and "#": short runs beside the three long ones
                continue
            if (i + 3 * j) % 11 != 5:
                w.add(f, c, 0, (i + j) % 3, (i + j) % 4 != 0, (i % 3) == 0, (j % 2) == 0, (i * 7 + j) % 150)
    w.check(16)
    for rnd in range(3):
        for i in range(rnd, 2600, 37):
            c = w.clients[i]
            for f in ("a/#", "+/+/+"):
                if (f, c) in w.sub_of and (i + rnd) % 2:
                    w.remove(f, c, 0)
                else:
                    w.add(f, c, 0, (i + rnd) % 3, (i + rnd) % 3 != 0, (i % 2) == 0, rnd % 2 == 0, (i + rnd) % 99)
        w.check(8)


@pytest.mark.gpu
def test_a_client_twice_in_one_run_is_deduplicated_too():
    """A caller whose table breaks the reference's one-relation-per-(filter, client) rule (types.rs:476: the relations of a filter are a map keyed
    by ClientId) — two subscriptions of ONE client on one filter: the device's rule is still "first position of the client wins", inside a run too."""
    r = capi.Router(device=0)
    big, other = r.filter_add("a/#"), r.filter_add("a/b")
    n = 3000
    for i in range(n):
        r.sub_add_ex(big, i, i % 3, capi.RGR_SUB_V5, 0, i % (n - 7), i % (n - 7))        # the last seven clients repeat the first seven
    for i in range(40):
        r.sub_add_ex(other, n + i, 1, capi.RGR_SUB_V5, 0, 5 * i, 5 * i)
    r.commit()
    blob, offs = pack(["a/b", "a/c"])
    attrs = np.zeros(2, dtype=capi.PUBLISH_ATTR_DTYPE)
    attrs["from_id"] = capi.ID_NONE
    attrs["qos_retain"] = 2
    got = r.match_batch_deliver(blob, offs, attrs)
    dup = got["tuples"][(got["tuples"]["qos_flags"] & capi.RGR_HIT_V5_DUP) != 0]
    # topic 0: 3 040 hits of 2 993 distinct clients -> 47 duplicates; topic 1: only the seven repeats inside the long run
    assert (dup["topic_idx"] == 0).sum() == 47 and (dup["topic_idx"] == 1).sum() == 7
    r.close()


SWITCH_SETS = {
    # the kernels that were the defaults until the r5a session measured their replacements (profiles/r05a_ab_*.jsonl); RGR_DELIVER_LEAN=0 or the
    # lean expansion (checked first by launch_expand) would still be the one that runs: this set reaches expand_kernel<true>
    "r4_defaults": {"RGR_DELIVER_LEAN": "0", "RGR_DELIVER_EARLY": "0", "RGR_PREP_BATCH": "0", "RGR_DEDUP_PROBE": "0"},
    # round 5's first default: the delivery expansion with its loads issued early
    "early_loads": {"RGR_DELIVER_LEAN": "0"},
    # the lean expansion's other geometry (256 threads x 8 positions)
    "lean_256x8": {"RGR_DELIVER_LEAN": "2"},
    # measured and not adopted: 2^30-hit delivery windows
    "large_windows": {"RGR_DELIVER_WINDOW_HITS": str(1 << 30)},
    # (r6) the topic pass of the v5 dedup as it was until r6s (lists read 64 entries at a time), on emptier tables
    "topic_pass_r5": {"RGR_DEDUP_PROBE": "3", "RGR_DEDUP_SLOT_FACTOR": "4"},
}


@pytest.mark.gpu
@pytest.mark.parametrize("test_slots", [0, 64])
@pytest.mark.parametrize("switches", sorted(SWITCH_SETS))
def test_v5_dedup_under_the_library_switches(switches, test_slots, monkeypatch):
    """The worlds of the two tests above under the library's environment switches (read per launch / per pass): the round-4 default
    kernels (expand_kernel<true>, count_kernel / compact_kernel one gather at a time, dedup in stream order) and the variants that
    were measured and not adopted."""
    for k, v in SWITCH_SETS[switches].items():
        monkeypatch.setenv(k, v)
    test_v5_dedup_topics_spanning_tiles_in_parts("hip", test_slots, monkeypatch)
    test_v5_dedup_many_candidates("hip")
    if switches in ("lean_256x8", "large_windows") and not test_slots:
        test_v5_dedup_long_runs_with_churn("hip")


@pytest.mark.parametrize("n_nodes", [1, 3, 300])
@pytest.mark.parametrize("kind", BACKENDS)
def test_delivery_grouped_by_node(kind, n_nodes):
    """rgr_match_batch_deliver_grouped: every topic's delivery tuples partitioned by node on the device (SubRelationsMap is keyed
    by node, types.rs:486-497) — the same tuples as the ungrouped call, stably reordered by node index, plus the directory.
    300 nodes need two radix-256 passes; one node needs none."""
    import random
    rng = random.Random(n_nodes)
    b = make_backend(kind, window_hits=900) if kind == "hip" else make_backend(kind, window_hits=900)
    filters = ["a/b/c", "a/+/c", "a/#", "#", "+/b/#", "a/b/+", "x/y", "x/+", "+/+"]
    sid = 0
    for f in filters:
        fid = b.filter_add(f)
        for _ in range(rng.randint(40, 160)):
            node = rng.randrange(n_nodes)
            b.sub_add_ex(fid, sid, rng.randrange(3), capi.RGR_SUB_V5 if rng.random() < 0.4 else 0, node, sid % 97, sid % 53)
            sid += 1
    b.commit()
    topics = ["a/b/c", "x/y", "a/q/c", "nothing/here/at/all/x", "q", "a/b/c/d", "a/b/c"] * 5
    blob, offs = pack(topics)
    attrs = np.zeros(len(topics), dtype=capi.PUBLISH_ATTR_DTYPE)
    attrs["from_id"] = capi.ID_NONE
    attrs["qos_retain"] = 2
    plain = b.match_batch_deliver(blob, offs, attrs)
    got = b.match_batch_deliver(blob, offs, attrs, grouped=True)
    assert np.array_equal(plain["hit_offsets"], got["hit_offsets"]) and np.array_equal(plain["status"], got["status"])
    ho = got["hit_offsets"].astype(np.int64)
    go, gn, gb = got["group_offsets"].astype(np.int64), got["group_node"], got["group_begin"].astype(np.int64)
    assert len(go) == len(topics) + 1 and go[0] == 0 and go[-1] == len(gn) and len(gb) == len(gn) + 1 and gb[-1] == len(got["tuples"])
    for t in range(len(topics)):
        a, e = ho[t], ho[t + 1]
        ref = plain["tuples"][a:e]
        exp = ref[np.argsort(ref["qos_flags"] >> 16, kind="stable")]
        assert np.array_equal(got["tuples"][a:e], exp), t
        nodes = got["tuples"]["qos_flags"][a:e] >> 16
        groups = list(range(go[t], go[t + 1]))
        assert (len(groups) == 0) == (e == a)
        assert list(gn[groups]) == sorted(set(nodes.tolist()))
        for g in groups:
            lo, hi = gb[g], gb[g + 1] if g + 1 < go[t + 1] else e
            assert a <= lo < hi <= e and (nodes[lo - a:hi - a] == gn[g]).all()
    assert len(set(gn.tolist())) == min(n_nodes, len(set(gn.tolist()))) and (n_nodes == 1) == (set(gn.tolist()) <= {0})


@pytest.mark.parametrize("kind", BACKENDS)
def test_delivery_without_registered_ids(kind):
    """Plain rgr_sub_add (no owner / client ids): qos downgrade and RAP still apply, No Local
    and the v5 dedup cannot (nothing to compare) and every hit is delivered unflagged."""
    b = make_backend(kind)
    fid = b.filter_add("t/+")
    b.sub_add(fid, 0, 2, 0)
    b.sub_add(fid, 1, 1, capi.RGR_SUB_V5 | capi.RGR_SUB_RAP | capi.RGR_SUB_NO_LOCAL)
    fid2 = b.filter_add("t/#")
    b.sub_add(fid2, 2, 2, capi.RGR_SUB_V5)
    b.commit()
    blob, offs = pack(["t/x", "t/y"])
    attrs = np.array([(capi.ID_NONE, 1 | 4), (5, 0)], dtype=capi.PUBLISH_ATTR_DTYPE)
    got = b.match_batch_deliver(blob, offs, attrs)
    words = {(int(t["topic_idx"]), int(t["sub_id"])): int(t["qos_flags"]) & 0xFF for t in got["tuples"]}
    assert words == {(0, 2): 1, (0, 0): 1, (0, 1): 1 | capi.RGR_HIT_RETAIN, (1, 2): 0, (1, 0): 0, (1, 1): 0}


# ---- property-based: arbitrary small worlds through the emulator (CPU) --------------------------
from hypothesis import HealthCheck, given, settings  # noqa: E402
from hypothesis import strategies as st  # noqa: E402

_LV = st.sampled_from(["a", "b", "+", "#", "$s", ""])
_NAME = st.lists(_LV, min_size=1, max_size=4).map("/".join)
_SUB = st.tuples(_NAME, st.integers(0, 5), st.sampled_from([0, 0, 9]), st.integers(0, 2), st.booleans(), st.booleans(), st.booleans(),
                 st.integers(0, 3))
_PUB = st.tuples(_NAME, st.integers(0, 6), st.sampled_from([0, 0, 9]), st.integers(0, 2), st.booleans())


@settings(max_examples=300, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow])
@given(subs=st.lists(_SUB, min_size=0, max_size=30), pubs=st.lists(_PUB, min_size=1, max_size=12),
       window=st.sampled_from([0, 1, 5]), slot_cap=st.sampled_from([0, 1, 2]))
def test_delivery_stage_property(subs, pubs, window, slot_cap):
    w = World("emu", 0, window_hits=window, slot_cap=slot_cap, tile=4)
    w.clients = [f"c{i}" for i in range(7)]
    w.client_node = {c: w.nodes[i % 3] for i, c in enumerate(w.clients)}
    for filt, ci, ct, qos, v5, nl, rap, ident in subs:
        if orc.parse_topic(filt) is None:
            continue
        w.add(filt, w.clients[ci], ct, qos, v5, v5 and nl, v5 and rap, ident if v5 else 0)
    blob, offs = pack([p[0] for p in pubs])
    attrs = np.zeros(len(pubs), dtype=capi.PUBLISH_ATTR_DTYPE)
    ids = []
    for i, (_, ci, ct, q, ret) in enumerate(pubs):
        client = w.clients[ci]
        node = w.client_node[client]
        ids.append((node, client, ct))
        attrs[i] = (w.owner_ids.get((node, client, ct), capi.ID_NONE), q | (4 if ret else 0))
    got = w.backend.match_batch_deliver(blob, offs, attrs)
    for i, (topic, _, _, q, ret) in enumerate(pubs):
        exp = w.oracle.forwards(orc.mk_id(*ids[i]), topic, q, ret)
        lo, hi = int(got["hit_offsets"][i]), int(got["hit_offsets"][i + 1])
        if exp is None:
            assert got["status"][i] < 0 and lo == hi
        else:
            assert w.fold(got["tuples"][lo:hi], i) == exp, (topic, ids[i])


@pytest.mark.parametrize("kind", BACKENDS)
def test_oracle_delivery_digest_equals_the_delivery_words(kind):
    """bench.py checks the delivery stage at full size through DefaultRouter::deliver_digest (oracle.hpp): per publish a digest of the
    per-hit verdicts, where WHICH hits are delivered comes from the oracle's matches() (the restated _matches + collector).  Here the
    same digest is computed from a backend's delivery words (emu on the CPU, the kernels under -m gpu) on a table with v5 / No Local /
    Retain-As-Published subscriptions — so the digest definition is pinned on the words that the text-level test above pins on forwards()."""
    rng = np.random.default_rng(77)
    c = wl.CONFIGS[3]
    n_sub = 12_000
    blob, offs, client, qos = wl.gen_subs(n_sub, wl.SUB_SEED + 3, c["p_plus"], c["p_hash"], c["p_sys"])
    is5 = rng.random(n_sub) < 0.35
    flags = (is5 * capi.RGR_SUB_V5 | (is5 & (rng.random(n_sub) < 0.4)) * capi.RGR_SUB_NO_LOCAL | (is5 & (rng.random(n_sub) < 0.5)) * capi.RGR_SUB_RAP).astype(np.uint8)
    o = orc.DefaultRouter()
    assert o.add_bulk_ex(blob, offs, client, qos, flags) == 0
    b = make_backend(kind)
    raw = bytes(blob)
    for i in range(n_sub):                                                        # owner id == client index: Id{node 1, "c<j>"}
        fid = b.filter_add(raw[int(offs[i]):int(offs[i + 1])].decode())
        b.sub_add_ex(fid, i, int(qos[i]), int(flags[i]), 0, int(client[i]), int(client[i]))
    b.commit()
    tb, to = wl.gen_topics(1_200, wl.PUB_SEED + 3, 0.01, c["p_blank"])
    n = len(to) - 1
    attrs = np.zeros(n, dtype=capi.PUBLISH_ATTR_DTYPE)
    attrs["from_id"] = rng.choice(client.astype(np.uint32), size=n)
    attrs["from_id"][::17] = capi.ID_NONE                                         # publishers the table does not know
    attrs["qos_retain"] = rng.integers(0, 3, size=n) | (rng.integers(0, 2, size=n) << 2)
    got = b.match_batch_deliver(tb, to, attrs)
    st, exp = o.deliver_digest(tb, to, attrs["from_id"], attrs["qos_retain"].astype(np.uint8), threads=3)
    assert np.array_equal(st < 0, got["status"] < 0)
    ho = got["hit_offsets"].astype(np.int64)
    M = (1 << 64) - 1
    n_dup = n_drop = 0
    for i in range(n):
        t = got["tuples"][ho[i]:ho[i + 1]]
        x = [int(s) * 32 + (int(w) & 31) for s, w in zip(t["sub_id"], t["qos_flags"])]
        d = (len(x), sum(x) & M, sum((k + 1) * v for k, v in enumerate(x)) & M, sum(v * v for v in x) & M)
        assert tuple(int(z) for z in exp[i]) == d, i
        n_dup += sum(1 for w in t["qos_flags"] if int(w) & capi.RGR_HIT_V5_DUP)
        n_drop += sum(1 for w in t["qos_flags"] if int(w) & capi.RGR_HIT_NO_LOCAL)
    assert n_dup > 50 and n_drop > 5 and int(exp[:, 0].sum()) > 20_000
    # the reference-shaped timed pass (cpu_baseline of the delivery record) delivers exactly the hits that are neither dropped nor duplicates
    sec, stt = o.forwards_timed(tb, to, attrs["from_id"], attrs["qos_retain"].astype(np.uint8), threads=2)
    assert stt["hits"] == int(exp[:, 0].sum()) and stt["rows"] == int(exp[:, 0].sum()) - n_dup - n_drop and sec > 0


@pytest.mark.gpu
@pytest.mark.parametrize("window_hits", [0, 5000])
def test_deliver8_windows_equal_the_delivery_tuples(window_hits):
    """RGR_FORMAT_DELIVER8 (r6): a device-resident delivery pass answers with 8-byte hits {sub_id, delivery word}; window for window they are the
    12-byte delivery tuples without the topic column (which the CSR offsets imply) — v5 duplicates and No Local drops included.
    The format is the delivery stage's own: without publish attributes it is refused, detaching them returns the batch to tuples."""
    rng = np.random.default_rng(5)
    r = capi.Router(device=0, window_hits=window_hits)
    filters = ["a/#", "a/b/c", "+/+/+", "a/b/+", "#", "x/y"]
    sid = 0
    for j, f in enumerate(filters):
        fid = r.filter_add(f)
        for c in rng.choice(4000, size=[3000, 2500, 2200, 300, 40, 9][j], replace=False):
            v5 = rng.random() < 0.5
            flags = (capi.RGR_SUB_V5 | (capi.RGR_SUB_NO_LOCAL if rng.random() < 0.3 else 0) | (capi.RGR_SUB_RAP if rng.random() < 0.5 else 0)) if v5 else 0
            r.sub_add_ex(fid, sid, int(rng.integers(0, 3)), flags, int(c) % 3, int(c), int(c))
            sid += 1
    r.commit()
    topics = ["a/b/c", "x/y", "a/b", "nothing", "a/b/c", "q/r/s", "a/b/c/d"] * 3
    blob, offs = pack(topics)
    b = r.batch(blob, offs)
    with pytest.raises(capi.RgrError):
        b.set_format(capi.RGR_FORMAT_DELIVER8)               # not without publish attributes
    attrs = np.zeros(len(topics), dtype=capi.PUBLISH_ATTR_DTYPE)
    attrs["from_id"] = rng.integers(0, 4000, size=len(topics))
    attrs["qos_retain"] = rng.integers(0, 3, size=len(topics)) | (rng.integers(0, 2, size=len(topics)) << 2)
    b.set_publish_attrs(attrs)

    def windows(fmt):
        out = []
        b.set_format(fmt)
        b.begin()
        while True:
            w = b.next_window()
            if w is None:
                return out
            nh = int(w.n_hits)
            offs_ = np.zeros(w.topic_end - w.topic_begin + 1, dtype=np.uint64)
            # (rgr_window_to_host synchronises the batch's stream: the window's expansion and dedup are stream-ordered, not finished, when
            # rgr_batch_next_window returns; without host_tuples it fetches the offsets only, in any format)
            assert capi.lib().rgr_window_to_host(b._b, C.byref(w), None, offs_.ctypes.data) == 0
            if fmt == capi.RGR_FORMAT_DELIVER8:
                assert not w.d_tuples
                h = capi.device_to_host(w.d_hits8, nh * 8).view(np.dtype([("sub_id", np.uint32), ("word", np.uint32)])) if nh else None
            else:
                assert not w.d_hits8
                h = capi.device_to_host(w.d_tuples, nh * 12).view(capi.TUPLE_DTYPE) if nh else None
            out.append((int(w.topic_begin), int(w.topic_end), nh, offs_.copy(), None if h is None else h.copy()))
    ref = windows(capi.RGR_FORMAT_TUPLE)
    got = windows(capi.RGR_FORMAT_DELIVER8)
    assert len(ref) == len(got) and sum(x[2] for x in ref) > 40000
    seen_dup = seen_drop = False
    for (tb0, te0, n0, o0, t), (tb1, te1, n1, o1, h) in zip(ref, got):
        assert (tb0, te0, n0) == (tb1, te1, n1) and np.array_equal(o0, o1)
        if n0:
            assert np.array_equal(t["sub_id"], h["sub_id"]) and np.array_equal(t["qos_flags"], h["word"])
            assert np.array_equal(t["topic_idx"], np.repeat(np.arange(tb0, te0, dtype=np.uint32), np.diff(o0).astype(np.int64)))
            seen_dup |= bool((h["word"] & capi.RGR_HIT_V5_DUP).any()); seen_drop |= bool((h["word"] & capi.RGR_HIT_NO_LOCAL).any())
    assert seen_dup
    del seen_drop            # (a No Local drop needs the publisher among the subscribers: not guaranteed by this draw)
    # the same pass in walk order (rgr_batch_set_order): the attributes follow their topics, every topic's hits are what they were
    per_topic = {}
    for tb1, te1, _, o1, h in got:
        for k in range(te1 - tb1):
            per_topic[tb1 + k] = h[int(o1[k]):int(o1[k + 1])] if h is not None else np.zeros(0, dtype=np.dtype([("sub_id", np.uint32), ("word", np.uint32)]))
    b.set_format(capi.RGR_FORMAT_DELIVER8)
    b.set_order(True)
    perm = b.topic_order()
    assert sorted(perm.tolist()) == list(range(len(topics)))
    walked = windows(capi.RGR_FORMAT_DELIVER8)
    seen = 0
    for tb1, te1, _, o1, h in walked:
        for k in range(te1 - tb1):
            a, e = int(o1[k]), int(o1[k + 1])
            assert np.array_equal(h[a:e] if h is not None else per_topic[int(perm[tb1 + k])][:0], per_topic[int(perm[tb1 + k])]), (tb1 + k, perm[tb1 + k])
            seen += 1
    assert seen == len(topics)
    b.set_order(False)
    with pytest.raises(capi.RgrError):
        b.set_format(capi.RGR_FORMAT_PACKED)                 # plain compact formats carry no delivery word
    b.set_publish_attrs(None)                                # detaching the attributes: back to tuples
    b.begin()
    w = b.next_window()
    assert w.d_tuples and not w.d_hits8
    while b.next_window() is not None:
        pass
    b.close(); r.close()

