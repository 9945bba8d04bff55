"""Property-based parity (hypothesis): arbitrary filter sets / topics over a hostile alphabet —
multi-byte UTF-8 levels, '$' in every position, empty levels, wildcards in publish topics,
near-duplicate filters — through the host emulator (the kernels' own per-lane code + the
product's table compiler; CPU) and through the HIP path (`-m gpu`, fewer examples) against the
oracle and the brute-force matcher."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import brute
from oracle import oracle as orc
from tests.emu import emu
from tests.parity import compare_flat, pack

LEVELS = st.sampled_from(["a", "b", "", "$", "$SYS", "é", "日本", "a b", "A", "+", "#", "a+", "#x", "x$"])
TOPIC = st.lists(LEVELS, min_size=1, max_size=6).map("/".join)


@settings(max_examples=600, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow])
@given(filters=st.lists(TOPIC, min_size=0, max_size=25), topics=st.lists(TOPIC, min_size=1, max_size=25),
       slot_cap=st.sampled_from([0, 1, 2]), window=st.sampled_from([0, 1, 7]), lds=st.sampled_from([0, 3, 2560]))
def test_router_parity_property(filters, topics, slot_cap, window, lds):
    _router_property(filters, topics, slot_cap, window, lds)


WILD = st.lists(st.sampled_from(["a", "b", "+", "#", "$s", ""]), min_size=1, max_size=5).map("/".join)


@settings(max_examples=500, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow])
@given(filters=st.lists(WILD, min_size=0, max_size=30), topics=st.lists(WILD, min_size=1, max_size=25),
       slot_cap=st.sampled_from([0, 1, 2]), window=st.sampled_from([0, 1, 7]), lds=st.sampled_from([0, 3, 2560]))
def test_router_parity_property_wildcard_heavy(filters, topics, slot_cap, window, lds):
    """Dense wildcard alphabet: overlapping filters, wildcard levels inside PUBLISH topics (the
    double-visit quirk of trie.rs:327-375), '$' and blank levels in every position."""
    _router_property(filters, topics, slot_cap, window, lds)


def _router_property(filters, topics, slot_cap, window, lds):
    o = orc.DefaultRouter()
    e = emu.EmuRouter(slot_cap=slot_cap, window_hits=window, lds_window=lds, tile=4)
    sub = 0
    for f in filters:
        ok = o.add(f, orc.mk_id(1, f"c{sub}"), orc.mk_opts(qos=sub % 3), rel_id=sub) == 0
        try:
            fid = e.filter_add(f)
            assert ok, f"emu accepted a filter the reference rejects: {f!r}"
            e.sub_add(fid, sub, sub % 3)
        except ValueError:
            assert not ok, f"emu rejected a filter the reference accepts: {f!r}"
        sub += 1
    blob, offs = pack(topics)
    exp = o.match_flat(blob, offs)
    got = e.match_batch(blob, offs)
    compare_flat(got, exp)
    # independent cross-check on wildcard-free valid topics
    valid_filters = [(i, f) for i, f in enumerate(filters) if brute.valid(f)]
    for ti, t in enumerate(topics):
        if not brute.valid(t) or any(l in ("+", "#") for l in t.split("/")):
            continue
        ids = sorted(got["tuples"]["sub_id"][int(got["hit_offsets"][ti]):int(got["hit_offsets"][ti + 1])].tolist())
        assert ids == sorted(i for i, f in valid_filters if brute.filter_matches(f, t)), (t, filters)


@settings(max_examples=600, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow])
@given(topics=st.lists(TOPIC, min_size=0, max_size=25, unique=True), filters=st.lists(TOPIC, min_size=1, max_size=20),
       removes=st.lists(st.integers(0, 24), max_size=6))
def test_retain_parity_property(topics, filters, removes):
    _retain_property(topics, filters, removes)


# a small alphabet dominated by wildcard levels: literal "+" / "#" levels stored in the tree and the
# exact-first branch of retain.rs:472 (also inside the '#' recursion) come up in most examples
WILD_TOPIC = st.lists(st.sampled_from(["a", "b", "+", "#", "$s", ""]), min_size=1, max_size=5).map("/".join)


@settings(max_examples=500, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow])
@given(topics=st.lists(WILD_TOPIC, min_size=0, max_size=30, unique=True), filters=st.lists(WILD_TOPIC, min_size=1, max_size=20),
       removes=st.lists(st.integers(0, 29), max_size=6))
def test_retain_parity_property_wildcard_heavy(topics, filters, removes):
    _retain_property(topics, filters, removes)


def _retain_property(topics, filters, removes):
    t = orc.RetainTree()
    e = emu.EmuRouter(window_hits=3, tile=4)
    for i, s in enumerate(topics):
        ok = t.insert(s, i) == 0
        assert (e.retain_add(s, i) == 0) == ok, s
    for r in removes:
        if r < len(topics):
            a = t.remove(topics[r])[0]
            b = e.retain_remove(topics[r])
            assert (a == 1) == (b == 0), topics[r]
    blob, offs = pack(filters)
    got = e.retain_match_batch(blob, offs)
    st_, eo, ev, _ = t.match_batch(blob, offs)
    assert np.array_equal(got["status"] < 0, st_ < 0)
    assert np.array_equal(got["hit_offsets"], eo)
    for a, b_ in zip(eo[:-1], eo[1:]):
        assert sorted(got["topic_ids"][int(a):int(b_)].tolist()) == sorted(ev[int(a):int(b_)].tolist())


@pytest.mark.gpu
@settings(max_examples=80, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow])
@given(filters=st.lists(TOPIC, min_size=0, max_size=25), topics=st.lists(TOPIC, min_size=1, max_size=25),
       retained=st.lists(TOPIC, min_size=0, max_size=20, unique=True), slot_cap=st.sampled_from([0, 1, 2]))
def test_hip_parity_property(filters, topics, retained, slot_cap):
    """The same properties through the real kernels and the C ABI (device tokeniser included)."""
    from rmqtt_amd import capi
    o = orc.DefaultRouter()
    r = capi.Router(device=0, slot_cap=slot_cap, window_hits=7)
    sub = 0
    for f in filters:
        ok = o.add(f, orc.mk_id(1, f"c{sub}"), orc.mk_opts(qos=sub % 3), rel_id=sub) == 0
        try:
            fid = r.filter_add(f)
            assert ok
            r.sub_add(fid, sub, sub % 3)
        except capi.RgrError:
            assert not ok
        sub += 1
    r.commit()
    blob, offs = pack(topics)
    compare_flat(r.match_batch(blob, offs), o.match_flat(blob, offs))
    t = orc.RetainTree()
    for i, s_ in enumerate(retained):
        ok = t.insert(s_, i) == 0
        assert (r.retain_add(s_, i) == 0) == ok
    r.retain_commit()
    got = r.retain_match_batch(blob, offs)
    st_, eo, ev, _ = t.match_batch(blob, offs)
    assert np.array_equal(got["status"] < 0, st_ < 0) and np.array_equal(got["hit_offsets"], eo)
    for a, b_ in zip(eo[:-1], eo[1:]):
        assert sorted(got["topic_ids"][int(a):int(b_)].tolist()) == sorted(ev[int(a):int(b_)].tolist())
    r.close()


_OP = st.tuples(st.booleans(), WILD, st.integers(0, 5))


@settings(max_examples=400, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow])
@given(ops=st.lists(_OP, min_size=1, max_size=60), topics=st.lists(WILD, min_size=1, max_size=20), slot_cap=st.sampled_from([0, 1, 2]))
def test_router_churn_property(ops, topics, slot_cap):
    _router_churn(ops, topics, slot_cap)


def _router_churn(ops, topics, slot_cap):
    """Arbitrary interleavings of subscribe / unsubscribe (trie insert, remove + prune, tombstones,
    filter-id and node reuse) and then a match: trie.rs:113-149, router.rs:434-496."""
    o = orc.DefaultRouter()
    e = emu.EmuRouter(slot_cap=slot_cap, window_hits=3, tile=4)
    subs, fids, refs = {}, {}, {}
    nxt = 0
    for add, f, ci in ops:
        client = f"c{ci}"
        if add:
            if (f, client) in subs:
                continue
            ok = o.add(f, orc.mk_id(1, client), orc.mk_opts(qos=ci % 3), rel_id=nxt) == 0
            if not ok:
                with pytest.raises(ValueError):
                    e.filter_add(f)
                continue
            fid = e.filter_add(f)
            e.sub_add(fid, nxt, ci % 3)
            subs[(f, client)] = nxt
            fids[f] = fid
            refs[f] = refs.get(f, 0) + 1
            nxt += 1
        elif (f, client) in subs:
            assert o.remove(f, orc.mk_id(1, client)) == 0
            assert e.sub_remove(fids[f], subs.pop((f, client))) == 0
            refs[f] -= 1
            if refs[f] == 0:
                assert e.filter_remove(fids[f]) == 0
                del fids[f], refs[f]
    blob, offs = pack(topics)
    compare_flat(e.match_batch(blob, offs), o.match_flat(blob, offs))
    assert e.counters()["n_filters"] == len(fids) and e.counters()["n_subs"] == len(subs)
