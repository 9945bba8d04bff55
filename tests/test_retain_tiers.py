"""Two-tier retained set: immutable compiled base + small delta + dead bits.

The tier bookkeeping (`TieredRetain`, rmqtt_amd/csrc/retain.cpp) and the merge of the two tiers'
answers (`merge_tier_hits`) are host code shared by the product and the emulator; these tests drive
them through the emulator against the oracle's RetainTree under random add / replace / remove /
commit sequences, with merges forced by a small delta limit."""
import random
import threading

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import oracle as orc
from tests.emu import emu
from tests.parity import pack

FILTERS = ["#", "+/#", "a/#", "a/+/#", "+/+", "a/b/#", "+/b/+", "$s/#", "a/+", "+", "a/b/c", "+/+/+/#", "b/#", "a/b", "/+"]


def check(e, t, filters=FILTERS):
    blob, offs = pack(filters)
    got = e.tier_match_batch(blob, offs)
    st_, eo, ev, _ = t.match_batch(blob, offs)
    assert np.array_equal(got["status"] < 0, st_ < 0)
    assert np.array_equal(got["hit_offsets"], eo)
    for a, b in zip(eo[:-1], eo[1:]):
        assert sorted(got["topic_ids"][int(a):int(b)].tolist()) == sorted(ev[int(a):int(b)].tolist())


@pytest.mark.parametrize("delta_max", [0, 5, 40, 10**9])
def test_tiers_random_churn(delta_max):
    rng = random.Random(delta_max + 1)
    e = emu.EmuRouter(window_hits=7, tile=4)
    t = orc.RetainTree()
    levels = ["a", "b", "c", "", "$s", "+", "#"]
    live = {}
    next_id = 0
    for rnd in range(60):
        for _ in range(rng.randint(1, 25)):
            r = rng.random()
            if r < 0.55 or not live:
                n = rng.randint(1, 4)
                name = "/".join(rng.choice(levels[:5] if (rng.random() < 0.85 or delta_max == 10**9) else levels) for _ in range(n))
                if orc.parse_topic(name) is None:
                    assert e.tier_add(name, 1) != 0
                    continue
                assert e.tier_add(name, next_id) == 0            # new topic, or value replaced (retain.rs:384)
                t.insert(name, next_id)
                live[name] = next_id
                next_id += 1
            elif r < 0.65:
                name = rng.choice(sorted(live))                   # re-published under the same id: not a change
                assert e.tier_add(name, live[name]) == 0
            else:
                name = rng.choice(sorted(live))
                assert e.tier_remove(name) == 0
                assert e.tier_remove(name) != 0                   # already gone
                t.remove(name)
                del live[name]
        e.tier_commit(delta_max)
        check(e, t)
        c = e.tier_counters()
        assert c["n_topics"] == len(live) == t.values_size()
        if delta_max < 10**9:
            assert c["n_delta"] <= max(delta_max, 0) or c["merges"] > 0
    c = e.tier_counters()
    if delta_max == 10**9:      # (this variant uses no wildcard-level names, which suspend tiering)
        assert c["merges"] <= 3 and c["delta_compiles"] > 10      # after the first commit: delta + dead bits, a merge only when > 25 % of the base died
    if delta_max == 0:
        assert c["merges"] > 20                                   # every commit that left topics in the delta merged


def test_dead_fraction_forces_a_merge():
    e = emu.EmuRouter()
    t = orc.RetainTree()
    names = [f"k/{i}/v" for i in range(20000)]
    for i, s in enumerate(names):
        assert e.tier_add(s, i) == 0
        t.insert(s, i)
    e.tier_commit(10**9)
    assert e.tier_counters()["merges"] == 1
    for s in names[:3000]:
        assert e.tier_remove(s) == 0
        t.remove(s)
    e.tier_commit(10**9)
    assert e.tier_counters()["merges"] == 1 and e.tier_counters()["n_dead"] == 3000    # below 25 %: flagged, not merged
    check(e, t, ["k/#", "k/+/v", "k/5/v", "k/2999/v", "k/3000/v"])
    for s in names[3000:9000]:
        assert e.tier_remove(s) == 0
        t.remove(s)
    e.tier_commit(10**9)
    assert e.tier_counters()["merges"] == 2 and e.tier_counters()["n_dead"] == 0
    check(e, t, ["k/#", "k/+/v", "k/8999/v", "k/9000/v"])


_LV = st.sampled_from(["a", "b", "+", "#", "$s", ""])
_NAME = st.lists(_LV, min_size=1, max_size=4).map("/".join)
_OPS = st.lists(st.tuples(st.sampled_from(["add", "add", "readd", "remove", "commit"]), _NAME), min_size=1, max_size=60)


@settings(max_examples=400, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow])
@given(ops=_OPS, delta_max=st.sampled_from([0, 1, 3, 1000]), filters=st.lists(_NAME, min_size=1, max_size=12))
def test_tiers_property(ops, delta_max, filters):
    tier_property(ops, delta_max, filters)


def tier_property(ops, delta_max, filters):
    e = emu.EmuRouter(window_hits=3, tile=4)
    t = orc.RetainTree()
    live = {}
    next_id = 0
    for op, name in ops:
        if op == "commit":
            e.tier_commit(delta_max)
            check(e, t, filters)
        elif op == "remove":
            rc = e.tier_remove(name)
            assert (rc == 0) == (name in live)
            if rc == 0:
                t.remove(name)
                del live[name]
        else:
            tid = live[name] if (op == "readd" and name in live) else next_id
            ok = orc.parse_topic(name) is not None
            assert (e.tier_add(name, tid) == 0) == ok
            if ok:
                t.insert(name, tid)
                live[name] = tid
                next_id += 1
    e.tier_commit(delta_max)
    check(e, t, filters)


# ---- the same churn through the C ABI on the GPU (rgr_config.retain_delta_max > 0)
@pytest.mark.gpu
@pytest.mark.parametrize("delta_max", [1, 40, 10**6])
def test_tiers_random_churn_hip(delta_max):
    from rmqtt_amd import capi
    rng = random.Random(delta_max + 7)
    r = capi.Router(device=0, retain_delta_max=delta_max, window_hits=64)
    t = orc.RetainTree()
    levels = ["a", "b", "c", "", "$s", "+", "#"]
    live = {}
    next_id = 0
    blob, offs = pack(FILTERS)
    for rnd in range(40):
        for _ in range(rng.randint(1, 25)):
            x = rng.random()
            if x < 0.6 or not live:
                n = rng.randint(1, 4)
                name = "/".join(rng.choice(levels[:5] if (rng.random() < 0.9 or delta_max == 10**6) else levels) for _ in range(n))
                if orc.parse_topic(name) is None:
                    continue
                assert r.retain_add(name, next_id) == 0
                t.insert(name, next_id)
                live[name] = next_id
                next_id += 1
            else:
                name = rng.choice(sorted(live))
                assert r.retain_remove(name) == 0
                t.remove(name)
                del live[name]
        r.retain_commit()
        got = r.retain_match_batch(blob, offs)
        st_, eo, ev, _ = t.match_batch(blob, offs)
        assert np.array_equal(got["hit_offsets"], eo)
        for a, b in zip(eo[:-1], eo[1:]):
            assert sorted(got["topic_ids"][int(a):int(b)].tolist()) == sorted(ev[int(a):int(b)].tolist())
        # the dense answer (ranges of the host-mirrored value arrays of both tiers, dead entries flagged) resolves to the same hits
        rg = r.retain_match_ranges(blob, offs)
        assert np.array_equal(rg["status"] < 0, st_ < 0) and np.array_equal(rg["hit_offsets"], eo)
        assert np.array_equal(rg["topic_ids"], got["topic_ids"])             # same order as rgr_retain_match_batch, too
        st = r.stats()
        assert st["retain_topics"] == len(live)
    st = r.stats()
    if delta_max == 10**6:
        assert st["retain_merges"] <= 3 and st["retain_delta_topics"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("host_tokenize", [False, True])
def test_tiers_concurrent_commit_and_match_hip(host_tokenize):
    """Queries from several threads while another thread adds / replaces / removes topics and commits:
    no deadlock (lock order retain_mu -> retain_pair_mu, queries never take retain_mu in two-tier mode) and a
    replaced topic is always answered with exactly one of its values — never missing, as
    RetainTree::insert replaces atomically (retain.rs:384)."""
    from rmqtt_amd import capi
    r = capi.Router(device=0, retain_delta_max=10**6, host_tokenize=host_tokenize)
    stable = [f"s/{i}/v" for i in range(300)]
    for i, s in enumerate(stable):
        assert r.retain_add(s, i) == 0
    assert r.retain_add("x/hot", 10_000) == 0
    r.retain_commit()
    blob, offs = pack(["s/#", "x/+", "s/+/v", "x/hot"])
    stop = threading.Event()
    errors = []

    def writer():
        try:
            k = 0
            while not stop.is_set() and k < 60:
                k += 1
                assert r.retain_add("x/hot", 10_000 + k) == 0          # replace: old value dies in the base, new one lives in the delta
                if k % 3 == 0:
                    assert r.retain_add(f"t/{k}", 20_000 + k) == 0
                if k % 7 == 0:
                    assert r.retain_remove(f"t/{k - 4}") in (0, capi.RGR_ENOENT)
                r.retain_commit()
        except Exception as e:      # pragma: no cover
            errors.append(e)
        finally:
            stop.set()

    def reader():
        try:
            while not stop.is_set():
                got = r.retain_match_batch(blob, offs)
                o = got["hit_offsets"]
                assert o[1] - o[0] == len(stable) and o[3] - o[2] == len(stable)
                for a, b in ((o[1], o[2]), (o[3], o[4])):
                    ids = got["topic_ids"][int(a):int(b)].tolist()
                    assert len(ids) == 1 and ids[0] >= 10_000, f"x/hot answered with {ids}"
        except Exception as e:      # pragma: no cover
            errors.append(e)
            stop.set()

    th = [threading.Thread(target=writer)] + [threading.Thread(target=reader) for _ in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in th), "deadlock between rgr_retain_commit and rgr_retain_match_batch"
    assert not errors, errors[0]
    r.close()
