"""RetainTree::matches parity (rmqtt/src/retain.rs:450-526; config 5 of BASELINE.json):
SUBSCRIBE filters against the trie of retained topics.  The reference's result order is
hash-map iteration order, so results compare as sorted sets per filter (App. A.5).
Backends as in test_parity.py: emu (CPU, host logic + index math) and hip (`-m gpu`)."""
import random

import numpy as np
import pytest

from oracle import brute
from oracle import oracle as orc
from rmqtt_amd import workload as wl
from tests.parity import make_backend, pack

BACKENDS = ["emu", pytest.param("hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def kind(request):
    return request.param


def per_filter(res):
    ho = res["hit_offsets"]
    return [sorted(res["topic_ids"][int(a):int(b)].tolist()) for a, b in zip(ho[:-1], ho[1:])]


def check(backend, tree, filters):
    got = backend.retain_match_batch(*pack(filters))
    blob, offs = pack(filters)
    st, eo, ev, _ = tree.match_batch(blob, offs)
    assert np.array_equal(got["status"] < 0, st < 0)
    exp = [sorted(ev[int(a):int(b)].tolist()) for a, b in zip(eo[:-1], eo[1:])]
    assert per_filter(got) == exp
    return exp


def test_reference_retain_vectors(kind):   # retain.rs:609-634
    b = make_backend(kind)
    t = orc.RetainTree()
    vec = [("/iot/b/x", 1), ("/iot/b/y", 2), ("/iot/b/z", 3), ("/iot/b", 123), ("/x/y/z", 4), ("/xx/yy", 9), ("/xx/yy/", 0),
           ("/xx/yy/1", 11), ("/xx/yy/2", 12), ("/xx/yy/3", 13), ("/xx/yy/3/4", 14), ("/xx/yy/3/4/5", 15)]
    for s, v in vec:
        assert b.retain_add(s, v) == 0
        t.insert(s, v)
    b.retain_commit()
    exp = check(b, t, ["/iot/b/y", "/iot/b/+", "/x/y/z", "/xx/yy/+", "/xx/yy/3/+", "/xx/yy/3/4/+", "/xx/yy/1/+", "#", "+/#", "/xx/#",
                       "/xx/yy/#", "/+/+/#", "a/#/b", "", "/"])
    assert exp[:7] == [[2], [1, 2, 3], [4], [0, 11, 12, 13], [14], [15], []]
    assert exp[7] == sorted(v for _, v in vec)


def test_semantics_and_mutation(kind):   # retain.rs:393-413 (remove + prune), 476-481, 502-524, '$' isolation
    b = make_backend(kind)
    t = orc.RetainTree()
    topics = ["a", "a/b", "a/b/c", "a/x", "b", "$SYS/up", "$SYS", "/lead", "a/", "$SYS/a/b", "deep/" + "/".join(["q"] * 60)]
    for i, s in enumerate(topics):
        b.retain_add(s, i); t.insert(s, i)
    b.retain_commit()
    filters = ["a/#", "#", "+", "$SYS/#", "+/#", "a/+", "a/b/c", "$SYS/+", "+/up", "+/+/c", "a/+/c", "a/b/#", "nope/#", "a/b/c/#",
               "deep/" + "/".join(["+"] * 60), "deep/" + "/".join(["q"] * 59) + "/#", "a/b/c/d"]
    exp = check(b, t, filters)
    assert exp[0] == [0, 1, 2, 3, 8] and exp[1] == [0, 1, 2, 3, 4, 7, 8, 10] and exp[2] == [0, 4]
    # replace a value, remove topics (pruning), re-check
    b.retain_add("a/b", 77); t.insert("a/b", 77)
    for s in ["a/b/c", "b", "$SYS/up"]:
        assert b.retain_remove(s) == 0
        assert t.remove(s)[0] == 1
    assert b.retain_remove("a/b/c") != 0 and b.retain_remove("zz/unknown") != 0
    b.retain_commit()
    exp = check(b, t, filters)
    assert exp[0] == [0, 3, 8, 77]


def test_wildcard_levels_stored_in_tree(kind):
    """A retained topic name may syntactically carry '+' / '#' levels (Topic::from_str accepts
    them); the exact-key-first else-if chain of retain.rs:472/483/502 must be reproduced."""
    b = make_backend(kind)
    t = orc.RetainTree()
    for i, s in enumerate(["x/+", "x/y", "x/z", "x/+/k", "x/y/k", "w/#", "w/v", "w/v/u", "#", "p", "+", "+/q", "p/q"]):
        b.retain_add(s, i); t.insert(s, i)
    b.retain_commit()
    check(b, t, ["x/+", "x/+/k", "x/#", "w/#", "w/+", "#", "+", "+/q", "+/#", "x/+/#"])


def test_literal_hash_level_shadows_its_siblings_from_above(kind):
    """retain.rs:472-483 is also hit *during* the '#' recursion (retain.rs:521 recurses with the same
    path): a node storing a literal "#" level answers `["#"]` through the exact branch, so from that
    node or from anywhere above it only the "#" child's value is visible below it — 'a/a' is NOT
    returned for the filter '#' once 'a/#' is stored.  (Found by the hypothesis suite.)"""
    b = make_backend(kind)
    t = orc.RetainTree()
    names = ["a/a", "a/#", "a", "a/b/c", "a/b/#", "a/b/d/e", "x/y", "x/y/z", "x/y/#", "x/q", "$SYS/u", "$SYS/#", "$SYS/u/v",
             "m/n/o/p", "m/n/#", "m/n/o/#", "m/k"]
    for i, s in enumerate(names):
        assert b.retain_add(s, i) == 0
        t.insert(s, i)
    b.retain_commit()
    filters = ["#", "a/#", "a/b/#", "a/b/d/#", "+/#", "a/+/#", "+/+/#", "x/#", "x/y/#", "x/+", "$SYS/#", "$SYS/+/#", "m/#", "m/n/#",
               "m/n/o/#", "m/+/#", "+/n/#", "a/a", "a/b/c", "+/+", "a/+", "m/n/o/p"]
    exp = check(b, t, filters)
    assert exp[0] == sorted(names.index(s) for s in ["a", "a/#", "x/y", "x/q", "x/y/#", "m/k", "m/n/#"])   # hand-derived
    # removing the literal "#" topics un-shadows their siblings
    for s in ["a/#", "m/n/#"]:
        assert b.retain_remove(s) == 0
        t.remove(s)
    b.retain_commit()
    exp = check(b, t, filters)
    assert names.index("a/a") in exp[0] and names.index("m/n/o/#") in exp[0] and names.index("m/n/o/p") not in exp[0]


def test_wide_nodes_long_descriptor_lists(kind):
    """'+' over nodes with hundreds / thousands of children: big frontier expansions and
    descriptor lists far beyond 64 entries per filter (block-cooperative count/compact)."""
    b = make_backend(kind)
    t = orc.RetainTree()
    i = 0
    for a in range(3):
        for k in range(1500 if a == 0 else 40):
            for leaf in (["x"] if k % 3 else ["x", "y/z"]):
                s = f"r{a}/k{k}/{leaf}"
                b.retain_add(s, i); t.insert(s, i); i += 1
        b.retain_add(f"r{a}", i); t.insert(f"r{a}", i); i += 1
    b.retain_commit()
    exp = check(b, t, ["r0/+/x", "r0/+", "r0/+/#", "+/+/x", "+/+/+", "+/+/y/z", "r0/+/y/+", "r1/+/x", "+/k7/x", "#", "r0/#", "+/+/#", "+/#"])
    assert len(exp[0]) == 1500 and len(exp[3]) == 1580


@pytest.mark.parametrize("opts", [dict(), dict(slot_cap=1, chunk_topics=100, window_hits=50, tile=8), dict(slot_cap=2, window_hits=1)])
def test_random_retain_tables(kind, opts):
    rng = random.Random(5)
    alpha = ["a", "b", "c", "d", "", "$s", "ee"]

    def rand(wild, maxd=6):
        n = rng.randint(1, maxd)
        lv = []
        for i in range(n):
            x = rng.random()
            if wild and x < 0.25:
                lv.append("+")
            elif wild and x < 0.45 and i == n - 1:
                lv.append("#")
            else:
                lv.append(rng.choice(alpha if i == 0 else [a for a in alpha if a != "$s"]))
        return "/".join(lv)

    b = make_backend(kind, **opts)
    t = orc.RetainTree()
    topics = sorted({rand(False) for _ in range(1500)})
    for i, s in enumerate(topics):
        b.retain_add(s, i); t.insert(s, i)
    b.retain_commit()
    filters = [rand(True) for _ in range(800)]
    exp = check(b, t, filters)
    for f, e in list(zip(filters, exp))[:300]:     # independent brute-force cross-check
        assert e == [i for i, s in enumerate(topics) if brute.filter_matches(f, s)], f
    for s in topics[::3]:
        assert b.retain_remove(s) == 0 and t.remove(s)[0] == 1
    b.retain_commit()
    check(b, t, filters)


def test_seeded_config5_small(kind):
    """BASELINE.json configs[4] shape at oracle-friendly size: retained topics from the publish
    generator (distinct), filters from the config-3 filter generator forced to hold a wildcard."""
    tb, to = wl.gen_topics(40_000, wl.PUB_SEED + 5, 0.01, 0.01)
    topics = sorted(s for s in set(wl.strings(tb, to)) if brute.valid(s))   # the generator emits a few invalid names ("/$SYS/..")
    fb, fo, _, _ = wl.gen_subs(3_000, wl.SUB_SEED + 5, 0.028, 0.10, 0.005, force_wildcard=True)
    b = make_backend(kind, chunk_topics=1024, window_hits=100_000)
    t = orc.RetainTree()
    blob, offs = pack(topics)
    assert b.retain_add_bulk(blob, offs) == 0
    for i, s in enumerate(topics):
        t.insert(s, i)
    b.retain_commit()
    got = b.retain_match_batch(fb, fo)
    st, eo, ev, _ = t.match_batch(fb, fo)
    assert np.array_equal(got["hit_offsets"], eo)
    assert per_filter(got) == [sorted(ev[int(a):int(b_)].tolist()) for a, b_ in zip(eo[:-1], eo[1:])]
    assert eo[-1] > 10_000


def test_table_version_tracks_effective_changes_only():
    """rgr_retain_commit skips the recompile when RetainTable::version() did not move: re-publishing
    a retained topic under its id is not a change; a new id, a new topic or a removal is."""
    from tests.emu import emu
    e = emu.EmuRouter()
    v0 = e.retain_version()
    assert e.retain_add("a/b", 1) == 0
    v1 = e.retain_version()
    assert v1 != v0
    assert e.retain_add("a/b", 1) == 0 and e.retain_version() == v1          # same topic, same id
    assert e.retain_add("a/#/b", 2) != 0 and e.retain_version() == v1        # rejected name
    assert e.retain_add("a/b", 5) == 0 and e.retain_version() != v1          # value replaced (retain.rs:384)
    v2 = e.retain_version()
    assert e.retain_remove("nope") != 0 and e.retain_version() == v2
    assert e.retain_remove("a/b") == 0 and e.retain_version() != v2


@pytest.mark.gpu
def test_commit_without_changes_keeps_the_epoch():
    from rmqtt_amd import capi
    r = capi.Router(device=0)
    assert r.retain_add("s/t", 3) == 0 and r.retain_add("s/u", 4) == 0
    r.retain_commit()
    e1 = r.stats()["retain_epoch"]
    assert e1 != 0 and r.stats()["retain_topics"] == 2
    r.retain_commit()                                        # nothing changed
    assert r.retain_add("s/t", 3) == 0                       # retained message re-published, same id
    r.retain_commit()
    assert r.stats()["retain_epoch"] == e1
    got = r.retain_match_batch(*pack(["s/+"]))
    assert sorted(got["topic_ids"].tolist()) == [3, 4]
    assert r.retain_add("s/v", 9) == 0
    r.retain_commit()
    assert r.stats()["retain_epoch"] != e1 and r.stats()["retain_topics"] == 3
    assert sorted(r.retain_match_batch(*pack(["s/+"]))["topic_ids"].tolist()) == [3, 4, 9]


def test_repeated_commits_on_one_table(kind):
    """One table compiled again and again while it grows, shrinks and gains / loses literal '#'
    levels: the compile's scratch buffers and the retained image are reused between commits."""
    import random
    rng = random.Random(5)
    b = make_backend(kind)
    t = orc.RetainTree()
    levels = ["a", "b", "c", "d", "", "$s", "+", "#"]
    live = {}
    filters = ["#", "+/#", "a/#", "a/+/#", "+/+", "a/b/#", "+/b/+", "$s/#", "a/+", "+", "a/b/c", "+/+/+/#"]
    next_id = 0
    for rnd in range(30):
        grow = rnd % 7 != 6
        for _ in range(rng.randint(5, 60) if grow else rng.randint(20, 120)):
            if grow or not live:
                n = rng.randint(1, 4)
                name = "/".join(rng.choice(levels[:6] if rng.random() < 0.8 else levels) for _ in range(n))
                if orc.parse_topic(name) is None:
                    continue
                rc = b.retain_add(name, next_id)
                assert rc == 0, name
                t.insert(name, next_id)
                live[name] = next_id
                next_id += 1
            else:
                name = rng.choice(sorted(live))
                assert b.retain_remove(name) == 0
                t.remove(name)
                del live[name]
        b.retain_commit()
        check(b, t, filters)
    assert len(live) > 20


@pytest.mark.gpu
def test_dense_range_answer_equals_the_hit_list():
    """rgr_retain_match_ranges (SURVEY 8(a) `get_message` consumers need topic ids only): per filter a short list of ranges of the
    preorder value array, mirrored on the host — `a/#` is one range.  Resolved through the mirror it must be rgr_retain_match_batch's
    answer (same ids, same order) and the oracle's RetainTree::matches as a set; and it must be SHORT: far fewer ranges than hits."""
    from rmqtt_amd import capi
    c = wl.CONFIGS[5]
    blob, offs = wl.gen_topics(60_000, wl.PUB_SEED + 5, 0.01, c["p_blank"], c["fixed_depth"], distinct=True)
    fb, fo, _, _ = wl.gen_subs(1_500, wl.SUB_SEED + 5, c["p_plus"], c["p_hash"], c["p_sys"], 0, c["fixed_depth"], force_wildcard=True)
    r = capi.Router(device=0, window_hits=20_000)                   # several windows per pass
    t = orc.RetainTree()
    assert r.retain_add_bulk(blob, offs) == t.insert_bulk(blob, offs)
    r.retain_commit()
    extra = pack(["#", "+/#", "$SYS/#", "l0x0/#", "+/+/+", "bad/#/x", "nope/#", "/#"])
    for b_, o_ in ((fb, fo), extra):
        ref = r.retain_match_batch(b_, o_)
        rg = r.retain_match_ranges(b_, o_)
        assert np.array_equal(rg["status"], ref["status"]) and np.array_equal(rg["hit_offsets"], ref["hit_offsets"])
        assert np.array_equal(rg["topic_ids"], ref["topic_ids"])
        st, eo, ev, _ = t.match_batch(b_, o_)
        assert np.array_equal(st < 0, rg["status"] < 0) and np.array_equal(eo, rg["hit_offsets"])
        for a, b in zip(eo[:-1], eo[1:]):
            assert sorted(rg["topic_ids"][int(a):int(b)].tolist()) == sorted(ev[int(a):int(b)].tolist())
        assert rg["n_entries"] == len(ref["topic_ids"])
    assert rg["n_ranges"] * 20 < rg["n_entries"]                     # '#' near the root: tens of thousands of hits in a handful of ranges
    # results stay valid across a later commit (they hold the mirror they index), and a new commit is seen by the next call
    assert r.retain_add("brand/new/topic", 999_999) == 0
    r.retain_commit()
    rg2 = r.retain_match_ranges(*pack(["brand/#", "#"]))
    assert rg2["topic_ids"][:1].tolist() == [999_999] and int(rg2["hit_offsets"][2] - rg2["hit_offsets"][1]) == int(eo[1] - eo[0]) + 1
    r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("delta_max", [0, 50])
def test_device_resident_windows_with_positions_and_packed_reads(delta_max, monkeypatch):
    """(r6) Where the retained path's tuples take their ids from (kernels.hpp launch_expand): the 8-byte value entries (RGR_RETAIN_PACKED_READS=0,
    as until round 5), the packed 4-byte side array of a single-tier epoch (the default there), or nothing at all — rgr_batch_set_retain_positions:
    the tuple carries the hit's POSITION in the epoch's preorder value array and the caller resolves it through rgr_batch_retain_vals' mirror.
    All three must describe the same hits, window for window; in two-tier mode (delta_max > 0) the dead bit of a position form hit is the
    mirror's flag word."""
    from rmqtt_amd import capi
    c = wl.CONFIGS[5]
    blob, offs = wl.gen_topics(40_000, wl.PUB_SEED + 5, 0.01, c["p_blank"], c["fixed_depth"], distinct=True)
    fb, fo, _, _ = wl.gen_subs(600, wl.SUB_SEED + 5, c["p_plus"], c["p_hash"], c["p_sys"], 0, c["fixed_depth"], force_wildcard=True)
    r = capi.Router(device=0, window_hits=30_000, retain_delta_max=delta_max)
    ids = np.random.default_rng(3).permutation(40_000).astype(np.uint32) + 7          # caller ids: neither dense from 0 nor in preorder
    r.retain_add_bulk(blob, offs, ids)
    r.retain_commit()
    if delta_max:                                   # some base topics replaced / removed after the base tier was compiled: dead entries
        names = [bytes(blob[int(offs[i]):int(offs[i + 1])]).decode() for i in range(0, 280, 7)]          # (fewer than delta_max: no merge)
        for k, nm in enumerate(names):
            if k % 2:
                r.retain_remove(nm)
            else:
                r.retain_add(nm, 1_000_000 + k)
        r.retain_commit()

    def windows(b):
        out = []
        b.begin()
        while True:
            w = b.next_window()
            if w is None:
                return out
            t, o = b.window_to_host(w)
            out.append((int(w.topic_begin), int(w.topic_end), o.copy(), t.copy()))
    b = r.retain_batch(fb, fo, tier=0 if delta_max else None)
    monkeypatch.setenv("RGR_RETAIN_PACKED_READS", "0")
    ref = windows(b)
    monkeypatch.delenv("RGR_RETAIN_PACKED_READS")
    dflt = windows(b)
    b.set_retain_positions(True)
    pos = windows(b)
    vals = b.retain_vals()
    b.set_retain_positions(False)
    again = windows(b)
    assert sum(len(x[3]) for x in ref) > 100_000 and len(ref) > 3
    seen_dead = False
    for (tb0, te0, o0, t0), (tb1, te1, o1, t1), (tb2, te2, o2, t2), (_, _, _, t3) in zip(ref, dflt, pos, again):
        assert (tb0, te0) == (tb1, te1) == (tb2, te2) and np.array_equal(o0, o1) and np.array_equal(o0, o2)
        assert np.array_equal(t0, t1) and np.array_equal(t0, t3)
        assert np.array_equal(t2["topic_idx"], t0["topic_idx"]) and not t2["qos_flags"].any()
        assert np.array_equal(vals["topic_id"][t2["sub_id"]], t0["sub_id"]) and np.array_equal(vals["flags"][t2["sub_id"]], t0["qos_flags"])
        seen_dead |= bool(t0["qos_flags"].any())
    assert seen_dead == bool(delta_max)
    bp = r.batch(*pack(["a/b"]))
    with pytest.raises(capi.RgrError):
        bp.set_retain_positions(True)               # a publish batch has no value array
    bp.close(); b.close(); r.close()
