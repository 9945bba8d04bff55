"""The oracle's digest and timing entry points (bench.py `parity_sample` / `cpu_baseline`, the full-size GPU tests) against the
oracle's own flat results: a digest is only a compression of `match_flat` / `RetainTree::matches`, and the reference-shaped
timed pass must count exactly the hits `DefaultRouter::matches` yields."""
import numpy as np

from oracle import oracle as orc
from rmqtt_amd import workload as wl

M64 = (1 << 64) - 1


def test_router_digest_is_a_compression_of_match_flat():
    c = wl.CONFIGS[3]
    blob, offs, client, qos = wl.gen_subs(30_000, wl.SUB_SEED + 3, c["p_plus"], c["p_hash"], c["p_sys"])
    tb, to = wl.gen_topics(2_500, wl.PUB_SEED + 3, 0.01, c["p_blank"])
    tb2, to2 = orc.pack_strings(["a/+", "sport/#/x", "", "$SYS/x"])         # wildcard-in-topic quirk, invalid, blank, meta
    o = orc.DefaultRouter()
    assert o.add_bulk(blob, offs, client, qos) == 0
    for b, f in ((tb, to), (np.frombuffer(tb2, dtype=np.uint8), f2 := to2)):
        flat = o.match_flat(b, f)
        for threads in (1, 5):
            st, d = o.match_digest(b, f, threads)
            assert np.array_equal(st, flat["status"])
            ho = flat["hit_offsets"].astype(np.int64)
            v = flat["sub_ids"].astype(object) * 4 + flat["qos"].astype(object)
            for i in range(len(f) - 1):
                x = v[ho[i]:ho[i + 1]]
                exp = (len(x), sum(x) & M64, sum((k + 1) * int(y) for k, y in enumerate(x)) & M64, sum(int(y) * int(y) for y in x) & M64)
                assert tuple(int(t) for t in d[i]) == exp, i


def test_retain_digest_is_a_compression_of_matches():
    c = wl.CONFIGS[5]
    blob, offs = wl.gen_topics(20_000, wl.PUB_SEED + 5, 0.01, c["p_blank"], c["fixed_depth"], distinct=True)
    fb, fo, _, _ = wl.gen_subs(200, wl.SUB_SEED + 5, c["p_plus"], c["p_hash"], c["p_sys"], 0, c["fixed_depth"], force_wildcard=True)
    t = orc.RetainTree()
    t.insert_bulk(blob, offs)
    st, d = t.match_digest(fb, fo, 3)
    st2, eo, ev, _ = t.match_batch(fb, fo)
    assert np.array_equal(st, st2)
    for k in range(200):
        x = [int(y) for y in ev[int(eo[k]):int(eo[k + 1])]]
        assert tuple(int(z) for z in d[k]) == (len(x), sum(x) & M64, sum(y * y for y in x) & M64)
    for dyn in (True, False):
        sec, s = t.match_timed(fb, fo, 2, dynamic=dyn)
        assert s["hits"] == len(ev) and sec > 0


def test_reference_shaped_timed_pass_counts_the_same_hits():
    c = wl.CONFIGS[3]
    blob, offs, client, qos = wl.gen_subs(20_000, wl.SUB_SEED + 3, c["p_plus"], c["p_hash"], c["p_sys"])
    tb, to = wl.gen_topics(1_500, wl.PUB_SEED + 3, 0.01, c["p_blank"])
    o = orc.DefaultRouter()
    assert o.add_bulk(blob, offs, client, qos) == 0
    flat = o.match_flat(tb, to)
    for refcounted in (True, False):
        for threads in (1, 4):
            sec, st = o.matches_timed(tb, to, threads, refcounted=refcounted)
            assert st["hits"] == len(flat["sub_ids"]) and st["invalid"] == int((flat["status"] < 0).sum())


def test_fast_router_digest_equals_the_per_hit_digest():
    """orc_router_match_digest_fast (per-filter pre-reduced digests, O(matched filters) per topic — what lets bench.py compare
    EVERY topic of the 10 M-publish batch) against the O(hits) digest, incl. after the table changed."""
    c = wl.CONFIGS[3]
    blob, offs, client, qos = wl.gen_subs(40_000, wl.SUB_SEED + 3, c["p_plus"], c["p_hash"], c["p_sys"])
    tb, to = wl.gen_topics(6_000, wl.PUB_SEED + 3, 0.01, c["p_blank"])
    tb2, to2 = orc.pack_strings(["a/+", "sport/#/x", "", "$SYS/x", "l0x0/#", "+/+", "#"])
    o = orc.DefaultRouter()
    assert o.add_bulk(blob, offs, client, qos) == 0
    for name in ("a/+", "#", "+/+", "l0x0/#"):
        for k in range(5):
            o.add(name, orc.mk_id(client_id=f"q{k}"), orc.mk_opts(qos=k % 3), rel_id=50_000 + 10 * len(name) + k)
    for b, f in ((tb, to), (np.frombuffer(tb2, dtype=np.uint8), to2)):
        st, d = o.match_digest(b, f, 3)
        for threads in (1, 4):
            st2, d2 = o.match_digest(b, f, threads, fast=True)
            assert np.array_equal(st, st2) and np.array_equal(d, d2)
    assert int(d[:, 0].sum()) > 0
    # the per-filter digests follow the table: remove a relation, add another
    assert o.remove("#", orc.mk_id(client_id="q1")) == 0
    o.add("l0x0/+", orc.mk_id(client_id="zz"), orc.mk_opts(qos=2), rel_id=77_777)
    st, d = o.match_digest(tb, to, 2)
    st2, d2 = o.match_digest(tb, to, 2, fast=True)
    assert np.array_equal(st, st2) and np.array_equal(d, d2)


def test_fast_retain_digest_equals_the_walk():
    """orc_retain_match_digest_fast ('#' through bottom-up subtree aggregates) against RetainTree::matches' own digest: generated
    trees, and a hand-made tree with the degenerate stored keys retain.rs handles ('#' and '+' as retained topic levels, '$' at the
    root and below it, blank levels)."""
    c = wl.CONFIGS[5]
    blob, offs = wl.gen_topics(30_000, wl.PUB_SEED + 5, 0.01, c["p_blank"], c["fixed_depth"], distinct=True)
    fb, fo, _, _ = wl.gen_subs(1_500, wl.SUB_SEED + 5, c["p_plus"], c["p_hash"], c["p_sys"], 0, c["fixed_depth"], force_wildcard=True)
    t = orc.RetainTree()
    t.insert_bulk(blob, offs)
    extra, eo = orc.pack_strings(["#", "+/#", "+", "+/+/#", "$SYS/#", "/#", "l0x0/#", "l0x0/+/#", "l0x1/l1x0/#", "bad/#/x"])
    for b, f in ((fb, fo), (np.frombuffer(extra, dtype=np.uint8), eo)):
        st, d = t.match_digest(b, f, 3)
        st2, d2 = t.match_digest(b, f, 3, fast=True)
        assert np.array_equal(st, st2) and np.array_equal(d, d2)
    assert int(d[0, 0]) > 20_000                     # '#' returned (nearly) the whole tree through the aggregates

    t2 = orc.RetainTree()
    names = ["a", "a/b", "a/b/c", "a/#", "a/#/x", "a/+", "a/+/y", "$SYS/x", "$SYS/x/y", "a/$m", "a/$m/n", "/", "/a", "a/", "a//b", "#", "+", "b/#",
             "b/c/#", "b/c/d", "b/c", "x/y/z/w"]
    for i, n in enumerate(names):
        t2.insert(n, i + 1)
    qs = ["#", "+", "+/#", "+/+", "+/+/#", "a/#", "a/+", "a/+/#", "a/b/#", "a/b/c/#", "b/#", "b/+/#", "b/c/#", "$SYS/#", "$SYS/+", "/#", "/+", "a//#",
          "x/#", "x/y/#", "x/y/z/#", "x/y/z/w/#", "nope/#", "a/$m/#", "+/$m/#", "a/#/x"]
    qb, qo = orc.pack_strings(qs)
    qb = np.frombuffer(qb, dtype=np.uint8)
    st, d = t2.match_digest(qb, qo, 1)
    st2, d2 = t2.match_digest(qb, qo, 2, fast=True)
    assert np.array_equal(st, st2)
    assert np.array_equal(d, d2), [(q, a.tolist(), b.tolist()) for q, a, b in zip(qs, d, d2) if not np.array_equal(a, b)]
    # the aggregates follow the tree: a removal and an insertion change node / value counts
    assert t2.remove("b/c/d")[0] == 1
    t2.insert("b/c/e/f", 99)
    st, d = t2.match_digest(qb, qo, 1)
    st2, d2 = t2.match_digest(qb, qo, 1, fast=True)
    assert np.array_equal(d, d2)
