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
