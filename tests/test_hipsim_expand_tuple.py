"""The tuple expansion kernels' SOURCE (rmqtt_amd/csrc/expand_tuple.inc) on the host (tests/hipsim): the plain kernel, its delivery
variant and the delivery variant with its loads issued early (RGR_DELIVER_EARLY) over synthetic windows, against a numpy restatement
of the per-hit rules of DefaultRouter::_matches + forwards_to (router.rs:194-201, shared.rs:886-903: qos downgrade, Retain-As-Published,
No Local; which hits go through the v5 collector) — and the early variant against the product kernel word for word, including the
per-tile candidate counts, their "a whole topic lies inside this tile" flags and topic ranges.  CPU only; the device twins are
tests/test_deliver_parity.py."""
import numpy as np
import pytest

from tests.hipsim import sim

pytestmark = pytest.mark.skipif(sim.clang() is None, reason="hipsim needs clang++")

TILE = 2048
NONE = 0xFFFFFFFF
V5, NOLOCAL, SHARED, RAP = 1, 2, 4, 8


def make_window(rng, topics, pool=1 << 15, n_clients=300, v5_frac=0.3, pair_lo=2, topic_lo=1000, attrs=True, first=(1 << 33) + 9):
    """topics: list of lists of run lengths (one list per publish topic)."""
    subs = np.zeros(pool, dtype=sim.SUB_DTYPE)
    subs["sub_id"] = rng.integers(0, 1 << 30, size=pool, dtype=np.uint32)
    fl = np.where(rng.random(pool) < v5_frac, V5 | (rng.integers(0, 8, size=pool) << 1), rng.integers(0, 8, size=pool) << 1).astype(np.uint32)    # v3 subs may carry stray bits
    subs["qos_flags"] = rng.integers(0, 3, size=pool, dtype=np.uint32) | (fl << 8) | (rng.integers(0, 40, size=pool, dtype=np.uint32) << 16)
    at = np.zeros(pool, dtype=sim.ATTR_DTYPE)
    at["owner_id"] = rng.integers(0, n_clients, size=pool)
    at["client_idx"] = np.where(rng.random(pool) < 0.05, NONE, at["owner_id"])
    n_topics = len(topics)
    pub = np.zeros(topic_lo + n_topics, dtype=sim.PUB_DTYPE)
    pub["from_id"] = np.where(rng.random(len(pub)) < 0.3, NONE, rng.integers(0, n_clients, size=len(pub)))
    pub["qos_retain"] = rng.integers(0, 3, size=len(pub)) | (rng.integers(0, 2, size=len(pub)) << 2)
    lens, ptopic = [int(x) for x in rng.integers(1, 40, size=pair_lo)], [0] * pair_lo
    for t, runs in enumerate(topics):
        lens += list(runs)
        ptopic += [topic_lo + t] * len(runs)
    lens += [5, 7]
    ptopic += [topic_lo + n_topics - 1 if False else 0] * 2
    lens = np.asarray(lens, dtype=np.int64)
    src = np.array([rng.integers(0, pool - n + 1) for n in lens], dtype=np.uint32)
    off = np.concatenate([[first], first + np.cumsum(lens)]).astype(np.uint64)
    ptopic = np.asarray(ptopic, dtype=np.uint32)
    qr = (pub["qos_retain"][ptopic] & 7).astype(np.uint8)
    hi = pair_lo + sum(len(r) for r in topics)
    return dict(subs=subs, attrs=at if attrs else None, pub=pub, src=src, topic=ptopic, off=off, qr=qr, lo=pair_lo, hi=hi, topic_lo=topic_lo)


def reference(W, deliver):
    lo, hi = W["lo"], W["hi"]
    lens = np.diff(W["off"].astype(np.int64))[lo:hi]
    idx = np.concatenate([np.arange(int(W["src"][p]), int(W["src"][p]) + int(n)) for p, n in zip(range(lo, hi), lens)])
    topic = np.repeat(W["topic"][lo:hi], lens)
    s = W["subs"][idx]
    qf = s["qos_flags"].astype(np.uint32)
    if not deliver:
        return topic, s["sub_id"], qf, None
    fl = (qf >> 8) & 0xFF
    pa = W["pub"][topic]
    sq, pq = qf & 0xFF, pa["qos_retain"] & 3
    w = (qf & np.uint32(0xFFFFFF00)) | np.minimum(sq, pq)
    v5 = (fl & V5) != 0
    have = v5 & (W["attrs"] is not None)
    owner = np.where(have, W["attrs"]["owner_id"][idx] if W["attrs"] is not None else NONE, NONE).astype(np.uint32)
    client = np.where(have, W["attrs"]["client_idx"][idx] if W["attrs"] is not None else NONE, NONE).astype(np.uint32)
    frm = np.where(have & ((fl & NOLOCAL) != 0), pa["from_id"], NONE).astype(np.uint32)
    w = np.where(v5 & ((fl & RAP) != 0) & ((pa["qos_retain"] & 4) != 0), w | 4, w)
    dropped = v5 & ((fl & NOLOCAL) != 0) & (frm != NONE) & (owner == frm)
    w = np.where(dropped, w | 8, w).astype(np.uint32)
    is_cand = v5 & ((fl & SHARED) == 0) & ~dropped & (client != NONE)
    pos = np.flatnonzero(is_cand)
    return topic, s["sub_id"], w, (pos, client[pos])


def config3_like(rng, n_topics):
    out = []
    for _ in range(n_topics):
        t = [int(rng.choice([20534, 10599, 7125, 2371]))] if rng.random() < 0.7 else []
        t += [int(x) for x in rng.choice([1, 1, 2, 3, 40, 300, 863, 2371], size=rng.integers(2, 12))]
        rng.shuffle(t)
        out.append(t)
    return out


CASES = {
    "config3_like": lambda rng: make_window(rng, config3_like(rng, 5), pool=1 << 16),
    "small_topics": lambda rng: make_window(rng, [[int(x) for x in rng.integers(1, 30, size=rng.integers(1, 6))] for _ in range(150)], v5_frac=0.6),
    "one_topic_one_run": lambda rng: make_window(rng, [[3 * TILE + 5]], v5_frac=0.5),
    # topics that start and end on tile boundaries, one that ends one position past its tile (no whole topic inside: not flagged)
    "topic_fills_tile_exactly": lambda rng: make_window(rng, [[TILE - 20, 20], [TILE], [5, TILE - 5], [TILE + 1], [TILE - 1], [1, TILE], [700]], v5_frac=0.4, pair_lo=0, first=TILE * 7),
    "no_attrs": lambda rng: make_window(rng, config3_like(rng, 2), attrs=False),
    "all_v3": lambda rng: make_window(rng, config3_like(rng, 2), v5_frac=0.0),
    # more than 128 v5 hits among a wave's 256 positions: the lean variant's two-round path; all v5: every list full
    "mostly_v5": lambda rng: make_window(rng, config3_like(rng, 2), v5_frac=0.8),
    "all_v5": lambda rng: make_window(rng, [[TILE + 700, 3, 900]], v5_frac=1.0),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_tuple_expansions_source_on_host(case):
    rng = np.random.default_rng(sum(map(ord, case)))
    W = CASES[case](rng)
    args = (W["subs"], W["attrs"], W["pub"], W["src"], W["topic"], W["off"], W["qr"], W["lo"], W["hi"], W["topic_lo"])
    # plain tuples
    t0, _, _, _ = sim.expand_tuple(0, *args)
    topic, sid, qf, _ = reference(W, False)
    assert np.array_equal(t0["topic_idx"], topic) and np.array_equal(t0["sub_id"], sid) and np.array_equal(t0["qos_flags"], qf)
    # delivery: the product kernel against the restated rules
    topic, sid, w, (cpos, ccl) = reference(W, True)
    t1, l1, n1, r1 = sim.expand_tuple(1, *args)
    assert np.array_equal(t1["topic_idx"], topic) and np.array_equal(t1["sub_id"], sid)
    bad = np.flatnonzero(t1["qos_flags"] != w)
    assert bad.size == 0, (bad[:5], t1["qos_flags"][bad[:5]], w[bad[:5]])
    want_lists = [[] for _ in l1]
    for p, c in zip(cpos.tolist(), ccl.tolist()):
        want_lists[p // TILE].append((p, c))
    assert l1 == want_lists
    # ... and the early-loads variant against the product kernel, word for word
    t2, l2, n2, r2 = sim.expand_tuple(2, *args)
    assert np.array_equal(t2, t1)
    assert l2 == l1 and np.array_equal(n2, n1)
    flagged = (n1 >> 31) != 0
    assert np.array_equal(r2.reshape(-1, 2)[flagged], r1.reshape(-1, 2)[flagged])
    if case in ("small_topics", "topic_fills_tile_exactly"):
        assert flagged.any()
    # an epoch without v5 candidates wanted: tuples only
    t3, _, _, _ = sim.expand_tuple(2, *args, want_cand=False)
    assert np.array_equal(t3, t1)
    # ... and the variant that compacts every wave's v5 hits first (RGR_DELIVER_LEAN): the same words, the same candidate SETS per tile,
    # the same count words / flags / topic ranges
    for lean in (3, 4):                                    # 512 threads x 4 positions, 256 x 8
        t4, l4, n4, r4 = sim.expand_tuple(lean, *args)
        assert np.array_equal(t4, t1)
        assert l4 == l1 and np.array_equal(n4, n1)
        assert np.array_equal(r4.reshape(-1, 2)[flagged], r1.reshape(-1, 2)[flagged])
        t5, _, _, _ = sim.expand_tuple(lean, *args, want_cand=False)
        assert np.array_equal(t5, t1)
        # ... and the same kernel writing 8-byte hits {sub_id, delivery word} (RGR_FORMAT_DELIVER8): the tuples without their topic column
        h8, l8, n8, r8 = sim.expand_tuple(lean + 2, *args)
        assert np.array_equal(h8["sub_id"], t1["sub_id"]) and np.array_equal(h8["word"], t1["qos_flags"])
        assert l8 == l1 and np.array_equal(n8, n1) and np.array_equal(r8.reshape(-1, 2)[flagged], r1.reshape(-1, 2)[flagged])


# ---- property: ANY window (topics of 0 .. 12 runs of 1 .. 5 000 hits, any v5 fraction) through the lean delivery expansion equals the
# restated per-hit rules: words, candidate sets, count words
try:
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st
    HAVE_HYPOTHESIS = True
except ImportError:                                     # pragma: no cover
    HAVE_HYPOTHESIS = False

if HAVE_HYPOTHESIS:
    run_len = st.one_of(st.integers(1, 4), st.integers(1, 70), st.integers(1, 5000), st.sampled_from([2047, 2048, 2049]))

    @settings(max_examples=25, deadline=None, suppress_health_check=list(HealthCheck))
    @given(topics=st.lists(st.lists(run_len, min_size=0, max_size=12), min_size=1, max_size=25), v5=st.sampled_from([0.0, 0.05, 0.3, 0.7, 1.0]),
           pair_lo=st.integers(0, 3), first=st.integers(0, 1 << 40), seed=st.integers(0, 1 << 16), geometry=st.sampled_from([3, 4]), attrs=st.booleans())
    def test_lean_delivery_expansion_any_window(topics, v5, pair_lo, first, seed, geometry, attrs):
        total, kept = 0, []
        for t in topics:                                 # at most ~12 tiles per example
            t = list(t)
            while t and total + sum(t) > 12 * TILE:
                t.pop()
            kept.append(t); total += sum(t)
        if total == 0:
            kept[0] = [3]
        rng = np.random.default_rng(seed)
        W = make_window(rng, kept, pool=1 << 14, v5_frac=v5, pair_lo=pair_lo, first=first, attrs=attrs)
        args = (W["subs"], W["attrs"], W["pub"], W["src"], W["topic"], W["off"], W["qr"], W["lo"], W["hi"], W["topic_lo"])
        topic, sid, w, (cpos, ccl) = reference(W, True)
        t4, l4, n4, _ = sim.expand_tuple(geometry, *args)
        assert np.array_equal(t4["topic_idx"], topic) and np.array_equal(t4["sub_id"], sid)
        bad = np.flatnonzero(t4["qos_flags"] != w)
        assert bad.size == 0, (bad[:5], t4["qos_flags"][bad[:5]], w[bad[:5]])
        want_lists = [[] for _ in l4]
        for p, c in zip(cpos.tolist(), ccl.tolist()):
            want_lists[p // TILE].append((p, c))
        assert l4 == want_lists
        assert np.array_equal(n4 & 0x7FFFFFFF, np.array([len(x) for x in want_lists], dtype=np.uint32))

