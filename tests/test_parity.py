"""Parity of the matching pipeline against the CPU oracle.

Every case runs on two backends:
  emu  — host emulation of the kernels' own per-lane code + the product table compiler
         (CPU, part of the `-m "not gpu"` suite; validates host logic and index math)
  hip  — the real thing: HIP kernels on cuda:0 through the C ABI (`-m gpu`)
Bit-exact bar (integer work): identical per-topic (sub_id, qos) sequences (App. A.5).
"""
import random

import numpy as np
import pytest

from oracle import brute
from oracle import oracle as orc
from tests import parity
from tests.parity import Pair, pack

BACKENDS = ["emu", pytest.param("hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def kind(request):
    return request.param


def test_reference_trie_vectors(kind):   # trie.rs:445-477 replayed through the router surface
    p = Pair(kind)
    vec = [("/iot/b/x", 1), ("/iot/b/x", 2), ("/iot/b/y", 3), ("/iot/cc/dd", 4), ("/ddl/22/#", 5), ("/ddl/+/+", 6),
           ("/ddl/+/1", 7), ("/ddl/#", 8), ("/xyz/yy/zz", 9), ("/xyz", 10)]
    for f, v in vec:
        p.add(f, f"c{v}", v, qos=v % 3)
    p.commit()
    topics = ["/iot/b/x", "/iot/b/y", "/iot/cc/dd", "/xyz/yy/zz", "/ddl/22/1/2", "/ddl/22/1", "/ddl/22/", "/ddl/22", "/nope", ""]
    exp, got = p.check(*pack(topics))
    per = [sorted(got["tuples"]["sub_id"][int(a):int(b)].tolist()) for a, b in zip(got["hit_offsets"][:-1], got["hit_offsets"][1:])]
    assert per == [[1, 2], [3], [4], [9], [5, 8], [5, 6, 7, 8], [5, 6, 8], [5, 8], [], []]
    p.remove("/iot/b/x", "c2", 2)
    p.remove("/xyz/yy/zz", "c9", 9)
    p.commit()
    exp, got = p.check(*pack(topics))
    assert got["hit_offsets"][4] - got["hit_offsets"][3] == 0     # /xyz/yy/zz no longer matches


def test_edge_cases(kind):
    p = Pair(kind)
    filters = ["#", "+", "+/+", "/+", "+/#", "a/#", "a/+", "a/b", "a/b/#", "a//b", "/", "", "$SYS/#", "$SYS/+/x", "+/monitor/Clients",
               "a/+/c", "a/b/c", "a/+/+", "x/y/z/#", "x/y/z/", "x/y/z/+", "test/+", "test/#", "Case/Topic"]
    for i, f in enumerate(filters):
        assert p.add(f, f"c{i}", i, qos=i % 3)
    assert not p.add("a/#/b", "bad", 999)
    assert not p.add("a+", "bad", 999)
    assert not p.add("a/$b", "bad", 999)
    p.add("a/b", "second", 100, qos=2, v5=True, no_local=True)      # two relations on one filter
    p.commit()
    topics = ["a", "a/", "a/b", "a/b/c", "a/b/c/d", "a//b", "/", "", "/a", "$SYS", "$SYS/", "$SYS/a/x", "$SYS/monitor/Clients",
              "b/monitor/Clients", "x/y/z/", "x/y/z/2", "x/y/z", "test/+", "test/#", "+", "#", "a/+", "case/topic", "Case/Topic",
              "a/$b", "a/#/b", "a+", "unknown/levels/everywhere", "a/unknown", "/".join(["a"] * 40), "a/b/" + "/".join(["q"] * 100)]
    exp, got = p.check(*pack(topics))
    assert list(got["status"][-7:-4]) == [-2, -2, -2]                # the three invalid topics
    # independent cross-check with the brute-force matcher (wildcard-free, valid topics only)
    for ti, t in enumerate(topics):
        if not brute.valid(t) or "+" in t or "#" in t:
            continue
        subs = got["tuples"]["sub_id"][int(got["hit_offsets"][ti]):int(got["hit_offsets"][ti + 1])]
        hit_filters = sorted(filters[s] if s < 100 else "a/b" for s in subs)
        want = sorted([f for f in filters if brute.filter_matches(f, t)] + (["a/b"] if brute.filter_matches("a/b", t) else []))
        assert hit_filters == want, t


def test_empty_inputs(kind):
    p = Pair(kind)
    p.commit()
    p.check(*pack(["a/b", "", "#"]))                 # empty table
    p.check(np.zeros(0, np.uint8), np.zeros(1, np.uint64))   # empty batch
    p.add("a/b", "c", 0)
    p.commit()
    p.check(np.zeros(0, np.uint8), np.zeros(1, np.uint64))


ALPHA = ["a", "b", "c", "d", "", "$s", "ee"]


def _rand_topic(rng, wild, maxd=6):
    n = rng.randint(1, maxd)
    lv = []
    for i in range(n):
        x = rng.random()
        if wild and x < 0.18:
            lv.append("+")
        elif wild and x < 0.3 and i == n - 1:
            lv.append("#")
        else:
            lv.append(rng.choice(ALPHA if i == 0 else [a for a in ALPHA if a != "$s"]))
    return "/".join(lv)


@pytest.mark.parametrize("opts", [
    dict(),                                                           # defaults
    dict(slot_cap=1),                                                 # nearly every topic overflows the slots
    dict(slot_cap=2, chunk_topics=300, window_hits=64),               # many chunks, tiny windows
    dict(slot_cap=3, chunk_topics=257, window_hits=1, lds_window=0, tile=4),    # one-topic windows, no LDS window
    dict(chunk_topics=1000, window_hits=5000, lds_window=64, tile=16),
])
def test_random_tables_all_paths(kind, opts):
    rng = random.Random(99)
    p = Pair(kind, **opts)
    rels, sub = [], 0
    for _ in range(900):
        f = _rand_topic(rng, True)
        for _ in range(rng.choice([1, 1, 1, 2, 5])):
            p.add(f, f"c{sub}", sub, qos=sub % 3, v5=bool(sub % 5 == 0), no_local=bool(sub % 10 == 0))
            rels.append((f, f"c{sub}", sub))
            sub += 1
    p.commit()
    topics = [_rand_topic(rng, False, 7) for _ in range(1500)] + [_rand_topic(rng, True, 4) for _ in range(100)]   # + A.4 quirk
    p.check(*pack(topics), what=str(opts))
    # churn: drop a third of the relations (pruning emptied filters), add new ones, re-check
    for f, c, sid in rels[::3]:
        p.remove(f, c, sid)
    for i in range(200):
        p.add(_rand_topic(rng, True), f"n{i}", 100000 + i, qos=i % 3)
    p.commit()
    p.check(*pack(topics), what=str(opts) + " after churn")


def test_incremental_commits(kind):
    """Many small add/remove + commit cycles (the broker's SUBSCRIBE/UNSUBSCRIBE traffic): every
    epoch must equal the oracle.  On the HIP backend this exercises the delta path of rgr_commit
    (dirty edge slots / filter runs patched into the ping-pong images, append-only subscriber pool,
    pool rebuild) and an in-flight pass pinning an old epoch across commits."""
    rng = random.Random(2024)
    p = Pair(kind)
    live, sub = [], 0
    for _ in range(400):
        f = _rand_topic(rng, True)
        p.add(f, f"c{sub}", sub, qos=sub % 3); live.append((f, f"c{sub}", sub)); sub += 1
    p.commit()
    topics = [_rand_topic(rng, False, 7) for _ in range(400)]
    tb, to = pack(topics)
    p.check(tb, to)
    pinned = None
    if kind == "hip":
        pinned = p.backend.batch(tb, to)
        pinned.begin()                              # binds the current epoch and keeps it alive
        pinned_expect = p.oracle.match_flat(tb, to)
    for it in range(60):
        for _ in range(rng.randint(1, 12)):
            if live and rng.random() < 0.45:
                f, c, sid = live.pop(rng.randrange(len(live)))
                p.remove(f, c, sid)
            else:
                f = _rand_topic(rng, True) if rng.random() < 0.6 else rng.choice(live)[0] if live else "a/b"
                c = f"n{sub}"
                p.add(f, c, sub, qos=sub % 3); live.append((f, c, sub)); sub += 1
        p.commit()
        p.check(tb, to, what=f"commit {it}")
    if kind == "hip":
        st = p.backend.stats()
        assert st["commits_delta"] >= 40, st        # most epochs were published by patching
        parts = []
        while True:
            w = pinned.next_window()
            if w is None:
                break
            parts.append(pinned.window_to_host(w)[0])
        got = np.concatenate(parts) if parts else np.zeros(0, dtype=capi_tuple_dtype())
        assert np.array_equal(got["sub_id"], pinned_expect["sub_ids"])     # the old epoch was never touched
        pinned.close()


def capi_tuple_dtype():
    from rmqtt_amd import capi
    return capi.TUPLE_DTYPE


def test_long_pair_lists(kind):
    """A topic matched by hundreds of filters (every '+' pattern over 8 levels = 256 filters, plus
    their '#' truncations): exercises slot overflow and the block-cooperative count/compact."""
    import itertools
    p = Pair(kind)
    sub = 0
    for mask in itertools.product([0, 1], repeat=8):
        f = "/".join("+" if m else "a" for m in mask)
        for k in range(1 + (sub % 3)):
            p.add(f, f"c{sub}", sub, qos=sub % 3)
            sub += 1
        if sum(mask) <= 2:
            for d in range(1, 8):
                p.add("/".join(f.split("/")[:d]) + "/#", f"h{sub}", sub)
                sub += 1
    p.commit()
    topics = ["a/a/a/a/a/a/a/a", "a/a/a/a/a/a/a/b", "b/a/a/a/a/a/a/a", "a/a/a/a", "a/a/a/a/a/a/a/a/a", "$a/a/a/a/a/a/a/a"] * 3
    exp, got = p.check(*pack(topics))
    assert exp["hit_offsets"][1] > 600          # the first topic really has a long list


def test_lds_window_boundary_small_batches(kind):
    """The walk keeps a block's tokens and DFS stacks in an LDS window (10 words per topic) and takes them from HBM
    for a topic that does not lie wholly inside it; the choice is per topic, so one wave runs both instances
    (kernels.hip walk_kernel).  Small batches of every size around the block / wave sizes, with long topics mixed in
    at varying positions so that the window ends inside, before and after them (r2: an explicit ds_read variant
    miscompared on small batches — this is its regression test)."""
    import random
    rng = random.Random(77)
    kw = dict(lds_window=24) if kind == "emu" else {}
    p = Pair(kind, **kw)
    fl = ["a/#", "a/+/c", "a/b/c", "+/b/#", "#", "a/b/c/d/e/f/g/h/i/j/k/l/m/n/o/p", "a/+/+/+/+/+/+/+/+/+/+/+/+/+/#", "+", "a", "a/b/+/d/+/f/#",
          "/".join(["+"] * 30), "/".join(["a"] * 30), "/".join(["a"] * 29) + "/#", "b/+", "b/c/#"]
    for i, f in enumerate(fl):
        assert p.add(f, f"c{i}", i, qos=i % 3)
    p.commit()
    short = ["a/b/c", "a/x/c", "b/c", "b/c/d", "a", "x", "a/b", "a/b/c/d", "$SYS/x", ""]
    long_ = ["/".join(["a"] * 30), "/".join(["a"] * 31), "a/b/c/d/e/f/g/h/i/j/k/l/m/n/o/p", "/".join(["a"] * 200), "a/" + "/".join(["q"] * 400),
             "/".join(["a", "b"] * 15), "/".join(["a"] * 3000)]
    for n in [1, 2, 3, 7, 31, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 513]:
        for frac in (0.0, 0.02, 0.3):
            topics = [rng.choice(long_) if rng.random() < frac else rng.choice(short) for _ in range(n)]
            if frac and n > 2:
                topics[rng.randrange(n)] = long_[3]
                topics[-1] = long_[1]
            p.check(*pack(topics), what=f"n={n} long fraction {frac}")


def test_matched_filter_order(kind):   # TopicTree::matches order incl. duplicates (App. A.2 / A.4)
    p = Pair(kind)
    fl = ["a/b", "a/#", "a/+", "+/b", "#", "+/#", "a/b/#", "+/+", "a/+/#", "test/+", "test/#"]
    for i, f in enumerate(fl):
        p.add(f, f"c{i}", i)
    p.commit()
    got = p.backend.match_filters(*pack(["a/b", "test/+", "test/#", "$x/b"]))
    by_id = {p.backend.filter_find(f): f for f in fl}
    seqs = [[by_id[i] for i in got["filter_ids"][int(a):int(b)]] for a, b in zip(got["pair_offsets"][:-1], got["pair_offsets"][1:])]
    assert seqs[0] == ["#", "+/#", "+/+", "+/b", "a/#", "a/+", "a/+/#", "a/b", "a/b/#"]
    # a wildcard level inside a PUBLISH topic is looked up literally and reaches the '+' / '#'
    # child a second time (trie.rs:358-370): duplicates are part of the reference behaviour
    assert seqs[1] == ["#", "+/#", "+/+", "+/+", "test/#", "test/+", "test/+"]
    assert seqs[2] == ["#", "+/#", "+/+", "+/#", "test/#", "test/+", "test/#"]
    # ... and must equal the oracle's own iteration order
    t = orc.TopicTree()
    for i, f in enumerate(fl):
        t.insert(f, i)
    for k, tp in enumerate(["a/b", "test/+", "test/#"]):
        assert seqs[k] == [f for f, _ in t.matches(tp)]
    assert seqs[3] == []                              # '$'-topic is isolated from root wildcards


@pytest.mark.parametrize("cfg,n_sub,n_pub", [(1, 10_000, 20_000), (2, 30_000, 20_000), (3, 30_000, 6_000)])
def test_seeded_workloads_small(kind, cfg, n_sub, n_pub):
    """BASELINE.json configs at sizes the oracle finishes in seconds."""
    (blob, offs, client, qos), (tb, to) = parity.workload(cfg, n_sub, n_pub)
    p = Pair(kind, chunk_topics=4096 if cfg == 3 else 0, window_hits=200_000 if cfg == 3 else 0)
    p.add_bulk(blob, offs, client, qos)
    exp, got = p.check(tb, to, what=f"config {cfg}")
    if cfg == 1:
        assert exp["stats"]["invalid"] == 0
    else:
        assert exp["stats"]["hits"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("shift", ["0", "3", "6"])
def test_small_batches_under_every_lane_mapping(shift, monkeypatch):
    """(r7) A chunk that cannot fill the chip spreads its walks — and a small batch its tokeniser kernels — over more waves: only every 2^shift-th lane
    takes an item (kernels.hip small_batch_lane_shift; the rule picks the shift by size).  Every test of a few thousand topics runs under the rule's
    choice; this one pins the other mappings — all 64 lanes (what a large chunk uses), every 8th, one item per wave — on batches of 1, 63, 257 and
    20 000 topics (parser edge cases included) against the oracle."""
    monkeypatch.setenv("RGR_WALK_LANE_SHIFT", shift)
    (blob, offs, client, qos), (tb, to) = parity.workload(2, 30_000, 20_000)
    p = Pair("hip")
    p.add_bulk(blob, offs, client, qos)
    for n in (1, 63, 257, 20_000):
        p.check(tb[:int(to[n])], to[:n + 1], what=f"lane shift {shift}, {n} topics")
    odd = ["", "/", "a//b", "+/x", "a/#/b", "$SYS/x", "a/b/c/d/e/f/g/h/i/j/k/l/m/n/o/p/q/r/s/t", "x" * 300] * 9
    ob, oo = pack(odd)
    p.check(ob, oo, what=f"lane shift {shift}, parser edge cases")


def test_several_tables_in_one_pass(kind):
    """the reference matches every publish against more `TopicTree`s than the
    router's — e.g. the egress bridges' `TopicTree<(BridgeName, EntryIndex)>`
    (rmqtt-bridge-egress-mqtt/src/bridge.rs:103,202).  Tagged with a table id in the flag bits their
    entries share the handle: one pass, results demultiplexed per table, each equal to that
    table's own oracle TopicTree."""
    from rmqtt_amd import capi
    rng = random.Random(8)
    b = parity.make_backend(kind)
    trees = [orc.TopicTree(), orc.TopicTree(), orc.TopicTree()]
    pool = ["a/b", "a/+", "a/#", "#", "+/b", "a/b/c", "+/+", "$SYS/#", "x/y", "a/b/#", "+/#", "x/+"]
    next_id = 0
    for table, tree in enumerate(trees):
        for _ in range(40 if table == 0 else 12):
            f = rng.choice(pool)
            fid = b.filter_add(f)
            b.sub_add(fid, next_id, rng.randrange(3), table << capi.RGR_SUB_TABLE_SHIFT)
            tree.insert(f, next_id)
            next_id += 1
    b.commit()
    topics = ["a/b", "a/b/c", "x/y", "a", "$SYS/q", "q", "a/+", "x/y/z"]
    blob, offs = pack(topics)
    got = b.match_batch(blob, offs)
    for i, t in enumerate(topics):
        tup = got["tuples"][int(got["hit_offsets"][i]):int(got["hit_offsets"][i + 1])]
        tab = (tup["qos_flags"] >> 8 & capi.RGR_SUB_TABLE_MASK) >> capi.RGR_SUB_TABLE_SHIFT
        for table, tree in enumerate(trees):
            exp = [v for _, vals in tree.matches(t) for v in sorted(vals)]
            assert tup["sub_id"][tab == table].tolist() == exp, (t, table)
