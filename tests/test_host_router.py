"""The C++ mirror of the reference's Router trait (rmqtt_amd/host/gpu_router.*), driven
through its ctypes shim, against the oracle's DefaultRouter: full SubRelationsMap semantics
(No-Local, v3 one-row-per-filter, v5 first-filter + sub-id accumulation, remove rules,
get/unique, counters).  These read like the reference's own router behaviour because the
method names and error behaviour are the trait's (rmqtt/src/router.rs:65-112)."""
import ctypes as C
import random
import re

import pytest

from oracle import brute
from oracle import oracle as orc
from rmqtt_amd import build

pytestmark = pytest.mark.gpu


class HrId(C.Structure):
    _fields_ = [("node_id", C.c_uint64), ("client_id", C.c_char_p), ("client_len", C.c_uint32), ("create_time", C.c_int64),
                ("lid", C.c_uint16)]


class HrOpts(C.Structure):
    _fields_ = [("v5", C.c_uint8), ("qos", C.c_uint8), ("no_local", C.c_uint8), ("rap", C.c_uint8), ("rh", C.c_uint8),
                ("sub_ident", C.c_uint32), ("shared_group", C.c_char_p), ("shared_group_len", C.c_uint32)]


@pytest.fixture(scope="module")
def hr():
    build.build_gpu()
    L = C.CDLL(build.build_host_router())
    vp = C.c_void_p
    L.hr_new.restype = vp; L.hr_new.argtypes = [C.c_uint64, C.c_int]
    L.hr_free.argtypes = [vp]; L.hr_free_str.argtypes = [vp]
    L.hr_new_sharded.restype = vp; L.hr_new_sharded.argtypes = [C.c_uint64, C.POINTER(C.c_int), C.c_uint32]
    L.hr_shards.argtypes = [vp]; L.hr_shards.restype = C.c_uint32
    L.hr_set_shared_policy.argtypes = [vp, C.c_int]; L.hr_set_shared_policy.restype = None
    L.hr_flag_mismatches.argtypes = [vp]; L.hr_flag_mismatches.restype = C.c_uint64
    L.hr_set_match_mode.argtypes = [vp, C.c_int]; L.hr_set_match_mode.restype = None
    L.hr_stale_expansions.argtypes = [vp]; L.hr_stale_expansions.restype = C.c_uint64
    L.hr_batcher_run.argtypes = [vp, C.POINTER(HrId), C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                 C.POINTER(C.c_uint64)]
    L.hr_batcher_run.restype = vp
    L.hr_batcher_run_async.argtypes = [vp, C.POINTER(HrId), C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    L.hr_batcher_run_async.restype = vp
    L.hr_add.argtypes = [vp, C.c_char_p, C.c_uint32, C.POINTER(HrId), C.POINTER(HrOpts)]
    L.hr_remove.argtypes = [vp, C.c_char_p, C.c_uint32, C.POINTER(HrId)]
    L.hr_matches.argtypes = [vp, C.POINTER(HrId), C.c_char_p, C.c_uint32]; L.hr_matches.restype = vp
    L.hr_get.argtypes = [vp, C.c_char_p, C.c_uint32]; L.hr_get.restype = vp
    for f in ("hr_topics", "hr_routes"):
        getattr(L, f).argtypes = [vp]; getattr(L, f).restype = C.c_int64
    L.hr_topics_tree.argtypes = [vp]; L.hr_topics_tree.restype = C.c_uint64
    L.hs_new.restype = vp; L.hs_new.argtypes = [C.c_int]
    L.hs_free.argtypes = [vp]
    L.hs_set.argtypes = [vp, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_int64, C.c_int64]
    L.hs_get.argtypes = [vp, C.c_char_p, C.c_uint32, C.c_int64]; L.hs_get.restype = vp
    L.hs_remove_expired.argtypes = [vp, C.c_int64]; L.hs_remove_expired.restype = C.c_uint64
    for f in ("hs_count", "hs_max"):
        getattr(L, f).argtypes = [vp]; getattr(L, f).restype = C.c_int64
    L.hm_new.restype = vp; L.hm_new.argtypes = [C.c_int]
    L.hm_free.argtypes = [vp]
    L.hm_set.argtypes = [vp, C.c_char_p, C.c_uint32, C.c_uint64]
    L.hm_remove.argtypes = [vp, C.c_char_p, C.c_uint32, C.c_uint64]
    L.hm_get.argtypes = [vp, C.c_char_p, C.c_uint32]; L.hm_get.restype = vp
    L.hm_values_size.argtypes = [vp]; L.hm_values_size.restype = C.c_uint64
    return L


def _id(node, client, ct=0):
    c = client.encode()
    return HrId(node, c, len(c), ct, 0), orc.mk_id(node, client, ct)


def _take(L, p):
    if not p:
        return None
    s = C.string_at(p).decode()
    L.hr_free_str(p)
    return s


def _strip_rel(s):   # oracle v3 rows carry a test-only rel_id column
    return None if s is None else re.sub(r"^(3 [^\t\n]*\t[^\t\n]*\t\d+)\t\d+", r"\1", s, flags=re.M)


MODES = [pytest.param(1, id="filters"), pytest.param(2, id="deliver")]     # GpuRouter::MatchMode: how `matches` asks the device


@pytest.mark.parametrize("mode", MODES)
def test_router_mirror_matches_oracle(hr, mode):
    g = hr.hr_new(1, 0)
    assert g
    hr.hr_set_match_mode(g, mode)
    o = orc.DefaultRouter()
    rng = random.Random(7)
    levels = ["a", "b", "c", "", "$SYS"]

    def rand_filter():
        n = rng.randint(1, 4)
        lv = [rng.choice(levels if i == 0 else levels[:4] + ["+"]) for i in range(n)]
        if rng.random() < 0.25:
            lv.append("#")
        return "/".join(lv)

    subs = []
    for i in range(600):
        f = rand_filter()
        client = f"cl{rng.randint(0, 80)}"
        node = rng.choice([1, 1, 2, 3])
        v5 = rng.random() < 0.5
        opts = dict(qos=rng.randint(0, 2), v5=v5, no_local=v5 and rng.random() < 0.5, sub_ident=rng.randint(0, 5) if v5 else 0)
        hid, oid = _id(node, client, ct=rng.randint(0, 1))
        ho = HrOpts(int(opts["v5"]), opts["qos"], int(opts["no_local"]), 0, 0, opts["sub_ident"])
        ra = hr.hr_add(g, f.encode(), len(f.encode()), C.byref(hid), C.byref(ho))
        rb = o.add(f, oid, orc.mk_opts(**opts), rel_id=i)
        assert ra == rb
        subs.append((f, node, client))
    assert hr.hr_topics(g) == o.topics() and hr.hr_routes(g) == o.routes() and hr.hr_topics_tree(g) == o.topics_tree()
    assert hr.hr_add(g, b"a/#/b", 5, C.byref(_id(1, "x")[0]), C.byref(HrOpts())) == -1      # Err like Topic::from_str

    def check_all():
        for _ in range(300):
            n = rng.randint(1, 5)
            t = "/".join(rng.choice(levels if i == 0 else levels[:4]) for i in range(n))
            if rng.random() < 0.05:
                t += "/$bad"
            f, node, client = rng.choice(subs)
            hid, oid = _id(node, client, ct=rng.randint(0, 1))
            got = _take(hr, hr.hr_matches(g, C.byref(hid), t.encode(), len(t.encode())))
            exp = _strip_rel(o.matches(oid, t))
            assert got == exp, t
            routes = _take(hr, hr.hr_get(g, t.encode(), len(t.encode())))
            if exp is None:
                assert routes is None

    check_all()
    # remove: wrong Id (create_time) is refused, right Id removes, last relation prunes the filter
    removed = 0
    for f, node, client in subs[::2]:
        for ct in (5, 0, 1):
            hid, oid = _id(node, client, ct=ct)
            ra = hr.hr_remove(g, f.encode(), len(f.encode()), C.byref(hid))
            rb = o.remove(f, oid)
            assert ra == rb, (f, client, ct)
            removed += ra == 0
    assert removed > 50
    assert hr.hr_topics(g) == o.topics() and hr.hr_routes(g) == o.routes() and hr.hr_topics_tree(g) == o.topics_tree()
    check_all()
    hr.hr_free(g)


def test_get_routes_unique(hr):   # router.rs:157-170: .unique() hides the wildcard-in-topic duplicate
    g = hr.hr_new(9, 0)
    for f in ["test/+", "test/#", "#"]:
        hid, _ = _id(9, "c")
        assert hr.hr_add(g, f.encode(), len(f), C.byref(hid), C.byref(HrOpts())) == 0
    got = _take(hr, hr.hr_get(g, b"test/+", 6))
    assert got.split("\n")[:-1] == ["#", "test/#", "test/+"]
    hr.hr_free(g)


def test_retain_storage_mirror(hr):
    """GpuRetainStorage (RetainStorage surface: set / get / expiry / counters) vs the oracle's
    RetainTree holding the same topics — DefaultRetainStorage semantics, retain.rs:229-267."""
    s = hr.hs_new(0)
    assert s
    t = orc.RetainTree()
    model = {}
    rng = random.Random(3)
    levels = ["a", "b", "c", "", "$SYS"]

    def rand_topic():
        n = rng.randint(1, 4)
        return "/".join(rng.choice(levels if i == 0 else levels[:4]) for i in range(n))

    def put(topic, payload, expiry=0, now=0):
        tb, pb = topic.encode(), payload.encode()
        assert hr.hs_set(s, tb, len(tb), pb, len(pb), expiry, now) == 0
        if payload:
            if topic not in model:
                model[topic] = len(model) + 1000
            t.insert(topic, model[topic])
        elif topic in model:
            t.remove(topic); del model[topic]

    payloads = {}
    for i in range(300):
        tp = rand_topic()
        pl = f"m{i}" if rng.random() < 0.85 else ""       # empty payload deletes the retained message
        put(tp, pl)
        if pl:
            payloads[tp] = pl
        else:
            payloads.pop(tp, None)
    assert hr.hs_count(s) == len(model) == t.values_size()
    assert hr.hs_set(s, b"a/#/b", 5, b"x", 1, 0, 0) == -1            # invalid topic name -> Err
    for f in ["#", "+", "a/#", "+/+", "a/+/c", "$SYS/#", "+/b/#", "a/b", "nope/#", "/+"]:
        got = _take(hr, hr.hs_get(s, f.encode(), len(f.encode()), 0))
        exp = "".join(f"{tp}\t{payloads[tp]}\n" for tp, _ in t.matches(f))
        assert got == exp, f
    assert hr.hs_get(s, b"a/#/b", 5, 0) is None
    # expiry: TimedValue::is_expired filters at get(); remove_expired_messages prunes
    put("exp/one", "e1", expiry=100, now=1000)
    put("exp/two", "e2", expiry=500, now=1000)
    assert _take(hr, hr.hs_get(s, b"exp/+", 5, 1050)) == "exp/one\te1\nexp/two\te2\n"
    assert _take(hr, hr.hs_get(s, b"exp/+", 5, 1200)) == "exp/two\te2\n"
    n0 = hr.hs_count(s)
    assert hr.hs_remove_expired(s, 1200) == 1 and hr.hs_count(s) == n0 - 1
    assert _take(hr, hr.hs_get(s, b"exp/#", 5, 1200)) == "exp/two\te2\n"
    assert hr.hs_max(s) >= hr.hs_count(s)
    hr.hs_free(s)


def test_message_index_mirror(hr):
    """GpuMessageIndex vs the oracle's RetainTree driven exactly as
    rmqtt-message-storage drives its `RetainTree<MsgID>`: _set appends the msg id as one more
    level (ram.rs:333-334), _get appends `+` unless the filter ends in `#` (ram.rs:381-384)."""
    m = hr.hm_new(0)
    assert m
    t = orc.RetainTree()
    rng = random.Random(11)
    levels = ["a", "b", "c", "", "$SYS"]
    stored = {}                                   # msg_id -> topic
    for msg_id in range(1, 401):
        n = rng.randint(1, 4)
        topic = "/".join(rng.choice(levels if i == 0 else levels[:4]) for i in range(n))
        tb = topic.encode()
        assert hr.hm_set(m, tb, len(tb), msg_id) == 0
        t.insert(f"{topic}/{msg_id}", msg_id)
        stored[msg_id] = topic
        if rng.random() < 0.25:                   # expiry / forwarded-to-all removal (ram.rs:226-233)
            victim = rng.choice(sorted(stored))
            vt = stored.pop(victim).encode()
            assert hr.hm_remove(m, vt, len(vt), victim) == 0
            assert hr.hm_remove(m, vt, len(vt), victim) == 1
            t.remove(f"{vt.decode()}/{victim}")
    assert hr.hm_values_size(m) == len(stored) == t.values_size()
    assert hr.hm_set(m, b"a/#/b", 5, 9999) == -1                      # Topic::from_str Err (ram.rs:333)
    for f in ["#", "+", "a", "a/b", "a/#", "+/+", "a/+/c", "$SYS/#", "$SYS", "+/b/#", "nope", "", "/", "a//b", "/#"]:
        fb = f.encode()
        q = f if (f == "#" or f.endswith("/#")) else f + "/+"
        exp = sorted(v for _, v in t.matches(q))
        got = _take(hr, hr.hm_get(m, fb, len(fb)))
        assert got is not None, f
        assert [int(x) for x in got.split(",") if x] == exp, f
    assert hr.hm_get(m, b"a/#/b", 5) is None
    hr.hm_free(m)


def _opts(qos=0, v5=False, no_local=False, sub_ident=0, group=None):
    g = group.encode() if group else None
    h = HrOpts(int(v5), qos, int(no_local), 0, 0, sub_ident, g, len(g) if g else 0)
    h._keep = g
    return h, orc.mk_opts(qos=qos, v5=v5, no_local=no_local, sub_ident=sub_ident, shared_group=group)


def _random_world(hr, g, o, rng, n_ops=700, groups=("g1", "g2")):
    """The same add / remove sequence on the mirror and on the oracle, with v3 / v5 / No Local / $share members."""
    levels = ["a", "b", "c", "", "$SYS"]
    live = []
    for step in range(n_ops):
        if rng.random() < 0.75 or not live:
            f = "/".join(rng.choice(levels + ["+"]) for _ in range(rng.randint(1, 4)))
            if rng.random() < 0.25:
                f = f.rsplit("/", 1)[0] + "/#" if "/" in f else "#"
            node, client = rng.choice([1, 1, 2, 3]), f"c{rng.randint(0, 25)}"
            v5 = rng.random() < 0.5
            grp = rng.choice(groups) if rng.random() < 0.3 else None
            ho, oo = _opts(rng.randint(0, 2), v5, v5 and rng.random() < 0.4, rng.randint(0, 9) if v5 else 0, grp)
            hi, oi = _id(node, client, rng.randint(0, 1))
            ok_o = o.add(f, oi, oo, rel_id=step) == 0
            ok_h = hr.hr_add(g, f.encode(), len(f.encode()), C.byref(hi), C.byref(ho)) == 0
            assert ok_o == ok_h, f
            if ok_o:
                live.append((f, node, client))
        else:
            f, node, client = live.pop(rng.randrange(len(live)))
            for ct in (0, 1):       # the Id must match in full (router.rs:460-467): at most one of the two create_times does
                hi, oi = _id(node, client, ct)
                assert (hr.hr_remove(g, f.encode(), len(f.encode()), C.byref(hi)) == 0) == (o.remove(f, oi) == 0)


def _topics(rng, n):
    levels = ["a", "b", "c", "", "$SYS", "zz"]
    return ["/".join(rng.choice(levels) for _ in range(rng.randint(1, 4))) for _ in range(n)] + ["a/+", "b/#", "a/#/b"]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("policy", [0, 1])
@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_shared_groups_and_sharded_router_match_oracle(hr, policy, devices, mode):
    """$share members go through SharedSubscription::choice exactly where router.rs:202-255 does (policy 0 = the
    reference's default: nobody is selected; policy 1 = an order-independent test policy installed on both sides),
    on one handle and on a router sharded over three shards."""
    devs = (C.c_int * len(devices))(*devices)
    g = hr.hr_new_sharded(1, devs, len(devices))
    assert g and hr.hr_shards(g) == len(devices)
    hr.hr_set_match_mode(g, mode)
    o = orc.DefaultRouter()
    hr.hr_set_shared_policy(g, policy)
    o.set_shared_policy(policy)
    rng = random.Random(policy * 10 + len(devices))
    _random_world(hr, g, o, rng)
    seen_group = False
    for t in _topics(rng, 250):
        for pub in (("c1", 1, 0), ("c7", 2, 1), ("nobody", 9, 0)):
            hi, oi = _id(pub[1], pub[0], pub[2])
            exp = _strip_rel(o.matches(oi, t))
            got = _take(hr, hr.hr_matches(g, C.byref(hi), t.encode(), len(t.encode())))
            assert got == exp, (t, pub)
            seen_group |= bool(exp) and "\t$g" in exp
    assert seen_group == (policy == 1)
    assert hr.hr_flag_mismatches(g) == 0
    hr.hr_free(g)


def test_batcher_many_threads_one_pass_per_batch(hr):
    """Publishes from 8 threads through the deadline micro-batcher: every caller gets exactly what the unbatched
    trait call returns, and the batcher needed far fewer device passes than publishes."""
    g = hr.hr_new(1, 0)
    o = orc.DefaultRouter()
    hr.hr_set_shared_policy(g, 1)
    o.set_shared_policy(1)
    rng = random.Random(99)
    _random_world(hr, g, o, rng, n_ops=500)
    topics = _topics(rng, 400)
    n = len(topics)
    ids_h, ids_o = zip(*[_id(rng.choice([1, 2]), f"c{rng.randint(0, 25)}", rng.randint(0, 1)) for _ in range(n)])
    arr_ids = (HrId * n)(*ids_h)
    enc = [t.encode() for t in topics]
    arr_t = (C.c_char_p * n)(*enc)
    arr_l = (C.c_uint32 * n)(*[len(e) for e in enc])
    passes = C.c_uint64(0)
    out = _take(hr, hr.hr_batcher_run(g, arr_ids, arr_t, arr_l, n, 8, 64, 2000, C.byref(passes)))
    got = out.split("\x1e")
    assert len(got) == n
    for i, t in enumerate(topics):
        exp = _strip_rel(o.matches(ids_o[i], t))
        assert got[i] == ("!ERR" if exp is None else exp), (i, t)
    assert 1 <= passes.value < n / 3
    hr.hr_free(g)


@pytest.mark.parametrize("workers", [0, 4])
def test_batcher_async_submit_pipelined_passes(hr, workers):
    """The asynchronous form of the boundary (Batcher::submit: what a tokio task awaiting `matches` is): publishes submitted without
    waiting from 4 threads, up to 3 device passes in flight, completions on pool threads (or on the drivers when there is no pool).
    Every publish gets exactly what the unbatched trait call returns."""
    g = hr.hr_new(1, 0)
    o = orc.DefaultRouter()
    hr.hr_set_shared_policy(g, 1)
    o.set_shared_policy(1)
    hr.hr_set_match_mode(g, 1)
    rng = random.Random(4242)
    _random_world(hr, g, o, rng, n_ops=600)
    topics = _topics(rng, 900)
    n = len(topics)
    ids_h, ids_o = zip(*[_id(rng.choice([1, 2]), f"c{rng.randint(0, 25)}", rng.randint(0, 1)) for _ in range(n)])
    arr_ids = (HrId * n)(*ids_h)
    enc = [t.encode() for t in topics]
    arr_t = (C.c_char_p * n)(*enc)
    arr_l = (C.c_uint32 * n)(*[len(e) for e in enc])
    passes = C.c_uint64(0)
    got = _take(hr, hr.hr_batcher_run_async(g, arr_ids, arr_t, arr_l, n, 4, 128, 500, 3, workers, C.byref(passes))).split("\x1e")
    assert len(got) == n
    for i, t in enumerate(topics):
        exp = _strip_rel(o.matches(ids_o[i], t))
        assert got[i] == ("!ERR" if exp is None else exp), (i, t)
    assert 1 <= passes.value < n / 3
    hr.hr_free(g)


@pytest.mark.gpu
def test_plain_subscribes_do_not_invalidate_passes_in_flight(hr):
    """Round-3 advisor: add() used to bump the mutation epoch, so under ordinary subscribe churn (no unsubscribe at all) nearly every
    batched publish fell into the exclusive one-publish re-match.  Only remove / restore bump it now: with a subscriber thread that
    only ADDS, no expansion may go stale — and every stable relation is still delivered."""
    import threading
    g = hr.hr_new(1, 0)
    hr.hr_set_match_mode(g, 1)
    stable = [("s/+/x", "keep1"), ("s/#", "keep2"), ("s/a/x", "keep3")]
    for f, c in stable:
        hid, _ = _id(1, c)
        assert hr.hr_add(g, f.encode(), len(f), C.byref(hid), C.byref(HrOpts())) == 0
    stop = threading.Event()

    def churn():
        k = 0
        while not stop.is_set():
            f = f"t/{k % 97}/+"
            hid, _ = _id(1, f"adder{k}")
            hr.hr_add(g, f.encode(), len(f), C.byref(hid), C.byref(HrOpts()))
            k += 1
    th = threading.Thread(target=churn)
    th.start()
    try:
        topics = ["s/a/x", "t/5/y", "u/q", "s/b/x"] * 100
        n = len(topics)
        ids_h = [_id(1, f"pub{i % 5}")[0] for i in range(n)]
        arr_ids = (HrId * n)(*ids_h)
        enc = [t.encode() for t in topics]
        arr_t = (C.c_char_p * n)(*enc)
        arr_l = (C.c_uint32 * n)(*[len(e) for e in enc])
        passes = C.c_uint64(0)
        for _ in range(3):
            out = _take(hr, hr.hr_batcher_run(g, arr_ids, arr_t, arr_l, n, 6, 64, 300, C.byref(passes))).split("\x1e")
            for t, dump in zip(topics, out):
                got = {(ln.split("\t")[0][2:], ln.split("\t")[1]) for ln in dump.split("\n") if ln.startswith("3 ")}
                for f, c in stable:
                    assert ((f, c) in got) == brute.filter_matches(f, t), (f, c, t)
    finally:
        stop.set()
        th.join()
    assert hr.hr_stale_expansions(g) == 0
    hr.hr_free(g)


def test_matches_while_the_table_changes(hr):
    """Publishes from 6 threads through the batcher while another thread subscribes and unsubscribes: a sub id freed by
    `remove` is quarantined until the next commit and every pass carries the mutation epoch it ran at, so an expansion never
    resolves a recycled id to a relation the device did not match.  Every returned row must be a relation whose filter
    matches the published topic (checked with the oracle's pairwise matcher), and relations that are never touched must
    always be delivered."""
    import threading
    import time
    g = hr.hr_new(1, 0)
    hr.hr_set_match_mode(g, 1)
    stable = [("s/+/x", "keep1"), ("s/#", "keep2"), ("s/a/x", "keep3")]
    for f, c in stable:
        hid, _ = _id(1, c)
        assert hr.hr_add(g, f.encode(), len(f), C.byref(hid), C.byref(HrOpts())) == 0
    stop = threading.Event()

    def churn():
        k = 0
        while not stop.is_set():
            f = ["t/+/y", "t/#", "u/+", "t/a/y", "+/a/#"][k % 5]
            hid, _ = _id(1, f"churn{k % 7}")
            hr.hr_add(g, f.encode(), len(f), C.byref(hid), C.byref(HrOpts()))
            if k % 3:
                hr.hr_remove(g, f.encode(), len(f), C.byref(hid))
            k += 1
            if k % 8 == 0:
                time.sleep(0.001)           # (leave some passes current: both the shared and the exclusive expansion run)
    th = threading.Thread(target=churn)
    th.start()
    try:
        topics = ["s/a/x", "t/a/y", "u/q", "s/b/x", "t/b/y"] * 60
        n = len(topics)
        ids_h = [_id(1, f"pub{i % 5}")[0] for i in range(n)]
        arr_ids = (HrId * n)(*ids_h)
        enc = [t.encode() for t in topics]
        arr_t = (C.c_char_p * n)(*enc)
        arr_l = (C.c_uint32 * n)(*[len(e) for e in enc])
        passes = C.c_uint64(0)
        for _ in range(4):
            out = _take(hr, hr.hr_batcher_run(g, arr_ids, arr_t, arr_l, n, 6, 32, 300, C.byref(passes))).split("\x1e")
            for t, dump in zip(topics, out):
                rows = [ln.split("\t") for ln in dump.split("\n") if ln.startswith("3 ")]
                got = {(r[0][2:], r[1]) for r in rows}
                for f, c in got:
                    assert brute.filter_matches(f, t), (f, t)
                for f, c in stable:
                    assert ((f, c) in got) == brute.filter_matches(f, t), (f, c, t)
    finally:
        stop.set()
        th.join()
    hr.hr_free(g)


# ---------------------------------------------------------------------------------------------- Shared::forwards (rmqtt_amd/host/gpu_shared.*)
def _shared_api(L):
    vp = C.c_void_p
    L.hr_shared_new.restype = vp; L.hr_shared_new.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32]
    L.hr_shared_free.argtypes = [vp]
    L.hr_shared_connect.argtypes = [vp, C.c_char_p, C.c_uint32, C.c_int]
    L.hr_shared_disconnect.argtypes = [vp, C.c_char_p, C.c_uint32]
    L.hr_shared_forwards.restype = vp
    L.hr_shared_forwards.argtypes = [vp, C.c_int, C.POINTER(HrId), C.c_char_p, C.c_uint64, C.c_uint8, C.c_uint8, C.c_char_p, C.c_uint32]
    L.hr_shared_counters.argtypes = [vp, C.POINTER(C.c_uint64)]


def _expected_sends(dump, node, connected, closed):
    """The oracle's forwards dump ("N <node>" sections of "<client>\\t<filter>\\t<qos'>\\t<retain'>\\t<ids>" rows) -> what the sessions of `node` are sent:
    sorted "<client>\\t<qos'>\\t<retain'>\\t<ids>" rows of the connected clients, the number of recipients reached, the sorted undelivered ones."""
    rows, errs, cur = [], [], None
    for ln in dump.splitlines():
        if ln.startswith("N "):
            cur = int(ln[2:])
            continue
        if cur != node:
            continue
        client, _filt, q, rt, ids = ln.split("\t")[:5]
        if client in closed:
            errs.append(f"! {client}\tConnection Tx is closed\n")
        elif client not in connected:
            errs.append(f"! {client}\tthe client has disconnected\n")
        else:
            rows.append(f"{client}\t{q}\t{rt}\t{ids}\n")
    return "".join(sorted(rows)) + f"= {len(rows)}\n" + "".join(sorted(errs))


def test_shared_forwards_from_delivery_words_equals_the_reference_path(hr):
    """Shared::forwards (shared.rs:735-820, 876-963) three ways for the same publishes: the oracle's _matches + collector + forwards_to dump, the C++
    restatement of the reference path (DefaultShared over Router::matches) and GpuShared — one delivery pass of the device, delivery words straight into
    the sessions' channels, no SubRelationsMap.  v3 and v5 relations, No Local publishers, Retain As Published, subscription identifiers collected
    across a v5 client's overlapping filters, relations of other nodes (not sent by a single-node Shared), disconnected clients and a closed channel;
    publishes that meet $share members and `target_clientid` publishes must take the reference's path unchanged."""
    _shared_api(hr)
    g = hr.hr_new(1, 0)
    assert g
    o = orc.DefaultRouter()
    hr.hr_set_shared_policy(g, 1); o.set_shared_policy(1)
    rng = random.Random(23)
    levels = ["a", "b", "c", ""]
    filters = ["a/#", "a/b", "a/+", "+/b", "#", "a/b/c", "+/+/c", "b/#", "a/b/#", "$SYS/#"]
    clients = [f"cl{i}" for i in range(60)]
    node_of = {c: rng.choice([1, 1, 1, 2]) for c in clients}
    for i in range(700):
        f = rng.choice(filters)
        client = rng.choice(clients)
        v5 = rng.random() < 0.55
        opts = dict(qos=rng.randint(0, 2), v5=v5, no_local=v5 and rng.random() < 0.4, rap=v5 and rng.random() < 0.5, sub_ident=rng.randint(1, 40) if v5 and rng.random() < 0.6 else 0)
        hid, oid = _id(node_of[client], client)
        ho = HrOpts(int(opts["v5"]), opts["qos"], int(opts["no_local"]), int(opts["rap"]), 0, opts["sub_ident"])
        assert hr.hr_add(g, f.encode(), len(f.encode()), C.byref(hid), C.byref(ho)) == o.add(f, oid, orc.mk_opts(**opts), rel_id=i)
    for i in range(8):                                 # $share members on one filter only
        client = f"sh{i}"
        node_of[client] = 1
        hid, oid = _id(1, client)
        sg = b"g1"
        ho = HrOpts(0, 1, 0, 0, 0, 0, sg, len(sg))
        assert hr.hr_add(g, b"b/x", 3, C.byref(hid), C.byref(ho)) == o.add("b/x", oid, orc.mk_opts(qos=1, shared_group="g1"), rel_id=900 + i)
    sh = hr.hr_shared_new(g, 1, 16, 100)
    connected = set(c for c in list(node_of) if rng.random() < 0.85)
    closed = {"cl7"}
    connected.add("cl9")
    connected -= closed
    for c in connected:
        hr.hr_shared_connect(sh, c.encode(), len(c.encode()), 0)
    hr.hr_shared_connect(sh, b"cl7", 3, 1)
    topics = ["a/b", "a/b/c", "a", "b/x", "b/y", "$SYS/x", "a//c", "x/y/z", "a/#/b", "c/b"]
    n_device = n_host = 0
    for k in range(160):
        t = rng.choice(topics)
        pub = rng.choice(clients + ["stranger"])
        hid, oid = _id(node_of.get(pub, 1), pub)
        q, rt = rng.randint(0, 2), int(rng.random() < 0.5)
        ref = _take(hr, hr.hr_shared_forwards(sh, 0, C.byref(hid), t.encode(), len(t.encode()), q, rt, None, 0))
        got = _take(hr, hr.hr_shared_forwards(sh, 1, C.byref(hid), t.encode(), len(t.encode()), q, rt, None, 0))
        assert got == ref, (t, pub, q, rt)
        dump = o.forwards(oid, t, q, bool(rt))
        if dump is None:
            assert got == "= 0\n"                        # Topic::from_str Err: logged, nobody is sent anything (shared.rs:774-777)
        elif t != "b/x":                                # (the $share publish: the chosen member's row is in the dump; counts expand the group, shared.rs:945-956)
            assert got == _expected_sends(dump, 1, connected, closed), (t, pub)
    cnt = (C.c_uint64 * 5)()
    hr.hr_shared_counters(sh, cnt)
    n_device, n_host = int(cnt[0]), int(cnt[1])
    assert n_device > 100 and n_host > 3 and int(cnt[3]) > 0          # device path taken, $share publishes handed back, other nodes' relations met
    # (r7) the publisher's owner id is cached with its From and checked against the owner index's epoch: a publisher that had no subscription when it
    # first published ("stranger": no owner id at all) subscribes with No Local, publishes, unsubscribes, publishes; other clients come and go in
    # between so that owner ids are recycled — every publish equals the reference path
    def both(pub, t, q=1, rt=0):
        hid, oid = _id(node_of.get(pub, 1), pub)
        ref = _take(hr, hr.hr_shared_forwards(sh, 0, C.byref(hid), t.encode(), len(t.encode()), q, rt, None, 0))
        got = _take(hr, hr.hr_shared_forwards(sh, 1, C.byref(hid), t.encode(), len(t.encode()), q, rt, None, 0))
        assert got == ref, (pub, t)
        dump = o.forwards(oid, t, q, bool(rt))
        assert got == _expected_sends(dump, 1, connected, closed), (pub, t)
        return got
    hr.hr_shared_connect(sh, b"stranger", 8, 0); connected.add("stranger")
    before = both("stranger", "a/b")
    assert "stranger\t" not in before
    sid, soid = _id(1, "stranger")
    nl = HrOpts(1, 2, 1, 0, 0, 0)                        # v5, qos 2, No Local
    assert hr.hr_add(g, b"a/b", 3, C.byref(sid), C.byref(nl)) == o.add("a/b", soid, orc.mk_opts(qos=2, v5=True, no_local=True), rel_id=5000)
    assert "stranger\t" not in both("stranger", "a/b")              # its own publish: dropped by No Local, on the device
    assert "stranger\t" in both("cl3", "a/b")                       # anybody else's reaches it
    for k, c in enumerate(["cl11", "cl12", "cl13"]):                # owner ids leave and come back
        for f in filters:
            hid, oid = _id(node_of[c], c)
            assert (hr.hr_remove(g, f.encode(), len(f.encode()), C.byref(hid)) == 0) == (o.remove(f, oid) == 0)
        both("stranger", "a/b"); both(c, "a/b")
        hid, oid = _id(node_of[c], c)
        ho = HrOpts(1, 1, 1, 0, 0, 0)
        assert hr.hr_add(g, b"a/#", 3, C.byref(hid), C.byref(ho)) == o.add("a/#", oid, orc.mk_opts(qos=1, v5=True, no_local=True), rel_id=6000 + k)
        both(c, "a/b"); both("stranger", "a/b/c")
    assert (hr.hr_remove(g, b"a/b", 3, C.byref(sid)) == 0) == (o.remove("a/b", soid) == 0)
    both("stranger", "a/b"); both("cl3", "a/b")
    # target_clientid: no matching at all (shared.rs:744-770)
    hid, _ = _id(1, "cl1")
    a = _take(hr, hr.hr_shared_forwards(sh, 0, C.byref(hid), b"a/b", 3, 2, 1, b"cl9", 3))
    b = _take(hr, hr.hr_shared_forwards(sh, 1, C.byref(hid), b"a/b", 3, 2, 1, b"cl9", 3))
    assert a == b and a.endswith("= 1\n")
    hr.hr_shared_free(sh)
    hr.hr_free(g)


def test_forwards_while_the_table_changes(hr):
    """(r7z) Shared::forwards from delivery words while another thread subscribes and unsubscribes.  A removal no longer makes a delivery pass stale: a hit
    whose relation is gone is skipped, and a freed sub id stays out of circulation until the device table has dropped it and no delivery pass that may
    hold it lives (gpu_router.hpp, limbo_).  So: the relations nobody touches are ALWAYS delivered to, the clients that only ever subscribe to filters
    the published topics cannot match are NEVER delivered to — although their relations take the sub ids the churning matchers free — and the device
    path is taken throughout (no publish falls back to the host path because of a removal)."""
    import threading
    import time
    _shared_api(hr)
    g = hr.hr_new(1, 0)
    stable = [("t/+/x", "keep1"), ("t/#", "keep2"), ("t/a/x", "keep3")]
    for f, c in stable:
        hid, _ = _id(1, c)
        assert hr.hr_add(g, f.encode(), len(f), C.byref(hid), C.byref(HrOpts())) == 0
    sh = hr.hr_shared_new(g, 1, 64, 100)
    everybody = [c for _, c in stable] + [f"churn{i}" for i in range(7)] + [f"never{i}" for i in range(7)]
    for c in everybody:
        hr.hr_shared_connect(sh, c.encode(), len(c.encode()), 0)
    stop = threading.Event()

    def churn():
        k = 0
        while not stop.is_set():
            f = ["t/+/x", "t/#", "t/a/+", "+/a/x", "t/a/x"][k % 5]               # matches the published topics
            hid, _ = _id(1, f"churn{k % 7}")
            hr.hr_add(g, f.encode(), len(f), C.byref(hid), C.byref(HrOpts()))
            nf = ["o/+/y", "o/#", "u/+"][k % 3]                                  # cannot match them: takes the ids the removals below free
            nid, _ = _id(1, f"never{(k * 3) % 7}")
            hr.hr_add(g, nf.encode(), len(nf), C.byref(nid), C.byref(HrOpts()))
            if k % 4:
                hr.hr_remove(g, f.encode(), len(f), C.byref(hid))
            if k % 3 == 0:
                hr.hr_remove(g, nf.encode(), len(nf), C.byref(nid))
            k += 1
            if k % 8 == 0:
                time.sleep(0.0005)
    th = threading.Thread(target=churn)
    th.start()
    try:
        pid, _ = _id(1, "publisher")
        for k in range(1500):
            t = ["t/a/x", "t/b/x"][k % 2]
            got = _take(hr, hr.hr_shared_forwards(sh, 1, C.byref(pid), t.encode(), len(t.encode()), 1, 0, None, 0))
            clients = {ln.split("\t")[0] for ln in got.splitlines() if ln and ln[0] not in "=!"}
            want = {"keep1", "keep2"} | ({"keep3"} if t == "t/a/x" else set())
            assert want <= clients, (k, t, got)
            assert not any(c.startswith("never") for c in clients), (k, t, got)
    finally:
        stop.set()
        th.join()
    cnt = (C.c_uint64 * 5)()
    hr.hr_shared_counters(sh, cnt)
    assert int(cnt[0]) == 1500 and int(cnt[1]) == 0                            # every publish finished on the device path
    hr.hr_shared_free(sh)
    hr.hr_free(g)
