"""Compact result formats of a device-resident batch (rgr_batch_set_format, SURVEY.md §8(b)'s SoA result):
the same hits in the same order as the 12-byte tuples, without the topic column — checked word for word
against the tuple format of the same pass, which the parity suites pin against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from rmqtt_amd import capi
from rmqtt_amd import workload as wl

pytestmark = pytest.mark.gpu

_hip = None


def d2h(ptr, nbytes):
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    out = np.empty(int(nbytes), dtype=np.uint8)
    if nbytes:
        assert _hip.hipMemcpy(out.ctypes.data, C.c_void_p(int(ptr)), int(nbytes), 2) == 0     # hipMemcpyDeviceToHost
    return out


def windows(batch, fmt):
    """-> list of (topic_begin, topic_end, offsets, sub_ids | tuples, qos | None) of one pass in format fmt."""
    batch.set_format(fmt)
    out = []
    batch.begin()
    while True:
        w = batch.next_window()
        if w is None:
            break
        offs = np.zeros(w.topic_end - w.topic_begin + 1, dtype=np.uint64)
        capi._check(capi.lib().rgr_window_to_host(batch._b, C.byref(w), None, offs.ctypes.data))      # syncs the stream
        nh = int(w.n_hits)
        if fmt == capi.RGR_FORMAT_TUPLE:
            assert not w.d_sub_ids and not w.d_qos
            a = d2h(w.d_tuples, nh * 12).view(capi.TUPLE_DTYPE) if nh else np.zeros(0, dtype=capi.TUPLE_DTYPE)
            b = None
        elif fmt == capi.RGR_FORMAT_IDS24:
            assert not w.d_tuples and not w.d_sub_ids and not w.d_qos
            raw = d2h(w.d_ids24, nh * 3).reshape(-1, 3).astype(np.uint32) if nh else np.zeros((0, 3), dtype=np.uint32)
            a = raw[:, 0] | (raw[:, 1] << 8) | (raw[:, 2] << 16)
            b = None
        else:
            assert not w.d_tuples and not w.d_ids24
            a = d2h(w.d_sub_ids, nh * 4).view(np.uint32) if nh else np.zeros(0, dtype=np.uint32)
            b = (d2h(w.d_qos, nh) if nh else np.zeros(0, dtype=np.uint8)) if fmt == capi.RGR_FORMAT_SOA else None
            assert (fmt == capi.RGR_FORMAT_SOA) == bool(w.d_qos) or nh == 0
        out.append((int(w.topic_begin), int(w.topic_end), offs, a, b))
    return out


@pytest.mark.parametrize("window_hits", [1000, 4099, 1 << 20])
def test_compact_formats_equal_tuples(window_hits):
    cfg = 3
    c = wl.CONFIGS[cfg]
    n_sub = 60_000
    blob, offs, client, qos = wl.gen_subs(n_sub, wl.SUB_SEED + cfg, c["p_plus"], c["p_hash"], c["p_sys"])
    tb, to = wl.gen_topics(5_000, wl.PUB_SEED + cfg, 0.01, c["p_blank"])
    rng = np.random.default_rng(3)
    flags = rng.integers(0, 64, size=n_sub).astype(np.uint8)            # every RGR_SUB_* bit + 2 table bits
    r = capi.Router(device=0, window_hits=window_hits)
    assert r.subscribe_bulk(blob, offs, None, qos, flags) == 0
    r.commit()
    batch = r.batch(tb, to)
    ref = windows(batch, capi.RGR_FORMAT_TUPLE)
    soa = windows(batch, capi.RGR_FORMAT_SOA)
    pk = windows(batch, capi.RGR_FORMAT_PACKED)
    i24 = windows(batch, capi.RGR_FORMAT_IDS24)
    assert len(ref) == len(soa) == len(pk) == len(i24) and len(ref) > (3 if window_hits < (1 << 20) else 0)
    total = 0
    for (tb0, te0, o0, t, _), (tb1, te1, o1, ids, q), (tb2, te2, o2, pw, _), (tb3, te3, o3, i3, _) in zip(ref, soa, pk, i24):
        assert (tb0, te0) == (tb1, te1) == (tb2, te2) == (tb3, te3) and np.array_equal(o0, o1) and np.array_equal(o0, o2) and np.array_equal(o0, o3)
        assert np.array_equal(ids, t["sub_id"])
        assert np.array_equal(i3, t["sub_id"])                      # 3-byte ids: same hits, same order
        qf = t["qos_flags"]
        assert np.array_equal(q, ((qf & 3) | (((qf >> 8) & 0x3F) << 2)).astype(np.uint8))
        assert np.array_equal(pw, t["sub_id"] | ((qf & 3) << 30))
        total += len(t)
    assert total > 100_000
    # back to tuples: the format is a per-pass choice
    again = windows(batch, capi.RGR_FORMAT_TUPLE)
    assert all(np.array_equal(a[3], b[3]) for a, b in zip(ref, again))
    batch.close(); r.close()


def test_compact_formats_retain_and_limits():
    r = capi.Router(device=0, window_hits=512)
    names = [f"k/{i}/v" for i in range(3000)] + ["k/7", "x"]
    for i, s in enumerate(names):
        assert r.retain_add(s, i) == 0
    r.retain_commit()
    fb, fo = capi.pack(["k/#", "k/+/v", "k/7/#", "#", "nope/+"])
    b = r.retain_batch(fb, fo)
    ref = windows(b, capi.RGR_FORMAT_TUPLE)
    soa = windows(b, capi.RGR_FORMAT_SOA)
    i24 = windows(b, capi.RGR_FORMAT_IDS24)
    for (_, _, o0, t, _), (_, _, o1, ids, q), (_, _, o2, i3, _) in zip(ref, soa, i24):
        assert np.array_equal(o0, o1) and np.array_equal(ids, t["sub_id"]) and not q.any()
        assert np.array_equal(o0, o2) and np.array_equal(i3, t["sub_id"])
    assert sum(len(x[3]) for x in ref) == 3001 + 3000 + 2 + 3002
    b.close()
    # delivery stage and compact formats exclude each other; PACKED needs ids below 2^30
    fid = r.filter_add("a/+")
    r.sub_add(fid, (1 << 30) + 5, 1)
    r.commit()
    tb, to = capi.pack(["a/b"])
    pb = r.batch(tb, to)
    pb.set_format(capi.RGR_FORMAT_PACKED)
    with pytest.raises(capi.RgrError) as e:
        pb.begin()
    assert e.value.code == capi.RGR_ECAPACITY
    pb.set_format(capi.RGR_FORMAT_IDS24)                             # ... and IDS24 below 2^24
    with pytest.raises(capi.RgrError) as e:
        pb.begin()
    assert e.value.code == capi.RGR_ECAPACITY
    pb.set_format(capi.RGR_FORMAT_SOA)
    assert windows(pb, capi.RGR_FORMAT_SOA)[0][3].tolist() == [(1 << 30) + 5]
    with pytest.raises(capi.RgrError) as e:
        pb.set_publish_attrs(np.zeros(1, dtype=capi.PUBLISH_ATTR_DTYPE))
    assert e.value.code == capi.RGR_ESTATE
    pb.set_format(capi.RGR_FORMAT_TUPLE)
    pb.set_publish_attrs(np.zeros(1, dtype=capi.PUBLISH_ATTR_DTYPE))
    with pytest.raises(capi.RgrError) as e:
        pb.set_format(capi.RGR_FORMAT_SOA)
    assert e.value.code == capi.RGR_ESTATE
    pb.close(); r.close()


def test_match_filters_dense_and_pinned_match_batch():
    """rgr_match_filters (dense device-side list) against the per-topic filter sequence rgr_match_batch implies,
    and rgr_match_batch's pooled pinned result across repeated calls of different sizes."""
    cfg = 3
    c = wl.CONFIGS[cfg]
    blob, offs, client, qos = wl.gen_subs(40_000, wl.SUB_SEED + cfg, c["p_plus"], c["p_hash"], c["p_sys"])
    tb, to = wl.gen_topics(9_000, wl.PUB_SEED + cfg, 0.01, c["p_blank"])
    r = capi.Router(device=0, window_hits=50_000, chunk_topics=4096, slot_cap=4)     # several chunks, overflow arena, many windows
    rej, fids = r.subscribe_bulk(blob, offs, None, qos, None, want_filter_ids=True)
    r.commit()
    sub_filter = fids                                             # sub_id i subscribes filter fids[i]
    full = r.match_batch(tb, to)
    mf = r.match_filters(tb, to)
    po = mf["pair_offsets"].astype(np.int64)
    ho = full["hit_offsets"].astype(np.int64)
    assert np.array_equal(mf["status"], full["status"])
    for t in range(len(to) - 1):
        seq = sub_filter[full["tuples"]["sub_id"][ho[t]:ho[t + 1]]]
        # the filters the hits came from, in order, runs collapsed == the matched filters that have subscribers
        runs = seq[np.r_[True, seq[1:] != seq[:-1]]] if len(seq) else seq
        assert np.array_equal(runs, mf["filter_ids"][po[t]:po[t + 1]]), t
    assert np.array_equal(full["tuples"]["topic_idx"], np.repeat(np.arange(len(to) - 1, dtype=np.uint32), np.diff(ho)))
    for n in (1, 17, 3000, 9000, 5):       # the pinned blocks are recycled across sizes
        got = r.match_batch(*wl.take(tb, to, np.arange(n)))
        assert np.array_equal(got["hit_offsets"], full["hit_offsets"][:n + 1])
        assert np.array_equal(got["tuples"]["sub_id"], full["tuples"]["sub_id"][:ho[n]])
    r.close()


def test_runs_format_is_the_hit_list_in_place():
    """RGR_FORMAT_RUNS: concatenating subs[run.src .. run.src + len) over the window's runs reproduces the tuples."""
    cfg = 3
    c = wl.CONFIGS[cfg]
    blob, offs, client, qos = wl.gen_subs(50_000, wl.SUB_SEED + cfg, c["p_plus"], c["p_hash"], c["p_sys"])
    tb, to = wl.gen_topics(4_000, wl.PUB_SEED + cfg, 0.01, c["p_blank"])
    r = capi.Router(device=0, window_hits=30_000, chunk_topics=1024)
    assert r.subscribe_bulk(blob, offs, None, qos) == 0
    r.commit()
    batch = r.batch(tb, to)
    batch.set_topic_ids(np.arange(len(to) - 1, dtype=np.uint32) + 1000)
    ref = windows(batch, capi.RGR_FORMAT_TUPLE)
    batch.set_format(capi.RGR_FORMAT_RUNS)
    batch.begin()
    k = 0
    total = 0
    while True:
        w = batch.next_window()
        if w is None:
            break
        offs_w = np.zeros(w.topic_end - w.topic_begin + 1, dtype=np.uint64)
        capi._check(capi.lib().rgr_window_to_host(batch._b, C.byref(w), None, offs_w.ctypes.data))
        assert not w.d_tuples and not w.d_sub_ids
        t = ref[k][3]; k += 1
        assert int(w.n_hits) == len(t)
        nr = int(w.n_runs)
        if not nr:
            assert len(t) == 0
            continue
        src = d2h(w.d_run_src, nr * 4).view(np.uint32)
        tpc = d2h(w.d_run_topic, nr * 4).view(np.uint32)
        off = d2h(w.d_run_off, (nr + 1) * 8).view(np.uint64).astype(np.int64) - int(w.offsets_bias)
        assert off[0] == 0 and off[-1] == len(t) and (np.diff(off) > 0).all()
        lo, hi = int(src.min()), int((src + np.diff(off)).max())
        subs = d2h(int(w.d_subs) + lo * 8, (hi - lo) * 8).view(np.dtype([("sub_id", np.uint32), ("qf", np.uint32)]))
        idx = np.concatenate([np.arange(s - lo, s - lo + n) for s, n in zip(src.astype(np.int64), np.diff(off))])
        assert np.array_equal(subs["sub_id"][idx], t["sub_id"]) and np.array_equal(subs["qf"][idx], t["qos_flags"])
        assert np.array_equal(np.repeat(tpc, np.diff(off)), t["topic_idx"]) and tpc.min() >= 1000
        total += len(t)
    assert k == len(ref) and total > 50_000
    batch.close(); r.close()


@pytest.mark.parametrize("prefetch", [True, False])
def test_chunk_prefetch_and_arena_regrow(prefetch, monkeypatch):
    """Many small chunks (the next one is prepared on a second stream while the current one expands) with a tiny
    overflow arena that has to grow and redo — in the prefetched and in the synchronous path — against one big
    chunk of the same batch."""
    cfg = 3
    c = wl.CONFIGS[cfg]
    blob, offs, client, qos = wl.gen_subs(40_000, wl.SUB_SEED + cfg, 0.2, c["p_hash"], c["p_sys"])     # wildcard-heavy: long matched-filter lists
    tb, to = wl.gen_topics(6_000, wl.PUB_SEED + cfg, 0.01, c["p_blank"])
    ref_r = capi.Router(device=0)
    assert ref_r.subscribe_bulk(blob, offs, None, qos) == 0
    ref_r.commit()
    ref = ref_r.match_batch(tb, to)
    ref_r.close()
    monkeypatch.setenv("RGR_ARENA_INIT", "8")
    if not prefetch:
        monkeypatch.setenv("RGR_NO_PREFETCH", "1")
    r = capi.Router(device=0, window_hits=20_000, chunk_topics=256, slot_cap=2)
    assert r.subscribe_bulk(blob, offs, None, qos) == 0
    r.commit()
    for _ in range(2):                     # second call: recycled workspace, arenas already grown
        got = r.match_batch(tb, to)
        assert np.array_equal(got["hit_offsets"], ref["hit_offsets"])
        assert np.array_equal(got["tuples"]["sub_id"], ref["tuples"]["sub_id"]) and np.array_equal(got["tuples"]["topic_idx"], ref["tuples"]["topic_idx"])
    b = r.batch(tb, to)
    h1, w1 = b.run()
    b.begin(); b.next_window(); b.next_window()      # abandon a pass midway (a prefetch may be in flight) ...
    h2, w2 = b.run()                                  # ... and start over
    assert h1 == h2 == len(ref["tuples"]) and w1 == w2 and r.stats()["overflow_topics"] > 0
    b.close(); r.close()


def _hot_and_cold_table(r, rng):
    """Filters whose subscriber runs are long (thousands: tiles inside ONE run), medium, short and single — and topics that match
    several of each, so that an expansion tile holds anything from one run to dozens of them."""
    sid = 0

    def sub(filt, n):
        nonlocal sid
        fid = r.filter_add(filt)
        for _ in range(n):
            r.sub_add(fid, sid, int(rng.integers(0, 3)))
            sid += 1

    sub("a/#", 9000); sub("+/#", 2371); sub("a/b/#", 5000); sub("a/+/c", 2048); sub("a/b/c", 700); sub("+/b/c", 64); sub("+/+/c", 5)
    for i in range(40):
        sub(f"a/b/c/x{i}/#", int(rng.integers(1, 4)))
    sub("a/b/+", 1); sub("#", 3); sub("a/+/+", 2); sub("+/b/+", 1)
    # Families of filters with one or two subscribers that all match ONE topic (every literal / '+' pattern of its levels below the
    # first, every '#' ending): 56 .. 62 tiny runs in a row (plus the '+/#' and '#' runs and the neighbours' ends), cut by the tile
    # boundaries at a different place in every repetition — pair lists of every length around the 64 a wave holds (62..64 is where
    # a lane looks past lane 63 for the runs after its own: profiles/r04p_*), and one family of 95 for the staged fallback
    def family(first, names, count):
        pats = []
        for mask in range(1 << len(names)):                       # exact depth: the first level literal, the others literal or '+'
            pats.append("/".join([first] + [names[j] if (mask >> j) & 1 else "+" for j in range(len(names))]))
        for k in range(0, len(names) + 1):                        # '#' endings below a prefix of k further levels
            for mask in range(1 << k):
                pats.append("/".join([first] + [names[j] if (mask >> j) & 1 else "+" for j in range(k)] + ["#"]))
        for i, f in enumerate(pats[:count]):
            sub(f, 1 + i % 2)
        return "/".join([first] + names)

    fam = [family("u%d" % j, ["b%d" % j, "c", "d", "e", "f"], 56 + j) for j in range(7)] + [family("m", ["n", "o", "p", "q", "r"], 95)]
    topics = []
    for i in range(60):
        topics += ["a/b/c", "a/b/c/x%d/y" % (i % 40), "a/q/c", "z/b/c", "a/b", "q", "a/b/c/x%d" % ((7 * i) % 40), fam[i % 8]]
        if i % 3 == 0:
            topics += [fam[(i + 3) % 8], "z/b/c", fam[(i + 5) % 8], "q", fam[(i + 6) % 8]]
    return topics


@pytest.mark.parametrize("window_hits", [50_000, 1 << 20, 8 * 2048])
def test_lane_held_expansion_equals_tile_kernel(window_hits, monkeypatch):
    """expand_compact_lp_kernel (RGR_COMPACT_LP = tiles per block; pairs held in lanes, expand_compact.inc) against the tile-per-block
    kernel and the tuples, PACKED and IDS24, on a table with long, medium, short and single-subscriber runs (tiles with one run, a
    handful, and more than a wave holds: the staged fallback).  The host twin of this test is tests/test_hipsim_expand.py."""
    rng = np.random.default_rng(11)
    r = capi.Router(device=0, window_hits=window_hits)
    topics = _hot_and_cold_table(r, rng)
    r.commit()
    tb, to = capi.pack(topics)
    batch = r.batch(tb, to)
    monkeypatch.setenv("RGR_COMPACT_LP", "0")
    ref = windows(batch, capi.RGR_FORMAT_TUPLE)
    base = {f: windows(batch, f) for f in (capi.RGR_FORMAT_PACKED, capi.RGR_FORMAT_IDS24)}
    assert sum(len(w[3]) for w in ref) > 1_000_000
    for lp in ("1", "2", "4"):
        monkeypatch.setenv("RGR_COMPACT_LP", lp)
        for f in (capi.RGR_FORMAT_PACKED, capi.RGR_FORMAT_IDS24):
            got = windows(batch, f)
            assert len(got) == len(ref)
            for (tb0, te0, o0, t, _), (tb1, te1, o1, a, _), (_, _, _, a0, _) in zip(ref, got, base[f]):
                assert (tb0, te0) == (tb1, te1) and np.array_equal(o0, o1)
                want = t["sub_id"] | ((t["qos_flags"] & 3) << 30) if f == capi.RGR_FORMAT_PACKED else t["sub_id"]
                assert np.array_equal(a0, want), (lp, f, "tile-per-block kernel")
                bad = np.flatnonzero(a != want)
                assert bad.size == 0, (lp, f, bad[:8], len(a))
    # the library's switches (measured in session r5a: X4 not adopted, fused tile records within noise; both stay selectable)
    monkeypatch.setenv("RGR_IDS24_X4", "1")          # IDS24 through 16-byte stores (expand_ids24_x4_kernel; host twin: tests/test_hipsim_expand.py)
    got = windows(batch, capi.RGR_FORMAT_IDS24)
    for (_, _, _, t, _), (_, _, _, a, _) in zip(ref, got):
        assert np.array_equal(a, t["sub_id"]), "RGR_IDS24_X4"
    monkeypatch.delenv("RGR_IDS24_X4")
    monkeypatch.setenv("RGR_TILES_FUSED", "1")       # the next window's tile records written by the tail blocks of this window's expansion
    got = windows(batch, capi.RGR_FORMAT_IDS24)
    for (_, _, _, t, _), (_, _, _, a, _) in zip(ref, got):
        assert np.array_equal(a, t["sub_id"]), "RGR_TILES_FUSED"
    monkeypatch.delenv("RGR_TILES_FUSED")
    batch.close(); r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("window_hits", [0, 40_000])
def test_walk_order_changes_nothing_per_topic(window_hits):
    """rgr_batch_set_order(RGR_ORDER_WALK) (r6): the library sorts the batch's topics by their leading tokens and walks them in that order.  Per
    topic nothing changes — the same tuples in the same order, naming the topic's BATCH index — but windows enumerate walk positions:
    d_topic_order says which batch topic the k-th offset belongs to, rgr_batch_topic_order returns the permutation (a bijection, sorted by the
    first six tokens).  Every format, caller topic ids composed with the order, and back to caller order."""
    rng = np.random.default_rng(17)
    c = wl.CONFIGS[3]
    blob, offs, _, qos = wl.gen_subs(60_000, wl.SUB_SEED + 3, c["p_plus"], c["p_hash"], c["p_sys"])
    tb, to = wl.gen_topics(9_000, wl.PUB_SEED + 3, 0.01, c["p_blank"])
    r = capi.Router(device=0, window_hits=window_hits, chunk_topics=2_000)
    assert r.subscribe_bulk(blob, offs, None, qos) == 0
    r.commit()
    b = r.batch(tb, to)
    n = len(to) - 1

    def per_topic(fmt, ids=None):
        """{batch topic (or its caller id): payload of its hits} from every window of one pass"""
        out = {}
        b.set_format(fmt)
        b.begin()
        pos = 0
        while True:
            w = b.next_window()
            if w is None:
                break
            nt = w.topic_end - w.topic_begin
            assert w.topic_begin == pos
            pos = w.topic_end
            offs_ = np.zeros(nt + 1, dtype=np.uint64)
            assert capi.lib().rgr_window_to_host(b._b, C.byref(w), None, offs_.ctypes.data) == 0
            order = capi.device_to_host(w.d_topic_order, nt * 4).view(np.uint32) if w.d_topic_order else np.arange(w.topic_begin, w.topic_end, dtype=np.uint32)
            nh = int(w.n_hits)
            if fmt == capi.RGR_FORMAT_TUPLE:
                t = capi.device_to_host(w.d_tuples, nh * 12).view(capi.TUPLE_DTYPE)
            elif fmt == capi.RGR_FORMAT_PACKED:
                t = capi.device_to_host(w.d_sub_ids, nh * 4).view(np.uint32)
            else:
                t = capi.device_to_host(w.d_ids24, nh * 3).reshape(-1, 3)
            for k in range(nt):
                a, e = int(offs_[k]), int(offs_[k + 1])
                if fmt == capi.RGR_FORMAT_TUPLE and e > a:
                    want = int(order[k]) if ids is None else int(ids[order[k]])
                    assert (t["topic_idx"][a:e] == want).all()
                out[int(order[k])] = t[a:e].copy()
        assert pos == n
        return out

    ref = {f: per_topic(f) for f in (capi.RGR_FORMAT_TUPLE, capi.RGR_FORMAT_PACKED, capi.RGR_FORMAT_IDS24)}
    assert b.topic_order() is None
    b.set_order(True)
    perm = b.topic_order()
    assert sorted(perm.tolist()) == list(range(n))
    strings = wl.strings(tb, to)
    st = b.status()
    lead = [tuple(strings[i].split("/")[:6]) for i in perm if st[i] == 0]
    groups = [lead[0]] + [x for p_, x in zip(lead, lead[1:]) if x != p_]
    assert len(groups) == len(set(groups)), "topics with the same six leading levels are contiguous in walk order"
    for f in ref:
        got = per_topic(f)
        assert got.keys() == ref[f].keys()
        for k in got:
            assert np.array_equal(got[k], ref[f][k]), (f, k)
    ids = rng.permutation(n).astype(np.uint32) + 1000
    b.set_topic_ids(ids)
    got = per_topic(capi.RGR_FORMAT_TUPLE, ids)
    for k in got:
        assert np.array_equal(got[k]["sub_id"], ref[capi.RGR_FORMAT_TUPLE][k]["sub_id"])
    b.set_topic_ids(None)
    b.set_order(False)
    back = per_topic(capi.RGR_FORMAT_TUPLE)
    for k in back:
        assert np.array_equal(back[k], ref[capi.RGR_FORMAT_TUPLE][k])
    attrs = np.zeros(n, dtype=capi.PUBLISH_ATTR_DTYPE)
    b.set_order(True)
    b.set_publish_attrs(attrs)                      # the delivery stage in walk order answers with 8-byte hits only (tests/test_deliver_parity.py) ...
    with pytest.raises(capi.RgrError):
        b.begin()                                   # ... a 12-byte tuple's topic column would name walk positions
    b.close(); r.close()
