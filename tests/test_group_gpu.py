"""Multi-GPU layer of the C ABI (rgr_group_*, rgr_comm_*) on the HIP backend.

A single-GPU box cannot host two RCCL ranks, so the group is built with a repeated device ordinal: several
shards on one GPU, exchanging through device copies with the same protocol (counts round, then the
all-gatherv payload).  Everything is checked against the oracle's DefaultRouter on the UNSHARDED table.  The
RCCL transport itself is exercised with a world of one (ncclCommInitRank, ncclAllGather, the empty
send/recv group); world > 1 over xGMI needs the multi-GPU node the driver runs bench.py on."""
import numpy as np
import pytest

from oracle import oracle as orc
from rmqtt_amd import capi
from rmqtt_amd import workload as wl

pytestmark = pytest.mark.gpu


def _world(n_sub=30_000, n_pub=6_000, cfg=3):
    c = wl.CONFIGS[cfg]
    blob, offs, client, qos = wl.gen_subs(n_sub, wl.SUB_SEED + cfg, c["p_plus"], c["p_hash"], c["p_sys"])
    tb, to = wl.gen_topics(n_pub, wl.PUB_SEED + cfg, 0.01, c["p_blank"])
    o = orc.DefaultRouter()
    assert o.add_bulk(blob, offs, client, qos) == 0
    return blob, offs, qos, tb, to, o.match_flat(tb, to)


@pytest.mark.parametrize("shards", [1, 2, 3])
def test_group_matches_the_unsharded_oracle(shards):
    blob, offs, qos, tb, to, exp = _world()
    g = capi.Group([0] * shards, window_hits=40_000, chunk_topics=2048)
    assert not g.uses_rccl()            # one GPU: the shards exchange through device copies (RCCL needs distinct devices)
    assert g.subscribe_bulk(blob, offs, None, qos) == 0
    g.commit()
    # host in / host out: identical to one handle holding the whole table
    got = g.match_batch(tb, to)
    assert np.array_equal(got["status"] < 0, exp["status"] < 0)
    assert np.array_equal(got["hit_offsets"], exp["hit_offsets"])
    assert np.array_equal(got["tuples"]["sub_id"], exp["sub_ids"])
    assert np.array_equal(got["tuples"]["qos_flags"] & 0xFF, exp["qos"])
    ho = exp["hit_offsets"].astype(np.int64)
    assert np.array_equal(got["tuples"]["topic_idx"], np.repeat(np.arange(len(to) - 1, dtype=np.uint32), np.diff(ho)))
    # device-resident: per-shard counts are all-gathered and add up to the oracle's total
    gb = g.batch(tb, to)
    sh, tot = gb.run()
    assert tot == len(exp["sub_ids"]) and int(sh.sum()) == tot and len(sh) == shards
    if shards > 1:
        assert (sh > 0).all()                                    # the hash really spreads the work
    # all-gatherv of the tuples: every hit exactly once, tagged with the caller's topic index
    for consumer in range(shards):
        tot2, tup = gb.gather(consumer_shard=consumer, collect=True)
        assert tot2 == tot and len(tup) == tot
        order = np.lexsort((np.arange(len(tup)), tup["topic_idx"]))          # stable: keeps each topic's own order
        tup = tup[order]
        assert np.array_equal(tup["topic_idx"], got["tuples"]["topic_idx"])
        assert np.array_equal(tup["sub_id"], exp["sub_ids"])
    # the same exchange with RUN DESCRIPTORS as the payload (16 B per (topic, filter) run instead of 12 B per hit): every shard
    # holds a replica of every shard's subs[]; descriptor d names peer_subs(d.shard)[d.src .. d.src + d.len) for topic d.topic
    for consumer in range(shards):
        n_runs, n_hits, runs = gb.gather_runs(consumer_shard=consumer, collect=True)
        assert n_hits == tot and n_runs == len(runs) and int(runs["len"].sum()) == tot and (runs["len"] > 0).all()
        subs = [g.peer_subs(consumer, p) for p in range(shards)]
        order = np.lexsort((np.arange(len(runs)), runs["topic"]))                # stable: keeps each topic's own run order
        rs = runs[order]
        sid = np.concatenate([subs[int(r["shard"])]["sub_id"][int(r["src"]):int(r["src"]) + int(r["len"])] for r in rs]) if len(rs) else np.zeros(0, np.uint32)
        qf = np.concatenate([subs[int(r["shard"])]["qos_flags"][int(r["src"]):int(r["src"]) + int(r["len"])] for r in rs]) if len(rs) else np.zeros(0, np.uint32)
        assert np.array_equal(sid, exp["sub_ids"]) and np.array_equal(qf & 0xFF, exp["qos"])
        assert np.array_equal(np.repeat(rs["topic"], rs["len"]), got["tuples"]["topic_idx"])
    # after a table change the replicas are refreshed by the next run gather
    g.subscribe("#", 4_000_000, qos=1)
    g.commit()
    n_runs2, n_hits2, runs2 = gb.gather_runs(consumer_shard=0, collect=True)
    n_valid = int((exp["status"] >= 0).sum()) - sum(1 for t, st in zip(wl.strings(tb, to), exp["status"]) if st >= 0 and t.startswith("$"))
    assert n_hits2 == tot + n_valid                                             # '#' matches every valid non-$ topic once more
    gb.close(); g.close()


def test_group_single_subscribe_unsubscribe_routes_like_bulk():
    g = capi.Group([0, 0], window_hits=64)
    o = orc.DefaultRouter()
    filters = ["a/b/c/d", "a/+/c/#", "+/b/#", "#", "x/y/z/w/#", "x/y/z/+", "a/b/c/+", "$SYS/a/#"]
    for i, f in enumerate(filters):
        g.subscribe(f, i, qos=i % 3)
        assert o.add(f, orc.mk_id(1, f"c{i}"), orc.mk_opts(qos=i % 3), rel_id=i) == 0
    g.commit()
    tb, to = capi.pack(["a/b/c/d", "x/y/z/w", "x/y/z/w/v", "$SYS/a/b", "q", "a/b/c/e"])
    exp = o.match_flat(tb, to)
    got = g.match_batch(tb, to)
    assert np.array_equal(got["hit_offsets"], exp["hit_offsets"]) and np.array_equal(got["tuples"]["sub_id"], exp["sub_ids"])
    g.unsubscribe("a/+/c/#", 1, last_of_filter=True)          # replicated filter: removed from every shard
    g.unsubscribe("a/b/c/d", 0, last_of_filter=True)          # owned filter
    assert o.remove("a/+/c/#", orc.mk_id(1, "c1")) == 0 and o.remove("a/b/c/d", orc.mk_id(1, "c0")) == 0
    g.commit()
    exp = o.match_flat(tb, to)
    got = g.match_batch(tb, to)
    assert np.array_equal(got["hit_offsets"], exp["hit_offsets"]) and np.array_equal(got["tuples"]["sub_id"], exp["sub_ids"])
    g.close()


def test_rccl_communicator_world_of_one():
    """ncclGetUniqueId / ncclCommInitRank / ncclAllGather / the send-recv group through the library, one rank."""
    blob, offs, qos, tb, to, exp = _world(8_000, 2_000)
    r = capi.Router(device=0, window_hits=30_000)
    assert r.subscribe_bulk(blob, offs, None, qos) == 0
    r.commit()
    uid = capi.Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    c = capi.Comm(r, uid, 0, 1)
    assert c.allgather_u64(12345678901234).tolist() == [12345678901234]
    b = r.batch(tb, to)
    b.set_topic_ids(np.arange(len(to) - 1, dtype=np.uint32)[::-1].copy())       # caller-chosen ids ride along
    mine, allh, tup = c.gather_pass(b, collect=True)
    assert mine == allh == len(exp["sub_ids"]) == len(tup)
    n = len(to) - 1
    assert np.array_equal(tup["sub_id"], exp["sub_ids"])
    ho = exp["hit_offsets"].astype(np.int64)
    assert np.array_equal(tup["topic_idx"], np.repeat((n - 1 - np.arange(n)).astype(np.uint32), np.diff(ho)))
    # the run-descriptor exchange over the same RCCL communicator: replicate subs[] (all-gatherv of 8-byte entries), then descriptors
    with pytest.raises(capi.RgrError) as e:
        c.gather_runs_pass(b)                                   # refused until the subscriber entries were replicated for this epoch
    assert e.value.code == capi.RGR_ESTATE
    c.replicate_subs()
    mine_r, all_r, all_h, runs = c.gather_runs_pass(b, collect=True)
    assert mine_r == all_r == len(runs) and all_h == len(exp["sub_ids"]) and (runs["shard"] == 0).all()
    subs = c.peer_subs(0)
    sid = np.concatenate([subs["sub_id"][int(x["src"]):int(x["src"]) + int(x["len"])] for x in runs])
    assert np.array_equal(sid, exp["sub_ids"])
    assert np.array_equal(np.repeat(runs["topic"], runs["len"]), np.repeat((n - 1 - np.arange(n)).astype(np.uint32), np.diff(ho)))
    b.close(); c.close(); r.close()


@pytest.mark.parametrize("shards,key_levels", [(1, 0), (2, 0), (3, 1)])
def test_group_retained_path_matches_the_unsharded_oracle(shards, key_levels):
    """(r6) The retained-message twin over the group (rgr_group_retain_*): retained topics sharded by the hash of their key levels,
    a filter with literal key levels asks its one shard, a filter with a wildcard among them asks every shard and the answers are concatenated.
    Against the oracle's RetainTree on the UNSHARDED set (as sets per filter: the reference's own order is hash-map order), with removals and
    replacements between commits; one shard = the single-handle answer, order included."""
    from tests.parity import pack
    c = wl.CONFIGS[5]
    blob, offs = wl.gen_topics(30_000, wl.PUB_SEED + 5, 0.01, c["p_blank"], c["fixed_depth"], distinct=True)
    fb, fo, _, _ = wl.gen_subs(800, wl.SUB_SEED + 5, c["p_plus"], c["p_hash"], c["p_sys"], 0, c["fixed_depth"], force_wildcard=True)
    extra = pack(["#", "+/#", "$SYS/#", "l0x0/#", "l0x0/l1x0/#", "l0x0/l1x0/l2x0/#", "+/+/+", "l0x0/+/l2x0", "bad/#/x", "nope/#", "/#", "l0x1/l1x3/l2x1/l3x7"])
    g = capi.Group([0] * shards, window_hits=50_000)
    if key_levels:
        g.set_key_levels(key_levels)
    t = orc.RetainTree()
    assert g.retain_add_bulk(blob, offs) == t.insert_bulk(blob, offs)
    g.retain_commit()

    def check():
        for b_, o_ in ((fb, fo), extra):
            got = g.retain_match_batch(b_, o_)
            st, eo, ev, _ = t.match_batch(b_, o_)
            assert np.array_equal(st < 0, got["status"] < 0) and np.array_equal(eo, got["hit_offsets"])
            for a, e in zip(eo[:-1], eo[1:]):
                assert sorted(got["topic_ids"][int(a):int(e)].tolist()) == sorted(ev[int(a):int(e)].tolist())
        return got
    last = check()
    if shards == 1:                                  # one shard IS the single handle
        r = capi.Router(device=0, window_hits=50_000)
        r.retain_add_bulk(blob, offs)
        r.retain_commit()
        ref = r.retain_match_batch(*extra)
        assert np.array_equal(ref["topic_ids"], last["topic_ids"])
        r.close()
    names = [bytes(blob[int(offs[i]):int(offs[i + 1])]).decode() for i in range(0, 3000, 7)]
    for k, nm in enumerate(names):
        if k % 3 == 0:
            assert g.retain_remove(nm) == 0 and t.remove(nm)[0] == 1
        elif k % 3 == 1:
            assert g.retain_add(nm, 500_000 + k) == 0
            t.insert(nm, 500_000 + k)
    assert g.retain_add("brand/new/topic", 999_999) == 0
    t.insert("brand/new/topic", 999_999)
    g.retain_commit()
    check()
    got = g.retain_match_batch(*pack(["brand/#", "brand/new/+"]))
    assert got["topic_ids"].tolist() == [999_999, 999_999]
    g.close()
