#!/usr/bin/env python3
"""bench.py — publish-topic matches/sec of the MI355X topic matcher (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, or by itself: without
                                                          WORLD_SIZE in the environment it spawns its own N ranks)

A "step" is one pass of the hot path (trie walk + subscriber expansion into
(topic_idx, sub_id, qos) tuples) over one batch of publish topics whose '/'-tokenised
form is already resident in HBM.  Workload at N=1: BASELINE.json configs[2] — 10 M
subscriptions with mixed '+'/'#' (seeded generator of SURVEY.md §8(d)), 10 M publishes,
1x MI355X.  N>1: the same table hash-sharded by the first three topic levels, publishes
routed to their owner rank (strong scaling: total work fixed); ranks exchange per-rank
hit counts (all-gather over RCCL), tuples stay on the owning GPU unless --gather tuples; --gather runs adds
the all-gatherv of 16-byte run descriptors (every rank then knows every rank's hits: DESIGN §7).

One JSON line on rank 0:
  value         whole-job publish-topic matches/s, tuples left in HBM
  roofline      dominant kernel: `frac` = HBM bytes the PMC counters saw (FETCH_SIZE + WRITE_SIZE, two
                separate rocprofv3 passes of this very script in --pmc-child mode) / HIP-event time /
                8 TB/s; `alg_frac` = SURVEY §8(d)'s algorithmic bytes over the same time (its 8 B/hit
                read term is served by L2/MALL for hot filters, so alg_frac can exceed frac — and 1)
  parity_sample per-topic digests (count, sum, order-dependent sum, sum of squares of sub_id*4+qos) of EVERY window
                of a full pass of the timed batch in each result format (tuple, soa, packed, runs): compact formats
                vs the tuple format on all topics, tuple format vs the oracle on the FULL table for EVERY topic of the
                batch (the oracle composes a topic's digest from per-filter pre-reduced digests, O(matched filters) per
                topic, and cross-checks that against its per-hit digest on a sample); N > 1: every rank digests its own
                topics, rank 0 compares the assembled batch with the oracle on the UNSHARDED table; the line FAILS
                (exit 1) when anything differs
  cpu_baseline  the oracle's DefaultRouter::_matches-shaped pass ("port") on this host's cores over a
                bounded prefix of the same batch
  pcie_inclusive_matches_per_s   the same path with every window copied to pinned host memory
  secondary     configs[1] and configs[4] (retained path) measured the same way in the same run
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
T0 = time.time()


def log(msg, rank=0):
    if rank == 0:
        print(f"[bench +{time.time() - T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------- workload
def gen_workload(cfg, scale, rank=0):
    """Seeded inputs of BASELINE config `cfg`; cached under $TMPDIR so the --pmc-child
    processes of the same run do not regenerate them."""
    from rmqtt_amd import workload as wl
    c = wl.CONFIGS[cfg]
    n_sub = max(1, int(c["n_sub"] * scale))
    n_pub = max(1, int(c["n_pub"] * scale))
    cache = os.path.join(tempfile.gettempdir(), f"rgr_bench_wl_cfg{cfg}_{n_sub}_{n_pub}_u{os.getuid()}.npz")
    if os.path.exists(cache):
        try:
            z = np.load(cache)
            W = {k: z[k] for k in z.files}
            W.update(cfg=cfg, n_sub=n_sub, n_pub=n_pub, retain=cfg == 5, c=c)
            if "client" not in W:
                W["client"] = W["qos"] = None
            return W
        except Exception:
            pass
    retain = cfg == 5
    if retain:
        # config 5: table = n_sub DISTINCT retained topics (publish generator), queries = n_pub wildcard SUBSCRIBE filters
        log(f"config 5: generating {n_sub} retained topics / {n_pub} wildcard filters", rank)
        blob, offs = wl.gen_topics(n_sub, wl.PUB_SEED + cfg, 0.01, c["p_blank"], c["fixed_depth"], distinct=True)
        client = qos = None
        tb, to, _, _ = wl.gen_subs(n_pub, wl.SUB_SEED + cfg, c["p_plus"], c["p_hash"], c["p_sys"], 0, c["fixed_depth"], force_wildcard=True)
    else:
        log(f"config {cfg}: generating {n_sub} subscriptions / {n_pub} publish topics", rank)
        blob, offs, client, qos = wl.gen_subs(n_sub, wl.SUB_SEED + cfg, c["p_plus"], c["p_hash"], c["p_sys"], 0, c["fixed_depth"])
        tb, to = wl.gen_topics(n_pub, wl.PUB_SEED + cfg, 0.01 if cfg != 1 else 0.0, c["p_blank"], c["fixed_depth"])
    W = dict(blob=blob, offs=offs, tb=tb, to=to)
    if not retain:
        W.update(client=client, qos=qos)
    if rank == 0 and n_sub >= 100_000:
        try:
            tmp = cache + f".{os.getpid()}.tmp.npz"
            np.savez(tmp, **W)
            os.replace(tmp, cache)
        except OSError:
            pass
    W.update(cfg=cfg, n_sub=n_sub, n_pub=n_pub, retain=retain, c=c)
    if retain:
        W["client"] = W["qos"] = None
    return W


def prefix(W, n):
    """First n query strings of the batch."""
    from rmqtt_amd import shard
    return shard.take(W["tb"], W["to"], np.arange(n))


_PERMS = {}


def sample_index(W, n, seed=20260922):
    """Indices (sorted = batch order) of the n-query sample: the first n entries of ONE seeded permutation of the batch, so that samples of
    different sizes are NESTED — the single-thread leg of the CPU baseline runs a subset of exactly the queries the all-threads leg ran."""
    key = (W["cfg"], W["n_pub"], seed)
    if key not in _PERMS:
        _PERMS.clear()
        _PERMS[key] = np.random.default_rng(seed).permutation(W["n_pub"])
    return np.sort(_PERMS[key][:min(n, W["n_pub"])])


def sample(W, n, seed=20260922):
    """n query strings of the batch drawn uniformly at random (seeded, without replacement, in batch order) — the CPU
    baseline's sample: a prefix would be as fair for typical topics, a draw over the whole batch is not open to the question."""
    from rmqtt_amd import shard
    if n >= W["n_pub"]:
        return W["tb"], W["to"]
    return shard.take(W["tb"], W["to"], sample_index(W, n, seed))


def cpu_baseline_leg(o, W, args, cores, primary, hits_per_topic, unit):
    """The CPU baseline of one record: the oracle's reference-shaped pass on this host's cores over a bounded, seeded
    sample of the SAME batch against the SAME (full, unsharded) table — on all threads, and on ONE thread over a nested subset that
    is also re-run on all threads, so the two figures of `single_thread` are comparable query for query."""
    retain, n_pub, n_sub = W["retain"], W["n_pub"], W["n_sub"]
    # bounded sample: ~15 s of CPU work at the oracle's measured rates (primary), ~5 s (secondary)
    budget_hits = (3.5e7 if retain else 1.2e9) * cores / 256 * (1.0 if primary else 0.3)
    n_s = args.cpu_sample if args.cpu_sample > 0 else int(min(n_pub, max(200 if retain else 2000, budget_hits / hits_per_topic)))
    sb, so = sample(W, n_s)

    def run(b_, o_, threads, **kw):
        return o.match_timed(b_, o_, threads, dynamic=True) if retain else o.matches_timed(b_, o_, threads, **kw)
    sec, ost = run(sb, so, cores)
    if retain:
        what = "RetainTree::matches (retain.rs:450-526), filters handed out one at a time"
    else:
        what = ("DefaultRouter::_matches-shaped (router.rs:174-265: parse, trie walk, relations lookup, per-hit ref-counted clones into the "
                "collector; no canonicalising sort), chunks of 16 topics from an atomic cursor")
    cpu = {"value": round(n_s / sec, 1), "unit": unit, "cores": cores, "kind": "port", "what": what,
           "sample": f"{n_s} queries drawn at random (seeded) from the same batch, against the full {n_sub}-entry table, {ost['hits']} hits, {sec:.2f}s wall",
           "hits_per_s": round(ost["hits"] / sec, 1)}
    if cores > 1:
        # SURVEY 8(d) also asks for the single-thread figure.  Sized adaptively to >= 1 s of single-thread work (the round-5 record timed 20
        # retained-path queries in 0.02 s: noise), on a NESTED subset of the sample above, and the same subset once more on all threads
        n1 = max(20, min(n_s, int(n_s / cores * 2.0)))
        sec1 = ost1 = None
        for _ in range(4):
            s1b, s1o = sample(W, n1)
            sec1, ost1 = run(s1b, s1o, 1)
            if sec1 >= 1.0 or n1 >= n_s:
                break
            n1 = int(min(n_s, max(n1 + 1, n1 * min(64.0, 1.6 / max(sec1, 1e-3)))))
        secn, ostn = run(s1b, s1o, cores)
        cpu["single_thread"] = {"value": round(n1 / sec1, 1), "hits_per_s": round(ost1["hits"] / sec1, 1),
                                "sample": f"the first {n1} queries of the same seeded draw (a subset of the sample above), {ost1['hits']} hits, {sec1:.2f}s wall",
                                "same_queries_all_threads": {"value": round(n1 / secn, 1), "hits_per_s": round(ostn["hits"] / secn, 1), "seconds": round(secn, 3)},
                                "speedup_all_threads": round(sec1 / max(secn, 1e-9), 2)}
    if not retain:      # the same pass without the contended refcount bumps, for scale
        sec_p, ost_p = o.matches_timed(sb, so, cores, refcounted=False)
        cpu["without_refcounting"] = {"value": round(n_s / sec_p, 1), "hits_per_s": round(ost_p["hits"] / sec_p, 1),
                                      "what": "same pass with plain pointer copies instead of ref-counted clones (no atomic increments on hot ClientIds)"}
        cpu["scaling_note"] = ("the threads share the ref counts of the hot ClientIds (the reference clones an Arc-backed ClientId per hit, router.rs:226): at Zipf "
                               "fan-out the all-threads figure is bounded by that cache-line contention, not by cores — `without_refcounting` is the same pass "
                               "without the atomic increments")
    return cpu


def shard_inputs(W, world, rank):
    """This rank's share of config `W` under the shard rule (rgr_shard_assign): the filters it owns plus the replicated
    (wildcard-in-the-key-levels) ones, and the publish topics it owns.  -> (blob, offs, sub_ids, qos, topic blob, topic offs, keep_t)"""
    from rmqtt_amd import shard
    sub_ids = np.arange(W["n_sub"], dtype=np.uint32)
    if world == 1:
        return W["blob"], W["offs"], sub_ids, W["qos"], W["tb"], W["to"], None
    f_owner = shard.assign(W["blob"], W["offs"], world, is_filter=True)
    t_owner = shard.assign(W["tb"], W["to"], world, is_filter=False)
    keep_f = np.nonzero((f_owner == rank) | (f_owner < 0))[0]
    keep_t = np.nonzero(t_owner == rank)[0]
    blob_r, offs_r = shard.take(W["blob"], W["offs"], keep_f)
    tb_r, to_r = shard.take(W["tb"], W["to"], keep_t)
    return blob_r, offs_r, sub_ids[keep_f], W["qos"][keep_f], tb_r, to_r, keep_t


def deliver_flags(n, deliver_frac):
    """RGR_SUB_* flags of the delivery-stage workload: `deliver_frac` of the subscriptions are MQTT v5, 30 % of those No Local, 50 % RAP (seeded)."""
    from rmqtt_amd import capi
    drng = np.random.default_rng(11)
    is5 = drng.random(n) < deliver_frac
    return (is5 * capi.RGR_SUB_V5 | (is5 & (drng.random(n) < 0.3)) * capi.RGR_SUB_NO_LOCAL | (is5 & (drng.random(n) < 0.5)) * capi.RGR_SUB_RAP).astype(np.uint8)


DELIVER_SECONDARY_V5 = 0.1          # v5 fraction of the delivery-stage secondary record
_ORACLES = {}                       # (cfg, scale) -> oracle DefaultRouter kept for the delivery record of the same run (its build is 30 s at config 3)


def build_table(r, W, blob, offs, sub_ids, qos, deliver_frac=-1.0):
    from rmqtt_amd import capi
    t = time.time()
    if W["retain"]:
        rej = r.retain_add_bulk(blob, offs)
        r.retain_commit()
    else:
        flags = None
        if deliver_frac >= 0:
            flags = deliver_flags(len(offs) - 1, deliver_frac)
            W["deliver_flags"] = flags
        rej = r.subscribe_bulk(blob, offs, sub_ids, qos, flags)
        if deliver_frac >= 0:
            r.sub_attrs_bulk(W["client"].astype(np.uint32), W["client"].astype(np.uint32))     # one Id per client
        r.commit()
    return rej, time.time() - t


# ------------------------------------------------------------------------------------------- GPU digests
class _DevArr:      # zero-copy torch view of library-owned device memory
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (int(ptr), False), "version": 2}


FORMAT_NAMES = ("tuple", "soa", "packed", "runs", "ids24")      # == RGR_FORMAT_* (plain passes)
DELIVERY_FORMAT_NAMES = {0: "tuple12: (topic_idx, sub_id, delivery word)", 5: "hits8: (sub_id, delivery word), topic implied by the CSR offsets"}


def gpu_digests(batch, n_topics, retain, fmt=0, subs_len=0, topic_ids_dev=None, qos_by_sub=None, retain_vals_dev=None):
    """Per-topic digests of EVERY window of one full pass in result format `fmt`, reduced on the device (torch is plumbing
    here: the hits were produced by the library's kernels; the per-topic sums are prefix-sum differences because a topic's
    hits are consecutive positions).  -> (int64 device tensor [n, 4] (router) / [n, 3] (retain) with the same definition as
    oracle.cpp's orc_router_match_digest / orc_retain_match_digest, structure_ok, info).  Structural checks: the tuple
    format's topic_idx column / the run format's topic column must restate the CSR offsets; windows must tile the batch."""
    import torch
    from rmqtt_amd import capi
    ncol = 3 if retain else 4
    out = torch.zeros((n_topics, ncol), dtype=torch.int64, device="cuda")
    runs_per_topic = torch.zeros(n_topics, dtype=torch.int32, device="cuda") if fmt == capi.RGR_FORMAT_RUNS else None
    structure_ok = True
    info = {"windows": 0, "last_window": (0, 0)}
    batch.set_format(fmt)
    batch.begin()
    expect_begin = 0
    M32 = 0xFFFFFFFF
    # a batch in walk order (rgr_batch_set_order): windows enumerate walk positions; order_dev[k] = batch topic walked k-th, inv_dev its inverse
    perm = None if retain else batch.topic_order()
    order_dev = inv_dev = None
    if perm is not None:
        order_dev = torch.from_numpy(perm.astype(np.int64)).cuda()
        inv_dev = torch.empty_like(order_dev)
        inv_dev[order_dev] = torch.arange(n_topics, dtype=torch.int64, device="cuda")
    while True:
        w = batch.next_window()
        if w is None:
            break
        tb_, te_ = int(w.topic_begin), int(w.topic_end)
        structure_ok &= tb_ == expect_begin and te_ > tb_
        expect_begin = te_
        nt, nh = te_ - tb_, int(w.n_hits)
        info["windows"] += 1
        info["last_window"] = (tb_, te_)
        torch.cuda.synchronize()                    # the library expands on its own stream
        if not nh:
            continue
        d_off = torch.as_tensor(_DevArr(w.d_hit_offsets, (nt + 1,), "<i8"), device="cuda") - int(w.offsets_bias)
        structure_ok &= int(d_off[0]) == 0 and int(d_off[-1]) == nh
        start, end = d_off[:-1], d_off[1:]
        cnt = end - start
        rows = order_dev[tb_:te_] if order_dev is not None else None          # batch topics of the window's positions
        if order_dev is not None:
            structure_ok &= bool(w.d_topic_order) and bool((torch.as_tensor(_DevArr(w.d_topic_order, (nt,), "<i4"), device="cuda").to(torch.int64) == rows).all())
        if fmt == capi.RGR_FORMAT_TUPLE:
            t = torch.as_tensor(_DevArr(w.d_tuples, (nh, 3), "<i4"), device="cuda")
            owner = torch.repeat_interleave(rows.to(torch.int32) if rows is not None else torch.arange(tb_, te_, dtype=torch.int32, device="cuda"), cnt)
            if topic_ids_dev is not None:              # sharded batch (rgr_batch_set_topic_ids): the column carries the GLOBAL publish index
                owner = topic_ids_dev[owner.to(torch.int64)]
            structure_ok &= bool((t[:, 0] == owner).all())       # tuple i names the topic whose CSR range holds position i
            del owner
            sid = t[:, 1].to(torch.int64) & M32
            q = t[:, 2].to(torch.int64) & 0xFF
            if retain_vals_dev is not None:            # retained path answering with positions (rgr_batch_set_retain_positions): id = vals[position]
                sid = retain_vals_dev[sid]
        elif fmt == capi.RGR_FORMAT_SOA:
            sid = torch.as_tensor(_DevArr(w.d_sub_ids, (nh,), "<i4"), device="cuda").to(torch.int64) & M32
            q = torch.as_tensor(_DevArr(w.d_qos, (nh,), "|u1"), device="cuda").to(torch.int64) & 3
        elif fmt == capi.RGR_FORMAT_PACKED:
            x = torch.as_tensor(_DevArr(w.d_sub_ids, (nh,), "<i4"), device="cuda").to(torch.int64) & M32
            sid, q = x & 0x3FFFFFFF, x >> 30
        elif fmt == capi.RGR_FORMAT_IDS24:           # 3 little-endian bytes per hit; the qos is table data, looked up by sub id
            x = torch.as_tensor(_DevArr(w.d_ids24, (nh, 3), "|u1"), device="cuda").to(torch.int64)
            sid = x[:, 0] | (x[:, 1] << 8) | (x[:, 2] << 16)
            q = qos_by_sub[sid] if qos_by_sub is not None else torch.zeros_like(sid)
            del x
        else:                                        # runs: the hits are read in place from the epoch's subs[]
            nr = int(w.n_runs)
            src = torch.as_tensor(_DevArr(w.d_run_src, (nr,), "<i4"), device="cuda").to(torch.int64) & M32
            rtp = torch.as_tensor(_DevArr(w.d_run_topic, (nr,), "<i4"), device="cuda").to(torch.int64) & M32
            roff = torch.as_tensor(_DevArr(w.d_run_off, (nr + 1,), "<i8"), device="cuda") - int(w.offsets_bias)
            lens = roff[1:] - roff[:-1]
            structure_ok &= int(roff[0]) == 0 and int(roff[-1]) == nh and bool((lens > 0).all())
            loc = (inv_dev[rtp] if inv_dev is not None else rtp) - tb_
            if bool(((loc < 0) | (loc >= nt)).any()):
                structure_ok = False
                continue
            structure_ok &= bool(((roff[:-1] >= start[loc]) & (roff[1:] <= end[loc])).all())     # a run lies inside its topic's range
            if rows is not None:
                runs_per_topic[rows] += torch.bincount(loc, minlength=nt).to(torch.int32)
            else:
                runs_per_topic[tb_:te_] += torch.bincount(loc, minlength=nt).to(torch.int32)
            idx = torch.repeat_interleave(src - roff[:-1], lens) + torch.arange(nh, dtype=torch.int64, device="cuda")
            hi = int(idx.max()) + 1
            e = torch.as_tensor(_DevArr(w.d_subs, (max(hi, subs_len),), "<i8"), device="cuda")[idx]
            sid, q = e & M32, (e >> 32) & 0xFF
            del idx, e, src, rtp, roff, lens, loc

        def seg(x):                                  # per-topic sums of x (mod 2^64)
            cs = torch.cumsum(x, 0)
            hi_ = torch.where(end > 0, cs[(end - 1).clamp(min=0)], torch.zeros_like(end))
            lo_ = torch.where(start > 0, cs[(start - 1).clamp(min=0)], torch.zeros_like(start))
            return hi_ - lo_
        acc = torch.zeros((nt, ncol), dtype=torch.int64, device="cuda") if rows is not None else out[tb_:te_]
        acc[:, 0] = cnt
        if retain:
            acc[:, 1] = seg(sid)
            acc[:, 2] = seg(sid * sid)
        else:
            v = sid * 4 + q
            s1 = seg(v)
            acc[:, 1] = s1
            acc[:, 2] = seg(torch.arange(1, nh + 1, dtype=torch.int64, device="cuda") * v) - start * s1     # sum (k+1) v_k, k = position in the topic
            acc[:, 3] = seg(v * v)
            del v, s1
        if rows is not None:
            out[rows] = acc
        del sid, q, d_off, start, end, cnt
        torch.cuda.synchronize()                    # torch's reads of the window must finish before the library expands the next one into the same buffer
    structure_ok &= expect_begin == n_topics
    batch.set_format(capi.RGR_FORMAT_TUPLE)
    info["runs_per_topic"] = runs_per_topic
    return out, bool(structure_ok), info


def _delivery_window_columns(w, fmt):
    """(topic, sub_id, word) int64 device columns of one delivery window in either delivery format: 12-byte tuples carry the topic, 8-byte hits
    (RGR_FORMAT_DELIVER8) leave it to the CSR offsets."""
    import torch
    from rmqtt_amd import capi
    nh, tb_, te_ = int(w.n_hits), int(w.topic_begin), int(w.topic_end)
    if fmt == capi.RGR_FORMAT_DELIVER8:
        t = torch.as_tensor(_DevArr(w.d_hits8, (nh, 2), "<i4"), device="cuda")
        d_off = torch.as_tensor(_DevArr(w.d_hit_offsets, (te_ - tb_ + 1,), "<i8"), device="cuda") - int(w.offsets_bias)
        # (a batch in walk order: the window's k-th topic is batch topic d_topic_order[k])
        tids = (torch.as_tensor(_DevArr(w.d_topic_order, (te_ - tb_,), "<i4"), device="cuda").to(torch.int64) if w.d_topic_order
                else torch.arange(tb_, te_, dtype=torch.int64, device="cuda"))
        topic = torch.repeat_interleave(tids, d_off[1:] - d_off[:-1])
        return topic, t[:, 0].to(torch.int64) & 0xFFFFFFFF, t[:, 1].to(torch.int64) & 0xFFFFFFFF
    t = torch.as_tensor(_DevArr(w.d_tuples, (nh, 3), "<i4"), device="cuda")
    return t[:, 0].to(torch.int64) & 0xFFFFFFFF, t[:, 1].to(torch.int64) & 0xFFFFFFFF, t[:, 2].to(torch.int64) & 0xFFFFFFFF


def delivery_parity(batch, W, pa, n_windows_wanted=3, fmt=0):
    """Full-size check of the delivery stage, in the bench line: the delivery words of whole windows of the
    timed pass — first, middle, last — against a restatement of the per-hit rules in torch on the device, independent of the
    library's kernels: qos' = min(publish, subscription), Retain-As-Published, No Local (owner == publisher), and the v5
    collector's first hit per (topic, client) in position order (types.rs:524-539) as a sort-free min-position reduction.
    (The oracle's forwards() pins the same rules at small sizes: tests/test_deliver_parity.py.)"""
    import torch
    from rmqtt_amd import capi
    qos = torch.from_numpy(np.ascontiguousarray(W["qos"]).astype(np.int64)).cuda()
    flags = torch.from_numpy(W["deliver_flags"].astype(np.int64)).cuda()
    client = torch.from_numpy(np.ascontiguousarray(W["client"]).astype(np.int64)).cuda()            # owner id == client index in this bench
    p_from = torch.from_numpy(pa["from_id"].astype(np.int64)).cuda()
    p_qr = torch.from_numpy(pa["qos_retain"].astype(np.int64)).cuda()
    # which windows: count them with a pass of the SAME kind (a delivery pass has its own window size — 2^27 hits, a plain device-resident
    # pass 2^30: round 5's record counted the windows of a plain pass and so checked windows 0 / 69 / 138 of 1 106, all in the first eighth)
    batch.set_publish_attrs(pa)
    batch.set_format(fmt)
    _, nwin = batch.run()
    want = sorted({0, nwin // 2, nwin - 1})[:n_windows_wanted]
    checked, bad, hits, dups, drops = [], 0, 0, 0, 0
    last_is_partial = None
    last_window_topics = None
    batch.begin()
    wi = -1
    max_hits_seen = 0
    while True:
        w = batch.next_window()
        if w is None:
            break
        wi += 1
        if wi not in want or not w.n_hits:
            max_hits_seen = max(max_hits_seen, int(w.n_hits))
            continue
        torch.cuda.synchronize()
        nh = int(w.n_hits)
        topic, sid, got = _delivery_window_columns(w, fmt)
        fl = flags[sid]
        pq = p_qr[topic]
        is5 = (fl & capi.RGR_SUB_V5) != 0
        exp = (fl << 8) | torch.minimum(qos[sid], pq & 3)
        exp = exp | torch.where(is5 & ((fl & capi.RGR_SUB_RAP) != 0) & ((pq & 4) != 0), capi.RGR_HIT_RETAIN, 0)
        drop = is5 & ((fl & capi.RGR_SUB_NO_LOCAL) != 0) & (client[sid] == p_from[topic])
        exp = exp | torch.where(drop, capi.RGR_HIT_NO_LOCAL, 0)
        cand = torch.nonzero(is5 & ~drop).squeeze(1)
        key = topic[cand] * (int(client.max()) + 1) + client[sid[cand]]
        uk, inv = torch.unique(key, return_inverse=True)
        first = torch.full((uk.numel(),), nh, dtype=torch.int64, device="cuda").scatter_reduce(0, inv, cand, reduce="amin")
        dup = first[inv] != cand
        di = cand[dup]
        exp.index_put_((di,), exp[di] | capi.RGR_HIT_V5_DUP)
        node_free = got & 0xFFFF                                  # (bits 16-31 carry the node index: 0 for bulk-loaded tables)
        bad += int((node_free != (exp & 0xFFFF)).sum()) + int(((got >> 16) != 0).sum())
        hits += nh; dups += int(dup.sum()); drops += int(drop.sum())
        checked.append(wi)
        if wi == nwin - 1:
            last_window_topics = [int(w.topic_begin), int(w.topic_end)]
            last_is_partial = bool(int(w.topic_end) == W["n_pub"] and nh < max_hits_seen)
        max_hits_seen = max(max_hits_seen, nh)
        del topic, sid, got, fl, pq, is5, exp, drop, cand, key, uk, inv, first, dup, di
        torch.cuda.synchronize()
    return {"format": DELIVERY_FORMAT_NAMES.get(fmt, fmt), "ok": bad == 0 and len(checked) == len(want) and wi + 1 == nwin, "windows_checked": checked, "of_windows": int(nwin), "windows_counted_by": "a delivery pass of the same batch",
            "last_window": {"index": int(nwin) - 1, "topics": last_window_topics, "ends_the_batch_and_is_partial": last_is_partial},
            "hits": int(hits), "v5_duplicates_flagged": int(dups),
            "no_local_drops": int(drops), "mismatching_words": int(bad),
            "what": "delivery words of whole windows of the timed delivery pass (first / middle / last, the last one being the batch's partial tail) vs a torch "
                    "restatement of the per-hit rules on the device; the oracle's own verdict on a stratified sample is `oracle` below"}


def device_digests(r, batch, n, retain, formats=True, topic_ids=None, qos=None):
    """ONE full pass of the timed batch per result format (tuple, soa, packed, runs), every window digested per topic on the device.
    The compact formats' digests must equal the tuple format's for EVERY topic, and each pass is checked structurally (windows
    tile the batch, topic columns restate the CSR offsets; with `topic_ids` — a sharded batch — the tuple's topic column must
    carry the caller's global index).  -> (tuple-format digests [n, ncol] on the device, per-format verdicts, info)"""
    import torch
    from rmqtt_amd import capi
    t0 = time.time()
    subs_len = 0 if retain else int(r.stats()["n_subs"])
    ids_dev = torch.from_numpy(np.ascontiguousarray(topic_ids).astype(np.int32)).cuda() if topic_ids is not None else None
    D, fmt_ok, struct, runs_pt, info0 = None, {}, {}, None, {}
    fmts = (capi.RGR_FORMAT_TUPLE, capi.RGR_FORMAT_SOA, capi.RGR_FORMAT_PACKED, capi.RGR_FORMAT_RUNS, capi.RGR_FORMAT_IDS24) if formats else (capi.RGR_FORMAT_TUPLE,)
    qos_by_sub = torch.from_numpy(np.ascontiguousarray(qos).astype(np.int64)).cuda() if (qos is not None and not retain) else None
    for fmt in fmts:
        name = FORMAT_NAMES[fmt]
        if topic_ids is not None:          # only the tuple format's topic column is checked against the global ids; the other passes index locally
            batch.set_topic_ids(topic_ids if fmt == capi.RGR_FORMAT_TUPLE else None)
        try:
            d, ok_s, info = gpu_digests(batch, n, retain, fmt, subs_len, ids_dev if fmt == capi.RGR_FORMAT_TUPLE else None, qos_by_sub)
        except capi.RgrError as e:                   # (packed needs ids below 2^30)
            fmt_ok[name] = f"n/a: {e}"
            continue
        struct[name] = ok_s
        if fmt == capi.RGR_FORMAT_TUPLE:
            D, info0 = d, info
            fmt_ok[name] = None                      # decided by the oracle comparison
        else:
            fmt_ok[name] = "ok" if (bool((d == D).all()) and ok_s) else "MISMATCH"
            if fmt == capi.RGR_FORMAT_RUNS:
                runs_pt = info["runs_per_topic"]
            del d
    if retain and formats:
        # the retained path's position form: tuples carry positions of the epoch's preorder value array, resolved through the host mirror
        batch.set_retain_positions(True)
        batch.begin()
        while batch.next_window() is not None:
            pass
        vals = torch.from_numpy(batch.retain_vals()["topic_id"].astype(np.int64)).cuda()
        d, ok_s, _ = gpu_digests(batch, n, True, capi.RGR_FORMAT_TUPLE, retain_vals_dev=vals)
        batch.set_retain_positions(False)
        struct["positions"] = ok_s
        fmt_ok["positions"] = "ok" if (bool((d == D).all()) and ok_s) else "MISMATCH"
        del d, vals
    if topic_ids is not None:
        batch.set_topic_ids(topic_ids)
    return D, fmt_ok, {"structure_ok": struct, "windows": info0.get("windows", 0), "last_window": info0.get("last_window", (0, 0)),
                       "runs_per_topic": runs_pt, "gpu_digest_s": round(time.time() - t0, 2)}


def compare_with_oracle(o, W, got, gpu_status, fmt_ok, dinfo, threads, primary, seed=20260921, extra=None):
    """EXHAUSTIVE full-size parity: the device's per-topic digests of the tuple format (`got`, uint64 [n, ncol], batch order) against
    the oracle's digests of EVERY query of the batch on the FULL (unsharded) table.  The oracle composes a topic's digest from
    per-filter pre-reduced digests (orc_router_match_digest_fast: O(matched filters) per topic; the retained path uses bottom-up
    subtree aggregates for its '#' step, orc_retain_match_digest_fast) — held equal to its own O(hits) digest by
    tests/test_oracle_digest.py and, in this very run, on a stratified cross-check sample (heaviest topics + a random draw)."""
    from rmqtt_amd import shard
    retain = W["retain"]
    n = W["n_pub"]
    t1 = time.time()
    status, exp = o.match_digest(W["tb"], W["to"], threads, fast=True)
    cpu_s = time.time() - t1
    same_status = bool(np.array_equal(gpu_status < 0, status < 0))
    bad = np.nonzero((got != exp).any(axis=1))[0]
    struct = dinfo["structure_ok"]
    tuple_ok = same_status and struct.get("tuple", False) and len(bad) == 0
    fmt_ok = dict(fmt_ok)
    fmt_ok["tuple"] = "ok" if tuple_ok else "MISMATCH"
    # ---- the fast oracle digest against the oracle's own per-hit digest, on the heaviest queries and a random draw (bounded by hits)
    rng = np.random.default_rng(seed)
    hits = exp[:, 0].astype(np.int64)
    budget = (2.0e7 if retain else 4.0e8) * max(1, threads) / 256 * (1.0 if primary else 0.5)
    heavy = np.argsort(hits)[::-1][:64 if retain else 256]
    heavy = heavy[:max(1, int(np.searchsorted(np.cumsum(hits[heavy]), 0.5 * budget, side="right")))]
    n_rand = int(min(n, max(32, 0.5 * budget / max(1.0, float(hits.mean())))))
    sel = np.unique(np.concatenate([heavy, rng.choice(n, size=n_rand, replace=False)]))
    sb, so = shard.take(W["tb"], W["to"], sel)
    t2 = time.time()
    st_s, exp_s = o.match_digest(sb, so, threads)
    cross_s = time.time() - t2
    cross_ok = bool(np.array_equal(exp_s, exp[sel]) and np.array_equal(st_s < 0, status[sel] < 0))
    ok = tuple_ok and cross_ok and all(v == "ok" or str(v).startswith("n/a") for v in fmt_ok.values())
    rec = {"topics": int(n), "exhaustive": True, "hits": int(hits.sum()), "ok": bool(ok), "formats": fmt_ok,
           "invalid_topics": int((status < 0).sum()),
           "full_pass": {"topics_digested": int(n), "hits_digested": int(got[:, 0].astype(np.int64).sum()), "windows": int(dinfo["windows"]),
                         "what": "every window of one full pass of the timed batch per format, digested per topic on the device; compact formats "
                                 "compared with the tuple format on ALL topics; tuple format compared with the oracle on ALL topics"},
           "oracle_cross_check": {"ok": cross_ok, "topics": int(len(sel)), "hits": int(exp_s[:, 0].sum()), "seconds": round(cross_s, 2),
                                  "what": "orc_*_match_digest_fast (used for all topics) vs the oracle's per-hit digest on the heaviest queries + a seeded random draw"},
           "max_hits_in_one_query": int(hits.max()) if n else 0,
           "digest": ("per filter: hits, sum(topic_id), sum(topic_id^2) mod 2^64 (set comparison: the reference's order is hash-map order)"
                      if retain else
                      "per topic: hits, sum(v), sum((k+1)*v) in canonical order, sum(v^2) mod 2^64, v = sub_id*4+qos"),
           "table": "full", "oracle_s": round(cpu_s, 2), "gpu_digest_s": dinfo["gpu_digest_s"], "seed": seed}
    if extra:
        rec.update(extra)
    if not ok:
        rec["first_bad_topic"] = int(bad[0]) if len(bad) else None
        rec["mismatching_topics"] = int(len(bad))
        rec["status_equal"] = same_status
        rec["structure_ok"] = struct
    return rec


def delivery_oracle_sample(o, W, batch, pa, threads, seed=20260923, fmt=0):
    """Delivery words of the timed batch against the ORACLE at full table size: one more full pass; per topic the digest of
    x = sub_id * 32 + (word & 31) over its hits in position order, reduced on the device, compared with DefaultRouter::deliver_digest
    (which hits are delivered comes from the oracle's matches(): the restated _matches + collector) for a stratified sample — the
    heaviest topics, the topics of the first / last window ends, and a seeded random draw — bounded by the oracle's cost (O(hits) with
    a row per delivered hit)."""
    import torch
    from rmqtt_amd import capi, shard
    n = W["n_pub"]
    t0 = time.time()
    D = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    batch.set_format(fmt)
    batch.begin()
    while True:
        w = batch.next_window()
        if w is None:
            break
        nh, tb_, te_ = int(w.n_hits), int(w.topic_begin), int(w.topic_end)
        torch.cuda.synchronize()
        if not nh:
            continue
        d_off = torch.as_tensor(_DevArr(w.d_hit_offsets, (te_ - tb_ + 1,), "<i8"), device="cuda") - int(w.offsets_bias)
        start, end = d_off[:-1], d_off[1:]
        if fmt == capi.RGR_FORMAT_DELIVER8:
            t = torch.as_tensor(_DevArr(w.d_hits8, (nh, 2), "<i4"), device="cuda")
            x = (t[:, 0].to(torch.int64) & 0xFFFFFFFF) * 32 + (t[:, 1].to(torch.int64) & 31)
        else:
            t = torch.as_tensor(_DevArr(w.d_tuples, (nh, 3), "<i4"), device="cuda")
            x = (t[:, 1].to(torch.int64) & 0xFFFFFFFF) * 32 + (t[:, 2].to(torch.int64) & 31)

        def seg(v):
            cs = torch.cumsum(v, 0)
            hi_ = torch.where(end > 0, cs[(end - 1).clamp(min=0)], torch.zeros_like(end))
            lo_ = torch.where(start > 0, cs[(start - 1).clamp(min=0)], torch.zeros_like(start))
            return hi_ - lo_
        rows = torch.as_tensor(_DevArr(w.d_topic_order, (te_ - tb_,), "<i4"), device="cuda").to(torch.int64) if w.d_topic_order else None
        acc = torch.zeros((te_ - tb_, 4), dtype=torch.int64, device="cuda") if rows is not None else D[tb_:te_]
        acc[:, 0] = end - start
        s1 = seg(x)
        acc[:, 1] = s1
        acc[:, 2] = seg(torch.arange(1, nh + 1, dtype=torch.int64, device="cuda") * x) - start * s1
        acc[:, 3] = seg(x * x)
        if rows is not None:
            D[rows] = acc
        del t, x, s1, d_off, start, end
        torch.cuda.synchronize()
    gpu_s = time.time() - t0
    hits = D[:, 0].cpu().numpy()
    rng = np.random.default_rng(seed)
    # >= 1 % of the batch's hits when the host affords it: the oracle's per-hit delivery pass runs ~50 M hits/s on 256 threads (~30 s for 1.5 G hits);
    # a smaller host checks proportionally less and the record says how much (`share_of_hits`)
    total = float(hits.sum())
    budget = min(max(0.0115 * total, 2.5e8), 1.8e9 * max(1, threads) / 256)
    heavy = np.argsort(hits)[::-1][:256]
    heavy = heavy[:max(1, int(np.searchsorted(np.cumsum(hits[heavy]), 0.1 * budget, side="right")))]
    n_rand = int(min(n, max(64, 0.92 * budget / max(1.0, float(hits.mean())))))
    sel = np.unique(np.concatenate([heavy, np.arange(min(n, 64)), np.arange(max(0, n - 64), n), rng.choice(n, size=n_rand, replace=False)]))
    sb, so = shard.take(W["tb"], W["to"], sel)
    t1 = time.time()
    st, exp = o.deliver_digest(sb, so, pa["from_id"][sel], pa["qos_retain"][sel].astype(np.uint8), threads)
    cpu_s = time.time() - t1
    got = D[torch.from_numpy(sel).cuda()].cpu().numpy().view(np.uint64)
    gst = batch.status()[sel]
    bad = np.nonzero((got != exp).any(axis=1))[0]
    ok = bool(np.array_equal(gst < 0, st < 0) and len(bad) == 0)
    rec = {"format": DELIVERY_FORMAT_NAMES.get(fmt, fmt), "ok": ok, "topics": int(len(sel)), "hits": int(exp[:, 0].sum()), "max_hits_in_one_topic": int(exp[:, 0].max()) if len(sel) else 0,
           "share_of_hits": round(float(exp[:, 0].sum()) / max(1.0, total), 5), "share_of_topics": round(len(sel) / max(1, n), 5),
           "oracle_s": round(cpu_s, 2), "gpu_digest_s": round(gpu_s, 2),
           "what": "per topic: hits, sum x, sum (k+1) x, sum x^2 with x = sub_id*32 + (delivery word & 31), device vs the oracle's DefaultRouter::deliver_digest "
                   "(delivered hits = the rows of its matches()) on the heaviest topics, both ends of the batch and a seeded random draw; full table"}
    if not ok:
        rec["first_bad_topic"] = int(sel[bad[0]]) if len(bad) else None
        rec["mismatching_topics"] = int(len(bad))
    del D
    return rec


def parity_sample(r, o, W, batch, threads, primary):
    """N = 1: device digests of the whole timed batch in every format, then the exhaustive comparison with the oracle."""
    D, fmt_ok, dinfo = device_digests(r, batch, W["n_pub"], W["retain"], qos=W["qos"])
    got = D.cpu().numpy().view(np.uint64)
    del D
    return compare_with_oracle(o, W, got, batch.status(), fmt_ok, dinfo, threads, primary)


def sharded_digests(r, batch, keep_t, n_pub, world, rank, dist, cdev, formats=True, qos=None):
    """N > 1 (collective): every rank digests ITS topics on its device (all formats; the tuple pass also checks that the tuples carry
    the global publish index), rank 0 receives every rank's rows and assembles the digests of the whole batch in batch order.
    -> on rank 0: (uint64 [n_pub, 4], status int32 [n_pub], per-format verdicts, info); None elsewhere."""
    import torch
    my = len(keep_t)
    D, fmt_ok, dinfo = device_digests(r, batch, my, False, formats=formats, topic_ids=keep_t.astype(np.uint32), qos=qos)
    rows = torch.zeros((my, 6), dtype=torch.int64)
    rows[:, 0] = torch.from_numpy(keep_t.astype(np.int64))
    rows[:, 1] = torch.from_numpy(batch.status().astype(np.int64))
    rows[:, 2:] = D.cpu()
    del D
    rows = rows.to(cdev)
    flags = torch.tensor([my, int(all(dinfo["structure_ok"].values())), int(all(v in (None, "ok") or str(v).startswith("n/a") for v in fmt_ok.values())),
                          int(dinfo["windows"])], dtype=torch.int64, device=cdev)
    allf = [torch.zeros_like(flags) for _ in range(world)]
    dist.all_gather(allf, flags)
    counts = [int(f[0]) for f in allf]
    pad = torch.zeros((max(counts + [1]), 6), dtype=torch.int64, device=cdev)
    pad[:my] = rows
    bufs = [torch.zeros_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0)
    if rank != 0:
        return None
    got = np.zeros((n_pub, 4), dtype=np.uint64)
    status = np.zeros(n_pub, dtype=np.int32)
    seen = np.zeros(n_pub, dtype=np.int32)
    for b_, c_ in zip(bufs, counts):
        x = b_[:c_].cpu().numpy()
        ids = x[:, 0]
        np.add.at(seen, ids, 1)
        got[ids] = x[:, 2:].view(np.uint64)
        status[ids] = x[:, 1].astype(np.int32)
    fmt_all = {k: ("ok" if all(int(f[2]) for f in allf) else "MISMATCH") for k in fmt_ok if k != "tuple"}
    fmt_all["tuple"] = None
    info = {"structure_ok": {"tuple": bool(all(int(f[1]) for f in allf)) and bool((seen == 1).all())},
            "windows": int(sum(int(f[3]) for f in allf)), "gpu_digest_s": dinfo["gpu_digest_s"]}
    extra = {"sharded": {"ranks": world, "topics_per_rank": counts, "every_topic_owned_exactly_once": bool((seen == 1).all()),
                         "what": "each rank digested its own topics on its device (tuples carry the global publish index); rank 0 compared the "
                                 "assembled batch with the oracle on the UNSHARDED table"}}
    return got, status, fmt_all, info, extra


# ------------------------------------------------------------------------------------------- PMC traffic
# kernel classes of the PMC replay, by name.  "expand" = every tuple expansion launch_expand can choose (kernels.hip): the plain kernel and
# the delivery variants (expand_kernel<true>, expand_deliver_early_kernel, expand_deliver_lean_kernel) — r5p's first driver-style run lost its
# whole roofline record because the delivery phase's new default kernel matched no needle (tests/test_bench_line.py pins the list now).
KCLASS = (("expand", ("expand_kernel", "expand_deliver_")), ("walk", ("walk_kernel<false>",)), ("retain", ("retain_",)))


def run_pmc_children(args, phases, world=1, rank=0):
    """FETCH_SIZE and WRITE_SIZE of this run's kernels: two rocprofv3 passes (the TCC counters do not fit one)
    over `bench.py --pmc-child`, which replays ONE pass of every phase.  Counters only + kernel trace — no
    sys/hip/hsa trace domains.  -> {phase: {class: {"fetch_KiB", "write_KiB", "dispatches", "avg_us"}}} or None."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        log("pmc: rocprofv3 not found — roofline.traffic stays null")
        return None
    work = tempfile.mkdtemp(prefix="rgr_pmc_")
    res = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            meta = os.path.join(work, f"{counter}.json")
            cmd = [rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(work, counter), "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", meta, "--pmc-phases", ",".join(phases),
                   "--config", str(args.config), "--scale", str(args.scale), "--pmc-topics", str(args.pmc_topics),
                   "--pmc-world", str(world), "--pmc-rank", str(rank), "--deliver-format", args.deliver_format, "--topic-order", args.topic_order]
            if args.window_hits:
                cmd += ["--window-hits", str(args.window_hits)]
            t = time.time()
            env = dict(os.environ, TMPDIR=tempfile.gettempdir())
            for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
                env.pop(k, None)            # the child is a plain single-process replay
            p = subprocess.run(cmd, cwd=tempfile.gettempdir(), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
            log(f"pmc: {counter} pass rc={p.returncode} in {time.time() - t:.0f}s")
            if p.returncode != 0 or not os.path.exists(meta):
                log("pmc: child failed: " + p.stderr.decode(errors="replace")[-600:])
                return None
            plan = json.load(open(meta))["phases"]
            files = glob.glob(os.path.join(work, counter, "**", "*counter_collection.csv"), recursive=True)
            rows = []
            for f in files:
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == counter:
                        rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"], float(row["Counter_Value"]),
                                     int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
            rows.sort()
            for cls, needles in KCLASS:
                seq = [x for x in rows if any(nd in x[1] for nd in needles)]
                at = 0
                for ph in plan:
                    # the retain_* kernels' dispatch count is only known from the trace: they all belong to the (one) retain phase
                    k = (len(seq) if ph.get("retain") else 0) if cls == "retain" else int(ph["dispatches"].get(cls, 0))
                    part = seq[at:at + k]
                    at += k
                    if len(part) != k:
                        # (this class's later phases cannot be attributed either; the phases before it and the other classes keep their numbers —
                        # a mismatch in a secondary's phase used to null the headline's roofline as well)
                        log(f"pmc: {counter}/{cls}: expected {k} dispatches in phase {ph['name']}, trace has {len(part)} — this class stops here")
                        at = len(seq)
                        break
                    d = res.setdefault(ph["name"], {}).setdefault(cls, {"dispatches": k, "hits": ph["hits"], "topics": ph["topics"]})
                    d[("fetch" if counter == "FETCH_SIZE" else "write") + "_KiB"] = sum(x[2] for x in part)
                    d["avg_us_under_pmc"] = round(sum(x[3] for x in part) / max(1, k) / 1e3, 2)
                if at != len(seq):
                    log(f"pmc: {counter}/{cls}: {len(seq) - at} dispatches not attributed to a phase")
        return res
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
        log(f"pmc: failed ({e}) — roofline.traffic stays null")
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def pmc_child(args):
    """One pass per phase under rocprofv3's counters; writes how many dispatches of each kernel class every
    phase issued so that the parent can split the counter rows."""
    from rmqtt_amd import capi, shard
    phases = []
    for name in args.pmc_phases.split(","):
        deliver = name == "deliver"           # the delivery-stage secondary: config 3 with v5 subscriptions and publish attributes
        cfg, scale = (args.config, args.scale) if name == "primary" else ((3, 1.0) if deliver else (int(name[6:]), 1.0))
        W = gen_workload(cfg, scale)
        r = capi.Router(device=0, window_hits=args.window_hits, collect_walk_stats=False)
        world = args.pmc_world if name == "primary" else 1
        blob_r, offs_r, sub_ids_r, qos_r, tb_r, to_r, _ = shard_inputs(W, world, args.pmc_rank if world > 1 else 0)
        build_table(r, W, blob_r, offs_r, sub_ids_r, qos_r, DELIVER_SECONDARY_V5 if deliver else -1.0)
        n_mine = len(to_r) - 1
        n = min(n_mine, args.pmc_topics)
        sb, so = shard.take(tb_r, to_r, np.arange(n)) if n < n_mine else (tb_r, to_r)
        b = r.retain_batch(sb, so) if W["retain"] else r.batch(sb, so)
        if args.topic_order == "walk" and not W["retain"] and (not deliver or args.deliver_format == "hits8"):
            b.set_order(True)
        if deliver:
            pa = np.zeros(n, dtype=capi.PUBLISH_ATTR_DTYPE)
            prng = np.random.default_rng(12)
            pa["from_id"] = prng.choice(W["client"].astype(np.uint32), size=n)
            pa["qos_retain"] = prng.integers(0, 3, size=n) | (prng.integers(0, 2, size=n) << 2)
            b.set_publish_attrs(pa)
            if args.deliver_format == "hits8":
                b.set_format(capi.RGR_FORMAT_DELIVER8)
        r.stats_reset()
        hits, _ = b.run()
        st = r.stats()
        # dispatches per class: one expand_kernel per window with hits; walk_kernel<false> once per chunk
        # (router) — the retain path has no walk kernel, its retain_* kernels are counted as one class
        disp = {"expand": int(st["expand_launches"]), "walk": 0 if W["retain"] else int(st["walk_launches"])}
        phases.append({"name": name, "dispatches": disp, "hits": int(hits), "topics": int(n), "retain": bool(W["retain"])})
        b.close(); r.close()
    json.dump({"phases": phases}, open(args.pmc_child, "w"))


def load_calibration():
    """FETCH_SIZE scale factors per access pattern, measured with tools/membench.hip under rocprofv3
    (profiles/pmc_calibration.json); 1.0 (raw) where no calibration exists."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_calibration.json")))
    except (OSError, ValueError):
        return {}


# ------------------------------------------------------------------------------------------- one config
def measure(args, cfg, scale, steps, warmup, primary, rank=0, world=1, local_rank=0, dist=None, cdev="cuda", deliver_frac=None):
    """Build the table of one BASELINE config, time `steps` passes, and (N=1) run the parity sample, the
    PCIe-inclusive pass and the CPU baseline.  -> (record dict, phase info for the PMC children)."""
    import torch
    from rmqtt_amd import capi, shard
    from rmqtt_amd import workload as wl

    W = gen_workload(cfg, scale, rank)
    c, n_sub, n_pub, retain = W["c"], W["n_sub"], W["n_pub"], W["retain"]
    blob, offs, tb, to, client, qos = W["blob"], W["offs"], W["tb"], W["to"], W["client"], W["qos"]
    if world > 1 and retain:
        raise SystemExit("config 5 (retained path) is a single-GPU config")
    blob_r, offs_r, sub_ids_r, qos_r, tb_r, to_r, keep_t = shard_inputs(W, world, rank)
    my_topics = len(to_r) - 1

    # ---- table build + device-resident batch
    deliver = deliver_frac if deliver_frac is not None else (args.deliver if (primary and world == 1 and not retain) else -1.0)
    r = capi.Router(device=local_rank, window_hits=args.window_hits, collect_walk_stats=True)
    rej, build_s = build_table(r, W, blob_r, offs_r, sub_ids_r, qos_r, deliver)
    st0 = r.stats()
    if retain:
        log(f"config {cfg}: retain table: {n_sub - rej} topics ({rej} names rejected by the parser), built in {build_s:.1f}s", rank)
    else:
        log(f"config {cfg}: table: {st0['n_filters']} filters, {st0['n_subs']} subs, {st0['n_nodes']} trie nodes, "
            f"{st0['table_bytes_device'] / 2**30:.2f} GiB in HBM, built in {build_s:.1f}s (rejected {rej})", rank)
    t = time.time()
    batch = r.retain_batch(tb_r, to_r) if retain else r.batch(tb_r, to_r)
    walk_order = args.topic_order == "walk" and not retain and (deliver < 0 or args.deliver_format == "hits8")
    if walk_order:
        # the library sorts the batch by its leading tokens and walks it in that order (rgr_batch_set_order): part of preparing the batch, like the tokeniser
        batch.set_order(True)
    batch_create_s = time.time() - t
    log(f"config {cfg}: batch: {my_topics} topics tokenised + uploaded{' + sorted into walk order' if walk_order else ''} in {batch_create_s:.1f}s", rank)
    dfmt = capi.RGR_FORMAT_TUPLE
    if deliver >= 0:
        pa = np.zeros(my_topics, dtype=capi.PUBLISH_ATTR_DTYPE)
        prng = np.random.default_rng(12)
        pa["from_id"] = prng.choice(client.astype(np.uint32), size=my_topics)
        pa["qos_retain"] = prng.integers(0, 3, size=my_topics) | (prng.integers(0, 2, size=my_topics) << 2)
        batch.set_publish_attrs(pa)
        # the delivery stage's answer: 8-byte hits {sub_id, delivery word} by default (RGR_FORMAT_DELIVER8: the topic is implied by the CSR
        # offsets, as in the compact formats of plain passes), the 12-byte tuple form timed beside it
        dfmt = capi.RGR_FORMAT_DELIVER8 if args.deliver_format == "hits8" else capi.RGR_FORMAT_TUPLE
        batch.set_format(dfmt)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step_gather_tuples():
        """all-gatherv of every window's tuples (RCCL has no allgatherv: counts + padded all_gather)."""
        hits = nwin = 0
        batch.begin()
        finished = False
        while True:
            w = None if finished else batch.next_window()
            finished = w is None
            done = torch.tensor([1 if w is None else 0], dtype=torch.int64, device=cdev)
            dist.all_reduce(done, op=dist.ReduceOp.MIN)      # ranks own different window counts
            if int(done.item()) == 1:
                break
            n = 0 if w is None else int(w.n_hits)
            torch.cuda.synchronize()
            local = torch.as_tensor(_DevArr(w.d_tuples, (n, 3), "<i4"), device="cuda") if n else torch.zeros((0, 3), dtype=torch.int32, device="cuda")
            if cdev == "cpu":
                local = local.cpu()
            shard.allgatherv_tuples(local, world, rank, dist, cdev)
            hits += n
            nwin += 1
        return hits, nwin

    rank_hits = []          # per-rank hit counts of the last step (N>1, --gather counts): the shard imbalance
    runs_replicated = False
    XKEYS = ("topics", "invalid_topics", "levels", "pairs", "hits", "visited_nodes", "overflow_topics", "walk_launches", "expand_launches", "walk_ms", "scan_ms",
             "expand_ms", "alg_bytes_walk", "alg_bytes_expand")
    xdelta = {}
    xstat = {"s": 0.0, "steps": 0, "runs": 0, "hits": 0, "my_runs": 0, "replicate_s": None, "replica_entries": None}      # the exchange step's own clock

    def step_gather_runs_torch():
        """--gather runs without a library communicator (gloo: several ranks share one GPU): the same exchange through torch.distributed.
        One expansion-free pass in the run-descriptor format; every window's descriptors {shard, src, len, topic} are all-gathered (counts,
        then one padded all_gather).  -> (my_runs, all_runs, hits described by all ranks' descriptors)"""
        nonlocal runs_replicated
        my_runs = all_runs = all_hits = 0
        batch.set_format(capi.RGR_FORMAT_RUNS)
        batch.begin()
        finished = False
        while True:
            w = None if finished else batch.next_window()
            finished = w is None
            done = torch.tensor([1 if w is None else 0], dtype=torch.int64, device=cdev)
            dist.all_reduce(done, op=dist.ReduceOp.MIN)      # ranks own different window counts
            if int(done.item()) == 1:
                break
            nr = 0 if w is None else int(w.n_runs)
            torch.cuda.synchronize()
            if w is not None and not runs_replicated:
                # once per epoch: every rank's subs[] on every rank (what rgr_comm_replicate_subs does over RCCL)
                t_r = time.time()
                ns = int(st0["n_subs"])
                mine = torch.as_tensor(_DevArr(w.d_subs, (ns, 2), "<i4"), device="cuda").cpu() if ns else torch.zeros((0, 2), dtype=torch.int32)
                _, cnts = shard.allgatherv_rows(mine, world, rank, dist, "cpu")
                xstat["replicate_s"], xstat["replica_entries"] = round(time.time() - t_r, 3), int(sum(cnts))
                runs_replicated = True
            if nr:
                src = torch.as_tensor(_DevArr(w.d_run_src, (nr,), "<i4"), device="cuda")
                rtp = torch.as_tensor(_DevArr(w.d_run_topic, (nr,), "<i4"), device="cuda")
                roff = torch.as_tensor(_DevArr(w.d_run_off, (nr + 1,), "<i8"), device="cuda")
                local = torch.stack([torch.full_like(src, rank), src, (roff[1:] - roff[:-1]).to(torch.int32), rtp], dim=1)
            else:
                local = torch.zeros((0, 4), dtype=torch.int32, device="cuda")
            if cdev == "cpu":
                local = local.cpu()
            allr, cnts = shard.allgatherv_rows(local, world, rank, dist, cdev)
            my_runs += nr
            all_runs += int(sum(cnts))
            all_hits += int(allr[:, 2].to(torch.int64).sum())
            del allr, local
        batch.set_format(capi.RGR_FORMAT_TUPLE)
        return my_runs, all_runs, all_hits

    # N>1: the exchange step runs inside the library over RCCL (rgr_comm_*: ncclAllGather of the counts,
    # all-gatherv of the tuples as a send/recv group) — torch.distributed only ships the 128-byte communicator
    # id.  Under gloo (several ranks sharing one GPU: logic tests) RCCL cannot form a communicator and the
    # torch collectives are used instead; `collective` in the output says which path ran.
    comm = None
    collective = "n/a"
    if world > 1:
        collective = "torch.distributed"
        if args.dist_backend == "nccl" and not args.torch_collectives:
            try:
                uid = torch.zeros(capi.RGR_COMM_ID_BYTES, dtype=torch.uint8, device=cdev)
                if rank == 0:
                    uid = torch.frombuffer(bytearray(capi.Comm.unique_id()), dtype=torch.uint8).to(cdev)
                dist.broadcast(uid, 0)
                # ncclCommInitRank is collective and blocks: run it on a helper thread so that a rendezvous that never
                # completes turns into the torch.distributed path instead of a hung bench (RCCL's first initialisation takes
                # about a minute on this image)
                import threading
                box = {}

                def _init():
                    try:
                        box["comm"] = capi.Comm(r, bytes(uid.cpu().numpy().tobytes()), rank, world)
                    except Exception as e:      # noqa: BLE001
                        box["err"] = e
                th = threading.Thread(target=_init, daemon=True)
                th.start()
                th.join(timeout=float(os.environ.get("RGR_COMM_INIT_TIMEOUT_S", "300")))
                if th.is_alive():
                    raise TimeoutError("rgr_comm_create did not return")
                if "err" in box:
                    raise box["err"]
                comm = box["comm"]
                collective = "rccl (rgr_comm_*, inside the library)"
            except Exception as e:
                log(f"library RCCL communicator unavailable ({e!r}): using torch.distributed collectives", rank)
                comm = None
            ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int64, device=cdev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and comm is not None:
                comm.close(); comm = None; collective = "torch.distributed"
        batch.set_topic_ids(keep_t.astype(np.uint32))          # tuples carry the GLOBAL publish index

    def step():
        if world > 1 and args.gather == "tuples":
            if comm is not None:
                mine, _, _ = comm.gather_pass(batch)
                rank_hits[:] = []
                return mine, 0
            return step_gather_tuples()
        hits, nwin = batch.run()      # synchronises the library's stream at the end of the pass
        if world > 1 and args.gather != "none":
            if comm is not None:
                rank_hits[:] = [int(x) for x in comm.allgather_u64(hits)]
            else:
                cnt = torch.tensor([hits], dtype=torch.int64, device=cdev)
                allc = [torch.zeros_like(cnt) for _ in range(world)]
                dist.all_gather(allc, cnt)
                rank_hits[:] = [int(x.item()) for x in allc]
            if args.gather == "runs":
                # BASELINE configs[3]'s "RCCL all-gatherv of subscriber hits" in the form that can run at this fan-out: the tuples stay
                # on the owning GPU (the pass above), and every rank additionally receives every rank's 16-byte run descriptors
                # (hits = the replicated subs[] of the owning rank, read in place): a second, expansion-free pass + the exchange
                nonlocal runs_replicated
                sx0 = r.stats()
                t_x = time.time()
                if comm is not None:
                    if not runs_replicated:
                        t_r = time.time()
                        comm.replicate_subs()
                        xstat["replicate_s"] = round(time.time() - t_r, 3)
                        runs_replicated = True
                    my_runs, all_runs, all_hits, _ = comm.gather_runs_pass(batch)
                else:
                    my_runs, all_runs, all_hits = step_gather_runs_torch()
                xstat["s"] += time.time() - t_x
                xstat["steps"] += 1
                sx1 = r.stats()         # the descriptor pass walks the batch a second time: its counters are the exchange's, not the timed pass's
                for k_ in XKEYS:
                    xdelta[k_] = xdelta.get(k_, 0) + sx1[k_] - sx0[k_]
                xstat["runs"], xstat["hits"], xstat["my_runs"] = int(all_runs), int(all_hits), int(my_runs)
        return hits, nwin

    for _ in range(warmup):
        step()
    r.stats_reset()
    xstat["s"], xstat["steps"] = 0.0, 0
    xdelta.clear()
    barrier()
    t_start = time.time()
    hits = nwin = 0
    for _ in range(steps):
        hits, nwin = step()
    barrier()
    elapsed = time.time() - t_start
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([hits, my_topics], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot)
        total_hits, total_topics = int(tot[0].item()), int(tot[1].item())
    else:
        total_hits, total_topics = hits, my_topics
    st = r.stats()
    exchange_kernel_ms = None
    if xdelta:
        exchange_kernel_ms = {"walk": round(xdelta["walk_ms"] / max(1, steps), 3), "scan_compact": round(xdelta["scan_ms"] / max(1, steps), 3)}
        for k_, v_ in xdelta.items():
            st[k_] -= v_
    comm_info = None
    if comm is not None:
        try:
            comm_info = comm.info()                 # ncclCommCount / ncclCommUserRank of the communicator the steps above ran on
        except Exception as e:      # noqa: BLE001   (reporting only)
            log(f"rgr_comm_info failed: {e!r}", rank)
    # ---- N > 1: per-topic digests of every rank's pass, assembled on rank 0 (collective), compared with the oracle below
    gathered = None
    if world > 1 and not args.no_parity:
        gathered = sharded_digests(r, batch, keep_t, n_pub, world, rank, dist, cdev, formats=not args.no_formats, qos=qos)
    if comm is not None:
        comm.close()
    if world > 1:
        # every rank leaves the process group TOGETHER, here; rank 0's tail (oracle build, exhaustive comparison, PMC children) is solo
        # work and must not leave peers waiting in a collective or a communicator half torn down
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        batch.close(); r.close()
        return None, None

    K = steps
    value = total_topics * K / elapsed
    exp_s, walk_s = st["expand_ms"] / 1e3, st["walk_ms"] / 1e3
    exp_gbs = st["alg_bytes_expand"] / exp_s / 1e9 if exp_s > 0 else 0.0
    walk_gbs = st["alg_bytes_walk"] / walk_s / 1e9 if walk_s > 0 else 0.0
    is_exp = exp_s >= walk_s
    exp_name = "expand_kernel"
    if deliver >= 0:        # the library's choice (kernels.hip launch_expand): the lean delivery expansion unless a switch says otherwise
        exp_name = ("expand_deliver_lean_kernel" if os.environ.get("RGR_DELIVER_LEAN", "1") != "0" else
                    "expand_deliver_early_kernel" if os.environ.get("RGR_DELIVER_EARLY", "1") != "0" else "expand_kernel<true>")
    dominant = exp_name if is_exp else ("retain_rounds" if retain else "walk_kernel")
    launches = st["expand_launches"] if is_exp else st["walk_launches"]
    dom_s = exp_s if is_exp else walk_s
    alg = exp_gbs if is_exp else walk_gbs
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "alg_achieved": round(alg, 1), "alg_frac": round(alg / HBM_PEAK_GBS, 4),
                "alg_note": "SURVEY 8(d) bytes (20 B/hit: 8 read + 12 written; 24 B/visited node + 8 B/matched filter + 4 B/level) over the same HIP-event "
                            "time; the 8 B/hit read term is served by L2/MALL for hot filters, so alg_frac may exceed frac and 1.0",
                "launches": int(launches), "avg_launch_ms": round(dom_s * 1e3 / max(1, launches), 4),
                "alg_bytes_per_launch": int((st["alg_bytes_expand"] if is_exp else st["alg_bytes_walk"]) / max(1, launches)),
                "hits_per_launch": int(st["hits"] / max(1, launches)) if is_exp else None,
                "topics_per_launch": int(st["topics"] / max(1, launches)),
                "walk_alg_GBps": round(walk_gbs, 1), "expand_alg_GBps": round(exp_gbs, 1)}

    rec = {
        "metric": f"publish-topic matches/sec with delivery stage (config {cfg}, scale {scale}, v5 fraction {deliver}, {'8-byte hits' if dfmt == capi.RGR_FORMAT_DELIVER8 else '12-byte tuples'})" if deliver >= 0 else
                  "publish-topic matches/sec @10M subs" if cfg in (3, 4) and scale == 1.0 else
                  (f"retained-path SUBSCRIBE-filter matches/sec (config 5, scale {scale})" if retain else
                   f"publish-topic matches/sec (config {cfg}, scale {scale})"),
        "value": round(value, 1), "unit": "SUBSCRIBE-filter matches/s" if retain else "publish-topic matches/s",
        "n_gpus": world, "steps": K, "warmup": warmup, "ms_per_step": round(elapsed * 1e3 / K, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": (f"BASELINE.json configs[{cfg - 1}]: {n_sub} retained topics (publish generator, distinct), {n_pub} wildcard SUBSCRIBE filters "
                                f"(config-3 filter generator, >=1 wildcard), seeds 0x{wl.PUB_SEED + cfg:X}/0x{wl.SUB_SEED + cfg:X}" if retain else
                                f"BASELINE.json configs[{cfg - 1}]: {n_sub} subscriptions (p_plus/level {c['p_plus']}, p_hash {c['p_hash']}, "
                                f"Zipf tokens s=1.1, Zipf clients s=1.0), {n_pub} publish topics, seeds 0x{wl.SUB_SEED + cfg:X}/0x{wl.PUB_SEED + cfg:X}"),
                   "subscriptions": n_sub, "publishes": n_pub, "sharding": f"hash of the first {shard.KEY_LEVELS} levels x{world}" if world > 1 else "none",
                   "topic_order": "walk order: the library sorts the batch by its leading level tokens (rgr_batch_set_order); tuples name the caller's index" if walk_order else "caller order",
                   "gather": args.gather if world > 1 else "n/a", "collective": collective,
                   "rccl_ranks": comm_info["ranks"] if comm_info and comm_info["transport"] == "rccl" else None,
                   "dist_backend": args.dist_backend if world > 1 else "n/a", "windows_per_step": int(nwin),
                   "library": capi.lib().rgr_version().decode()},                 # names the expansion kernels the formats run on
        "hits_per_step": int(total_hits), "hits_per_s": round(total_hits * K / elapsed, 1),
        "mean_hits_per_topic": round(total_hits / max(1, total_topics), 2),
        "mean_visited_nodes_per_topic": round(st["visited_nodes"] / max(1, st["topics"]), 2),
        "overflow_topics_per_step": int(st["overflow_topics"] / K),      # topics with more than slot_cap matched filters (re-walked into the arena)
        "kernel_ms_per_step": {"walk": round(st["walk_ms"] / K, 3), "scan_compact_tiles": round(st["scan_ms"] / K, 3),
                               "expand": round(st["expand_ms"] / K, 3)},
        "alg_bytes_per_step": {"walk": int(st["alg_bytes_walk"] / K), "expand": int(st["alg_bytes_expand"] / K)},
        "table": {"filters": int(st0["n_filters"]), "subs": int(st0["n_subs"]), "trie_nodes": int(st0["n_nodes"]),
                  "hbm_bytes": int(st0["table_bytes_device"]), "host_build_s": round(build_s, 1)},
        "roofline": roofline,
        # SURVEY 8(d) "end-to-end incl. H2D": the topics' host blob -> HBM + device tokeniser (rgr_batch_create, once per batch)
        # and one pass; the D2H side is pcie_inclusive_matches_per_s below.  `value` itself starts with the tokens in HBM.
        "h2d_inclusive": {"batch_create_ms": round(batch_create_s * 1e3, 2),
                          "matches_per_s": round(my_topics / (batch_create_s + elapsed / K), 1),
                          "what": "host topic strings -> rgr_batch_create (H2D + device tokeniser) -> one pass, tuples left in HBM"},
    }
    if world > 1:
        if args.gather == "runs" and xstat["steps"]:
            rec["exchange"] = {
                "form": "run descriptors: all-gatherv of {shard, src, len, topic} (16 B per (topic, matched filter with subscribers) run) per window; hits of a "
                        "descriptor = the owning rank's subs[src .. src+len), replicated on every rank once per epoch",
                "transport": collective, "ms_per_step": round(xstat["s"] * 1e3 / xstat["steps"], 3),
                "share_of_step": round(xstat["s"] / max(1e-9, elapsed), 3),
                "runs_all_ranks": xstat["runs"], "runs_this_rank": xstat["my_runs"], "hits_described": xstat["hits"],
                "describes_every_hit": bool(xstat["hits"] == total_hits),
                "bytes_received_per_rank_per_step": int(xstat["runs"] - xstat["my_runs"]) * 16, "bytes_all_ranks_per_step": int(xstat["runs"]) * 16 * (world - 1),
                "vs_tuple_bytes_all_ranks_per_step": int(total_hits) * 12 * (world - 1),
                "kernel_ms_per_step": exchange_kernel_ms,
                "subs_replication": {"seconds": xstat["replicate_s"], "entries": xstat["replica_entries"], "when": "once per epoch, outside the timed steps (first warmup step)"}}
        elif args.gather == "tuples":
            rec["exchange"] = {"form": "all-gatherv of the 12-byte tuples of every window", "transport": collective, "bytes_all_ranks_per_step": int(total_hits) * 12 * (world - 1)}
        elif args.gather == "counts":
            rec["exchange"] = {"form": "per-rank hit counts only (8 bytes per rank per step)", "transport": collective, "bytes_all_ranks_per_step": 8 * world * (world - 1)}
        rec["config"]["ranks"] = world
    try:
        if world > 1 and rank_hits and sum(rank_hits) > 0:
            rec["shard_hits"] = rank_hits
            rec["shard_imbalance_max_over_mean"] = round(max(rank_hits) * len(rank_hits) / sum(rank_hits), 3)
    except Exception as e:      # reporting only: never fail the bench line over it
        log(f"shard imbalance not reported: {e}", 0)
    if deliver >= 0:
        bph = 8 if dfmt == capi.RGR_FORMAT_DELIVER8 else 12
        rec["delivery_stage"] = {"v5_fraction": deliver, "format": DELIVERY_FORMAT_NAMES[dfmt], "bytes_written_per_hit": bph,
                                 "dedup_ms_per_step": round(st["dedup_ms"] / K, 3),
                                 "dedup_candidates_per_step": int(st["dedup_candidates"] / K),
                                 "dedup_launches_per_step": int(st["dedup_launches"] / K),
                                 "expand_store_GBps": round(total_hits * K * bph / max(1e-9, st["expand_ms"] / 1e3) / 1e9, 1),
                                 "frac_of_hbm_peak_stores_kernel": round(total_hits * K * bph / max(1e-9, st["expand_ms"] / 1e3) / 8.0e12, 3)}
        roofline["bytes_written_per_hit"] = bph
        if dfmt == capi.RGR_FORMAT_DELIVER8:
            # the same batch, same table, in the 12-byte tuple form and in the caller's topic order (what rounds 2-5 reported as the delivery record)
            if walk_order:
                batch.set_order(False)
            batch.set_format(capi.RGR_FORMAT_TUPLE)
            batch.run()
            r.stats_reset()
            torch.cuda.synchronize()
            t = time.time()
            for _ in range(steps):
                batch.run()
            dt = time.time() - t
            s12 = r.stats()
            rec["delivery_stage"]["tuple12"] = {"value": round(my_topics * steps / dt, 1), "unit": rec["unit"], "ms_per_step": round(dt * 1e3 / steps, 3),
                                                "expand_avg_launch_ms": round(s12["expand_ms"] / max(1, s12["expand_launches"]), 4),
                                                "dedup_ms_per_step": round(s12["dedup_ms"] / steps, 3),
                                                "expand_store_GBps": round(total_hits * steps * 12 / max(1e-9, s12["expand_ms"] / 1e3) / 1e9, 1), "topic_order": "caller"}
            batch.set_format(dfmt)
            if walk_order:
                batch.set_order(True)
        if not args.no_parity:
            rec["parity_sample"] = delivery_parity(batch, W, pa, fmt=dfmt)
            log(f"config {cfg}: delivery parity {rec['parity_sample']}", 0)
            if dfmt == capi.RGR_FORMAT_DELIVER8:
                if walk_order:
                    batch.set_order(False)
                p12 = delivery_parity(batch, W, pa, fmt=capi.RGR_FORMAT_TUPLE)
                rec["parity_sample"]["tuple12"] = _pick(p12, ["ok", "windows_checked", "of_windows", "hits", "mismatching_words"])
                rec["parity_sample"]["ok"] = bool(rec["parity_sample"]["ok"] and p12["ok"])
                batch.set_format(dfmt)
                if walk_order:
                    batch.set_order(True)
        if args.cpu_sample != 0 and world == 1:
            # the oracle's side of the delivery record: its own delivery verdicts for a stratified sample of publishes (digests of the
            # per-hit words in canonical order, DefaultRouter::deliver_digest) against the device's, and the reference-shaped
            # _matches + collector + forwards_to pass as the CPU baseline
            from oracle import oracle as orc
            cores = args.cpu_threads or os.cpu_count() or 1
            o = _ORACLES.pop((cfg, scale, deliver), None)
            if o is None:
                t = time.time()
                o = orc.DefaultRouter()
                o.add_bulk_ex(blob, offs, client, qos, W["deliver_flags"])
                log(f"config {cfg}: oracle table with v5 flags built in {time.time() - t:.1f}s", 0)
            if not args.no_parity:
                rec["parity_sample"]["oracle"] = delivery_oracle_sample(o, W, batch, pa, cores, fmt=dfmt)
                rec["parity_sample"]["ok"] = bool(rec["parity_sample"]["ok"] and rec["parity_sample"]["oracle"]["ok"])
                log(f"config {cfg}: delivery oracle sample {rec['parity_sample']['oracle']}", 0)
            hpt = max(1.0, total_hits / max(1, total_topics))
            n_s = int(min(n_pub, max(500, 4.0e8 * cores / 256 / hpt)))
            idx = np.sort(np.random.default_rng(20260922).choice(n_pub, size=n_s, replace=False))
            sb, so = shard.take(tb, to, idx)
            sec, ost = o.forwards_timed(sb, so, pa["from_id"][idx], pa["qos_retain"][idx].astype(np.uint8), cores)
            rec["cpu_baseline"] = {"value": round(n_s / sec, 1), "unit": rec["unit"], "cores": cores, "kind": "port",
                                   "what": "DefaultRouter::_matches with the v3 / v5 collector (router.rs:174-265, types.rs:510-540: No Local, first hit per v5 client) + "
                                           "forwards_to's per-recipient qos / retain transform (shared.rs:886-908); ref-counted clones, no canonicalising sort",
                                   "sample": f"{n_s} publishes drawn at random (seeded) from the same batch, {ost['hits']} relations visited, {ost['rows']} rows delivered, {sec:.2f}s wall",
                                   "hits_per_s": round(ost["hits"] / sec, 1)}
            del o

    hits_per_topic = max(1.0, total_hits / max(1, total_topics))
    # ---- opt-in compact result formats (SURVEY 8(b)'s SoA result; rgr_batch_set_format), reported BESIDE the 12-byte
    # tuple headline, never instead of it: same hits, same order, topic implied by the CSR offsets
    if world == 1 and deliver < 0 and not args.no_formats:
        rec["compact_formats"] = []
        for name, fmt, bph in (("soa: sub_id u32[] + qos u8[]", capi.RGR_FORMAT_SOA, 5), ("packed: sub_id | qos << 30 u32[]", capi.RGR_FORMAT_PACKED, 4),
                               ("ids24: sub_id as 3 little-endian bytes u8[3n] (the qos is table data, indexed by sub id)", capi.RGR_FORMAT_IDS24, 3),
                               ("runs: (topic, subscriber-run) descriptors, hits read in place from the epoch's subs[]", capi.RGR_FORMAT_RUNS, 0)):
            try:
                batch.set_format(fmt)
                batch.run()
                r.stats_reset()
                torch.cuda.synchronize()
                t = time.time()
                for _ in range(steps):
                    h2, _ = batch.run()
                dt = time.time() - t
                s2 = r.stats()
                rec["compact_formats"].append({
                    "format": name, "bytes_written_per_hit": bph, "value": round(my_topics * steps / dt, 1), "unit": rec["unit"],
                    "ms_per_step": round(dt * 1e3 / steps, 3), "hits_per_s": round(h2 * steps / dt, 1),
                    "expand_avg_launch_ms": round(s2["expand_ms"] / max(1, s2["expand_launches"]), 4),
                    "expand_store_GBps": round(h2 * steps * bph / max(1e-9, s2["expand_ms"] / 1e3) / 1e9, 1),
                    # the two halves of north_star in one record: matches/s (value) and the share of the 8 TB/s HBM peak the format's
                    # own stores reach — inside its expansion kernel, and over the whole pass (preparation, launches and gaps included)
                    "frac_of_hbm_peak_stores_kernel": round(h2 * steps * bph / max(1e-9, s2["expand_ms"] / 1e3) / 8.0e12, 3),
                    "frac_of_hbm_peak_stores_whole_pass": round(h2 * steps * bph / dt / 8.0e12, 3),
                    "speedup_vs_tuple": round((my_topics * steps / dt) / value, 3)})
            except capi.RgrError as e:
                rec["compact_formats"].append({"format": name, "error": str(e)})
        batch.set_format(capi.RGR_FORMAT_TUPLE)
        if retain:
            # the retained path answering with POSITIONS of the epoch's preorder value array (rgr_batch_set_retain_positions): the same 12-byte
            # tuples, nothing read per hit — beside the headline, whose tuples carry the caller's topic ids
            batch.set_retain_positions(True)
            batch.run()
            r.stats_reset()
            torch.cuda.synchronize()
            t = time.time()
            for _ in range(steps):
                h2, _ = batch.run()
            dt = time.time() - t
            s2 = r.stats()
            batch.set_retain_positions(False)
            rec["retain_positions"] = {"value": round(my_topics * steps / dt, 1), "unit": rec["unit"], "ms_per_step": round(dt * 1e3 / steps, 3),
                                       "expand_avg_launch_ms": round(s2["expand_ms"] / max(1, s2["expand_launches"]), 4),
                                       "expand_store_GBps": round(h2 * steps * 12 / max(1e-9, s2["expand_ms"] / 1e3) / 1e9, 1),
                                       "frac_of_hbm_peak_stores_kernel": round(h2 * steps * 12 / max(1e-9, s2["expand_ms"] / 1e3) / 8.0e12, 3),
                                       "speedup_vs_topic_ids": round((my_topics * steps / dt) / value, 3),
                                       "what": "rgr_batch_set_retain_positions: rgr_tuple.sub_id = position in the epoch's preorder value array (topic id = "
                                               "mirror[position], rgr_batch_retain_vals); a trailing '#' is a contiguous range of it, so the expansion reads nothing per hit"}
    # ---- PCIe-inclusive rate: the same pass with every window copied into pinned host memory (bounded prefix:
    # at config-3 fan-out a publish carries 178 KB of tuples, the full batch would be 1.8 TB over the link)
    if world == 1 and not args.no_d2h and deliver < 0:
        n_d = int(min(n_pub, max(1000, 6.0e9 / hits_per_topic)))        # ~72 GB of tuples at most
        db, do = prefix(W, n_d) if n_d < n_pub else (tb, to)
        b2 = r.retain_batch(db, do) if retain else r.batch(db, do)
        b2.run_to_host()                           # warm the pinned staging ring
        t = time.time()
        h2, _ = b2.run_to_host()
        dt = time.time() - t
        b2.close()
        rec["pcie_inclusive_matches_per_s"] = round(n_d / dt, 1)
        rec["pcie_inclusive"] = {"sample": f"first {n_d} topics of the batch, every window streamed to pinned host memory (rgr_batch_run_to_host)",
                                 "tuple_GBps": round(h2 * 12 / dt / 1e9, 2), "seconds": round(dt, 3)}
        if retain:
            # the retained path's dense answer (rgr_retain_match_ranges): per filter the ranges of the host-mirrored preorder value array —
            # host strings in, host-resident answer out, for the WHOLE batch (the tuple form above can only afford a prefix)
            r.retain_match_ranges(*prefix(W, min(n_pub, 1000)), flatten=False)
            t = time.time()
            rg = r.retain_match_ranges(tb, to, flatten=False)
            dt = time.time() - t
            rec["pcie_inclusive_ranges"] = {"matches_per_s": round(n_pub / dt, 1), "seconds": round(dt, 3), "filters": int(n_pub), "ranges": rg["n_ranges"],
                                            "entries_described": rg["n_entries"], "equals_hits_of_the_timed_pass": bool(rg["n_entries"] == total_hits),
                                            "bytes_over_pcie": rg["n_ranges"] * 16, "vs_tuple_bytes": int(rg["n_entries"]) * 12,
                                            "speedup_vs_tuples_to_host": round((n_pub / dt) / max(1e-9, rec["pcie_inclusive_matches_per_s"]), 1),
                                            "what": "rgr_retain_match_ranges over the full batch: filter strings from host memory -> per filter the ranges of "
                                                    "the value array the library mirrors on the host (16 B per range over PCIe; `a/#` is one range)"}

    # ---- CPU baseline (reference-shaped port) + parity sample against the oracle on the full table (N=1 only)
    if world > 1 and deliver < 0 and (gathered is not None or args.cpu_sample != 0):
        # N > 1: rank 0's oracle holds the UNSHARDED table; every topic of the batch is compared, and the CPU baseline is the same leg as at N = 1
        from oracle import oracle as orc
        cores = args.cpu_threads or os.cpu_count() or 1
        t = time.time()
        o = orc.DefaultRouter()
        o.add_bulk(blob, offs, client, qos)
        log(f"config {cfg}: oracle table (unsharded) built in {time.time() - t:.1f}s", 0)
        if gathered is not None:
            got, gst, fmt_all, dinfo, extra = gathered
            rec["parity_sample"] = compare_with_oracle(o, W, got, gst, fmt_all, dinfo, cores, primary, extra=extra)
            log(f"config {cfg}: parity_sample {rec['parity_sample']}", 0)
        if args.cpu_sample != 0:
            rec["cpu_baseline"] = cpu_baseline_leg(o, W, args, cores, primary, hits_per_topic, rec["unit"])
            rec["cpu_baseline"]["where"] = "rank 0's host, after every rank left the process group (solo work): full unsharded table, whole-batch sample"
        else:
            rec["cpu_baseline"] = None
        del o, gathered
    elif args.cpu_sample != 0 and world == 1 and deliver < 0:
        from oracle import oracle as orc
        cores = args.cpu_threads or os.cpu_count() or 1
        t = time.time()
        if retain:
            o = orc.RetainTree()
            o.insert_bulk(blob, offs)
        else:
            o = orc.DefaultRouter()
            keep = primary and cfg == 3 and scale == 1.0 and not args.no_secondary
            if keep:        # the delivery-stage secondary of this run reuses this table: same subscriptions, its v5 flags ride along (they
                            # change nothing here: the digests read rel ids and qos, the timed pass's publisher matches no subscriber Id)
                o.add_bulk_ex(blob, offs, client, qos, deliver_flags(n_sub, DELIVER_SECONDARY_V5))
            else:
                o.add_bulk(blob, offs, client, qos)
        log(f"config {cfg}: oracle table built in {time.time() - t:.1f}s; cpu_baseline on {cores} threads", 0)
        rec["cpu_baseline"] = cpu_baseline_leg(o, W, args, cores, primary, hits_per_topic, rec["unit"])
        if not args.no_parity:
            rec["parity_sample"] = parity_sample(r, o, W, batch, cores, primary)
            log(f"config {cfg}: parity_sample {rec['parity_sample']}", 0)
        if not retain and primary and cfg == 3 and scale == 1.0 and not args.no_secondary:
            _ORACLES[(cfg, scale, DELIVER_SECONDARY_V5)] = o
        del o
    else:
        rec.setdefault("cpu_baseline", None)      # (the delivery record set its own above)

    phase = {"st": st, "dominant": dominant, "retain": retain}
    batch.close(); r.close()
    return rec, phase


def attach_traffic(rec, phase, pmc, cal):
    """roofline.frac from the PMC children's counters of the dominant kernel, per launch like alg."""
    rf = rec["roofline"]
    if not pmc:
        rf["traffic_note"] = "PMC passes unavailable in this run: frac not computed (alg_frac is the SURVEY 8(d) figure)"
        return
    cls = "expand" if rf["kernel"].startswith("expand_") else ("retain" if phase["retain"] else "walk")
    d = pmc.get(cls)
    if not d or "fetch_KiB" not in d or "write_KiB" not in d or not d["dispatches"]:
        rf["traffic_note"] = f"no PMC rows for {rf['kernel']}"
        return
    fs = float(cal.get(cls, {}).get("fetch_scale", 1.0))
    ws = float(cal.get(cls, {}).get("write_scale", 1.0))
    unit = d["hits"] if cls == "expand" else d["topics"]
    f_b, w_b = d["fetch_KiB"] * 1024 * fs, d["write_KiB"] * 1024 * ws
    per_unit = (f_b + w_b) / max(1, unit)
    units_per_launch = rf["hits_per_launch"] if cls == "expand" else rf["topics_per_launch"]
    traffic = per_unit * units_per_launch
    ach = traffic / (rf["avg_launch_ms"] / 1e3) / 1e9 if rf["avg_launch_ms"] > 0 else 0.0
    rf.update({"traffic": int(traffic), "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
               "traffic_source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over {d['dispatches']} dispatches of this run's "
                                 f"--pmc-child replay ({unit} {'hits' if cls == 'expand' else 'queries'})",
               "traffic_per_unit_B": {"fetch_raw": round(d["fetch_KiB"] * 1024 / max(1, unit), 3), "fetch_scale": fs,
                                      "write_raw": round(d["write_KiB"] * 1024 / max(1, unit), 3), "write_scale": ws,
                                      "unit": "hit" if cls == "expand" else "query"},
               "avg_launch_us_under_pmc": d.get("avg_us_under_pmc"),
               # the counters sit on the fabric side of L2: reads served by the 256 MiB Infinity Cache are counted too, so `frac`
               # is an upper bound on HBM traffic when the read set fits it (config 5's 40 MB value array does); stores always reach HBM
               "frac_stores_only": round(w_b / max(1, unit) * units_per_launch / (rf["avg_launch_ms"] / 1e3) / 1e9 / HBM_PEAK_GBS, 4) if rf["avg_launch_ms"] > 0 else None,
               "frac_note": "frac = HBM bytes the PMC counters saw (traffic) / HIP-event time / 8 TB/s — the physical figure.  alg_frac = SURVEY 8(d)'s "
                            "algorithmic bytes (20 B/hit) over the same time; it exceeds frac, and 1, on the expansion because its 8 B/hit READ term never "
                            "reaches HBM: 64 filters produce 98 % of config 3's hits, their subscriber runs stay in L2 / Infinity Cache (the counters see "
                            "0.25 B fetched per hit).  frac_stores_only = the 12 B/hit of tuple stores alone, the floor no cache can remove."})
    if phase["retain"] and cls == "expand" and rf.get("frac_stores_only") is not None:
        # the retained path's value array (40 MB at BASELINE configs[4]) is served by the Infinity Cache: FETCH_SIZE counts those reads although
        # they never reach HBM (MI355X guide, HBM section), and `traffic`-based frac came out above the ~6.3 TB/s HBM can sustain (round 5: 0.905).
        # The physical HBM figure of this kernel is its stores: frac = frac_stores_only, the counter-inclusive one is kept beside it.
        rf["frac_counters_incl_infinity_cache_reads"] = rf["frac"]
        rf["achieved_counters_incl_infinity_cache_reads"] = rf["achieved"]
        rf["frac"] = rf["frac_stores_only"]
        rf["achieved"] = round(rf["frac_stores_only"] * HBM_PEAK_GBS, 1)
        rf["frac_note"] = ("frac = the tuple STORES alone (WRITE_SIZE) / HIP-event time / 8 TB/s: the reads of this kernel (value entries, 4 B per hit from the "
                           "packed side array) are served by the 256 MiB Infinity Cache, which the fabric-side FETCH_SIZE counter includes; "
                           "frac_counters_incl_infinity_cache_reads is that figure")
    gc = cal.get("walk", {}).get("gather_ceiling_Ggathers_per_s")
    if cls == "walk" and gc and rf["avg_launch_ms"] > 0:
        # the walk's own yardstick: dependent random 32-byte gathers.  tools/membench.hip `calib` measures how many such
        # gathers per second this chip sustains from a table far larger than its caches; upper trie levels are cache hits,
        # so the walk can exceed it.
        nodes_per_s = rec["mean_visited_nodes_per_topic"] * rf["topics_per_launch"] / (rf["avg_launch_ms"] / 1e3)
        rf["random_gather"] = {"visited_nodes_per_s": round(nodes_per_s, 1), "ceiling_gathers_per_s": gc * 1e9,
                               "frac_of_ceiling": round(nodes_per_s / (gc * 1e9), 3),
                               "note": "ceiling = random 32-byte gathers/s from a 16 GiB table (tools/membench.hip calib, profiles/pmc_calibration.json)"}
    other = "walk" if cls == "expand" else "expand"
    o = pmc.get(other)
    if o and "fetch_KiB" in o and "write_KiB" in o and o["dispatches"]:
        ou = o["hits"] if other == "expand" else o["topics"]
        rf[f"{other}_traffic_per_unit_B"] = {"fetch_raw": round(o["fetch_KiB"] * 1024 / max(1, ou), 3), "write_raw": round(o["write_KiB"] * 1024 / max(1, ou), 3),
                                             "fetch_scale": float(cal.get(other, {}).get("fetch_scale", 1.0)),
                                             "unit": "hit" if other == "expand" else "query"}


def measure_group(args):
    """BASELINE configs[3] on the hardware at hand: the 10 M-subscription table hash-sharded over --group shards inside ONE
    process (rgr_group_*), all shards on GPU 0 when the box has a single device.  Not a scaling measurement — every shard
    shares one GPU — but the sharded product path at full size: placement, per-shard hit counts all-gathered through the
    group's communicators, the total checked against the unsharded table, and (--gather tuples) the all-gatherv pass."""
    import torch
    from rmqtt_amd import capi
    W = gen_workload(args.config, args.scale)
    ndev = torch.cuda.device_count()
    devices = [i % ndev for i in range(args.group)]
    g = capi.Group(devices, window_hits=args.window_hits)
    if args.key_levels:
        g.set_key_levels(args.key_levels)
    t = time.time()
    rej = g.subscribe_bulk(W["blob"], W["offs"], None, W["qos"])
    g.commit()
    build_s = time.time() - t
    per = [g.shard_stats(s) for s in range(args.group)]
    log(f"group: {args.group} shards on devices {devices} (rccl: {g.uses_rccl()}), built in {build_s:.1f}s; subs per shard {[p['n_subs'] for p in per]}")
    gb = g.batch(W["tb"], W["to"])
    for _ in range(args.warmup):
        gb.run()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(args.steps):
        sh, tot = gb.run()
    dt = time.time() - t
    rec = {"metric": f"publish-topic matches/sec @10M subs, table hash-sharded x{args.group} in one process (rgr_group) on {len(set(devices))} GPU(s)",
           "value": round(W["n_pub"] * args.steps / dt, 1), "unit": "publish-topic matches/s", "n_gpus": len(set(devices)), "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(dt * 1e3 / args.steps, 3), "higher_is_better": True, "scaling": "n/a (shards share the device)",
           "vs_baseline": None, "dtype": "u32", "data": "synthetic",
           "config": {"workload": f"BASELINE.json configs[3] layout: {W['n_sub']} subscriptions sharded by the first {args.key_levels or 3} topic level(s) over {args.group} shards, {W['n_pub']} publishes",
                      "key_levels": args.key_levels or 3,
                      "shards": args.group, "devices": devices, "transport": "rccl" if g.uses_rccl() else "device copies (shards share a GPU)"},
           "hits_per_step": int(tot), "shard_hits": [int(x) for x in sh], "shard_imbalance_max_over_mean": round(float(sh.max()) * args.group / max(1, int(sh.sum())), 3),
           "shard_subs": [int(p["n_subs"]) for p in per], "replicated_subs": int(sum(p["n_subs"] for p in per) - (W["n_sub"] - rej)),
           "table_hbm_bytes_per_shard": [int(p["table_bytes_device"]) for p in per]}
    if args.gather == "tuples":
        t = time.time()
        tot2, _ = gb.gather(0)
        rec["allgatherv_pass_s"] = round(time.time() - t, 3)
        rec["allgatherv_total_hits"] = int(tot2)
    if args.gather == "runs":
        # the exchange step that can run at full fan-out: 16-byte run descriptors, subs[] replicated on every shard
        gb.gather_runs(0)                           # (first call replicates the subscriber entries)
        t = time.time()
        n_runs, n_hits, _ = gb.gather_runs(0)
        dt2 = time.time() - t
        rec["run_gather"] = {"pass_s": round(dt2, 3), "matches_per_s": round(W["n_pub"] / dt2, 1), "runs": int(n_runs), "hits_described": int(n_hits),
                             "bytes_gathered_per_shard": int(n_runs) * 16, "vs_tuples_bytes": int(n_hits) * 12,
                             "what": "rgr_group_batch_gather_runs: every shard receives every shard's run descriptors (16 B per (topic, filter) run); "
                                     "hits are read in place from the replicated subs[]"}
    print(json.dumps(rec), flush=True)
    gb.close(); g.close()
    return 0


def measure_router_e2e(args, quick=False):
    """What a broker sees through the drop-in boundary: N host threads call Router::matches (one publish per call, as
    DefaultShared::forwards does, shared.rs:772) on the C++ twin of the GpuRouter through its deadline micro-batcher; beside it
    the oracle's DefaultRouter::_matches-shaped pass on the same number of threads of the same host.  Both build the full
    SubRelationsMap per publish.  One JSON line per config (2 and 3) with publishes/s and per-call latency."""
    import ctypes as C

    from oracle import oracle as orc
    from rmqtt_amd import build
    build.build_gpu()
    L = C.CDLL(build.build_host_router())
    vp = C.c_void_p
    L.hr_new.restype = vp; L.hr_new.argtypes = [C.c_uint64, C.c_int]
    L.hr_free.argtypes = [vp]
    L.hr_set_match_mode.argtypes = [vp, C.c_int]; L.hr_set_match_mode.restype = None
    L.hr_stale_expansions.argtypes = [vp]; L.hr_stale_expansions.restype = C.c_uint64
    L.hr_restore_bulk.argtypes = [vp, vp, vp, vp, vp, C.c_uint64]
    L.hr_e2e_run.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, vp, vp, vp, C.c_uint32, vp]
    L.hr_e2e_run_async.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, vp, vp, vp, C.c_uint32, vp]
    L.hr_restore_bulk_ex.argtypes = [vp, vp, vp, vp, vp, vp, C.c_uint64]
    L.hr_forwards_run_async.argtypes = [vp, vp, vp, C.c_uint32, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, vp, vp, vp, C.c_uint32, vp]
    legs = set(args.e2e_legs.split(","))
    cores = args.cpu_threads or os.cpu_count() or 1
    out = []
    for cfg in [int(x) for x in args.e2e_configs.split(",")]:
        W = gen_workload(cfg, args.scale)
        n_t = int(min(W["n_pub"], 200_000))
        tb, to = prefix(W, n_t)
        tb = np.ascontiguousarray(tb, dtype=np.uint8); to = np.ascontiguousarray(to, dtype=np.uint64)
        blob = np.ascontiguousarray(W["blob"], dtype=np.uint8); offs = np.ascontiguousarray(W["offs"], dtype=np.uint64)
        client = np.ascontiguousarray(W["client"], dtype=np.uint32); qos = np.ascontiguousarray(W["qos"], dtype=np.uint8)
        if "forwards" in legs:
            # ---- Shared::forwards (shared.rs:735-820, 876-963) through GpuShared: one delivery pass per batch, delivery words -> sessions, no SubRelationsMap;
            # beside it the oracle's _matches + collector + forwards_to pass on the same publishes (same table: 10 % MQTT v5 relations, same publishers)
            frec = measure_forwards_e2e(args, L, W, cfg, n_t, tb, to, blob, offs, client, qos, cores, quick)
            out.append(frec)
            if not quick:
                print(json.dumps(frec), flush=True)
        if "matches" not in legs:
            continue
        g = L.hr_new(1, 0)
        t = time.time()
        assert L.hr_restore_bulk(g, blob.ctypes.data, offs.ctypes.data, client.ctypes.data, qos.ctypes.data, W["n_sub"]) == 0
        log(f"router e2e config {cfg}: GpuRouter::restore of {W['n_sub']} relations in {time.time() - t:.1f}s")
        rec = {"metric": f"Router::matches publishes/sec through the drop-in boundary (config {cfg})", "unit": "publishes/s", "threads": cores,
               "config": {"workload": f"BASELINE.json configs[{cfg - 1}]: {W['n_sub']} subscriptions; {n_t} publish topics cycled, one publish per trait call",
                          "batcher": {"max_batch": 4096, "max_delay_us": 200}}, "gpu": []}
        for mode, name in ((1, "filters: rgr_group_match_filter_subs + host expansion in the callers"), (2, "deliver: 12-byte tuples with delivery words, grouped by node on the device")):
            L.hr_set_match_mode(g, mode)
            res = (C.c_uint64 * 3)()
            wall = C.c_double(0)
            lat = np.zeros(200_000, dtype=np.float32)
            nl = C.c_uint32(0)
            if quick and mode == 2:
                continue
            secs = 2.5 if quick else (6.0 if mode == 1 else 4.0)
            L.hr_e2e_run(g, tb.ctypes.data, to.ctypes.data, n_t, cores, 4096, 200, 1.0, res, C.byref(wall), None, 0, None)      # warm
            L.hr_e2e_run(g, tb.ctypes.data, to.ctypes.data, n_t, cores, 4096, 200, secs, res, C.byref(wall), lat.ctypes.data, len(lat), C.byref(nl))
            l = np.sort(lat[:nl.value])
            rec["gpu"].append({"mode": name, "value": round(res[0] / wall.value, 1), "rows_per_s": round(res[1] / wall.value, 1), "device_passes": int(res[2]),
                               "publishes": int(res[0]), "wall_s": round(wall.value, 2),
                               "latency_us": {"p50": round(float(l[len(l) // 2]), 1), "p99": round(float(l[int(len(l) * 0.99)]), 1)} if len(l) else None})
            log(f"router e2e config {cfg}: {rec['gpu'][-1]}")
        # ---- the boundary at its design point: publishes submitted asynchronously (what tokio's task-level concurrency gives the
        # reference's callers, shared.rs:772): a few submitter threads keep `outstanding` publishes in flight, up to `passes` device passes
        # overlap, the completions (each publish's SubRelationsMap) are built on `workers` pool threads
        rec["gpu_async"] = []
        L.hr_set_match_mode(g, 1)
        shapes = [(args.e2e_submitters, args.e2e_outstanding, args.e2e_workers or max(8, min(64, cores // 4)), args.e2e_passes)]
        if args.e2e_sweep:
            shapes += [(4, 8192, 32, 2), (8, 32768, 64, 3), (16, 65536, 96, 4), (8, 16384, 32, 1)]
        for subm, outst, workers, passes in shapes:
            res = (C.c_uint64 * 11)()
            wall = C.c_double(0)
            lat = np.zeros(400_000, dtype=np.float32)
            nl = C.c_uint32(0)
            L.hr_e2e_run_async(g, tb.ctypes.data, to.ctypes.data, n_t, subm, outst, workers, passes, 4096, 200, 1.0, res, C.byref(wall), None, 0, None)      # warm
            churn = start_churn(L, g) if getattr(args, "e2e_churn", False) else None
            L.hr_e2e_run_async(g, tb.ctypes.data, to.ctypes.data, n_t, subm, outst, workers, passes, 4096, 200, 4.0 if quick else (5.0 if cfg == 2 else 4.0), res, C.byref(wall),
                               lat.ctypes.data, len(lat), C.byref(nl))
            churn_rec = stop_churn(churn)
            l = np.sort(lat[:nl.value])
            rec["gpu_async"].append({"submitters": subm, "outstanding": outst, "workers": workers, "passes_in_flight": passes,
                                     "value": round(res[0] / wall.value, 1), "rows_per_s": round(res[1] / wall.value, 1), "device_passes": int(res[2]),
                                     "publishes_per_pass": round(res[0] / max(1, res[2]), 1), "errors": int(res[3]), "wall_s": round(wall.value, 2),
                                     "batcher_ms_per_pass": {"collect": round(res[4] / max(1, res[2]) / 1e6, 3), "device_pass": round(res[5] / max(1, res[2]) / 1e6, 3),
                                                             "dispatch": round(res[6] / max(1, res[2]) / 1e6, 3)},
                                     "worker_task_us": round(res[7] / max(1, res[8]) / 1e3, 1), "worker_tasks": int(res[8]), "max_task_queue": int(res[9]), "requeued_publishes": int(res[10]),
                                     "latency_us": {"p50": round(float(l[len(l) // 2]), 1), "p99": round(float(l[int(len(l) * 0.99)]), 1)} if len(l) else None})
            if churn_rec:
                rec["gpu_async"][-1]["subscribe_churn_under_load"] = churn_rec
            log(f"router e2e config {cfg} async: {rec['gpu_async'][-1]}")
        L.hr_free(g)
        o = orc.DefaultRouter()
        o.add_bulk(W["blob"], W["offs"], W["client"], W["qos"])
        n_c = n_t if cfg == 2 else int(min(n_t, 80_000 * cores / 256 + 2000))
        cb, co = prefix(W, n_c)
        # (at config 2 one sweep over the sample takes tens of milliseconds on 256 threads — thread start-up dominates and single sweeps
        # read anywhere between 2.3 and 9 M/s — so the sample is swept until ~2 s have gone by and the total is what counts)
        o.matches_timed(cb, co, cores)
        sec, ost, reps = 0.0, {"hits": 0}, 0
        while sec < 2.0 and reps < 400:
            s1, o1 = o.matches_timed(cb, co, cores)
            sec += s1; ost["hits"] += o1["hits"]; reps += 1
        sec1, ost1 = o.matches_timed(*prefix(W, max(50, n_c // cores * 2)), 1)
        rec["cpu_reference_port"] = {"value": round(n_c * reps / sec, 1), "rows_per_s": round(ost["hits"] / sec, 1), "threads": cores, "kind": "port",
                                     "what": "oracle DefaultRouter::_matches-shaped pass (router.rs:174-265), per-hit ref-counted clones", "sample": n_c, "sweeps": reps,
                                     "single_thread": round(max(50, n_c // cores * 2) / sec1, 1)}
        best = max(x["value"] for x in rec["gpu"] + rec["gpu_async"])
        rec["value"] = best
        rec["value_blocking_callers"] = max(x["value"] for x in rec["gpu"])
        rec["value_async_submit"] = max(x["value"] for x in rec["gpu_async"])
        rec["vs_cpu_port"] = round(best / rec["cpu_reference_port"]["value"], 2)
        del o
        out.append(rec)
        if not quick:
            print(json.dumps(rec), flush=True)
    return out if quick else 0


def start_churn(L, g):
    """A subscriber thread beside the publishes of an e2e leg (--e2e-churn): one relation added and removed over and over (Router::add / remove take the
    table's lock exclusively, the passes and the completions hold it shared) — how long a subscribe waits under full publish load, and what it does to it."""
    import ctypes as C
    import threading

    class HrId(C.Structure):
        _fields_ = [("node_id", C.c_uint64), ("client_id", C.c_char_p), ("client_len", C.c_uint32), ("create_time", C.c_int64), ("lid", C.c_uint16)]

    class HrOpts(C.Structure):
        _fields_ = [("v5", C.c_uint8), ("qos", C.c_uint8), ("no_local", C.c_uint8), ("rap", C.c_uint8), ("rh", C.c_uint8), ("sub_ident", C.c_uint32),
                    ("shared_group", C.c_char_p), ("shared_group_len", C.c_uint32)]
    L.hr_add.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.POINTER(HrId), C.POINTER(HrOpts)]
    L.hr_remove.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.POINTER(HrId)]
    stop, lat_add, lat_rm = threading.Event(), [], []

    def churner():
        cid = b"churner"
        hid, ho, f = HrId(1, cid, len(cid), 0, 0), HrOpts(1, 1, 0, 0, 0, 0, None, 0), b"churn/+/x"
        while not stop.is_set():
            t0 = time.perf_counter(); L.hr_add(g, f, len(f), C.byref(hid), C.byref(ho)); lat_add.append(time.perf_counter() - t0)
            t0 = time.perf_counter(); L.hr_remove(g, f, len(f), C.byref(hid)); lat_rm.append(time.perf_counter() - t0)
            time.sleep(0.002)
    th = threading.Thread(target=churner)
    th.start()
    return th, stop, lat_add, lat_rm


def stop_churn(churn):
    if churn is None:
        return None
    th, stop, lat_add, lat_rm = churn
    stop.set(); th.join()
    q = lambda a, f: round(float(np.quantile(np.asarray(a), f)) * 1e3, 3) if a else None
    return {"adds": len(lat_add), "removes": len(lat_rm), "add_ms": {"p50": q(lat_add, 0.5), "p99": q(lat_add, 0.99), "max": q(lat_add, 1.0)},
            "remove_ms": {"p50": q(lat_rm, 0.5), "p99": q(lat_rm, 0.99), "max": q(lat_rm, 1.0)},
            "what": "one relation added and removed in a loop (2 ms apart) by another thread during the timed run: Router::add / remove wait for the table's exclusive lock"}


def measure_forwards_e2e(args, L, W, cfg, n_t, tb, to, blob, offs, client, qos, cores, quick):
    """Shared::forwards end to end (the consumer of SURVEY 8(f)-1's delivery stage): publishes submitted asynchronously to the C++ twin of GpuShared
    (rmqtt_amd/host/gpu_shared.*), every batch ONE device pass (rgr_group_match_batch_deliver with the publishes' qos / retain), every publish consumed
    from delivery words on a pool thread — per recipient a Publish clone with qos' / retain' handed to the session's (counting) channel.  CPU side: the
    oracle's DefaultRouter::forwards_shaped (_matches + v3 / v5 collector + forwards_to's transform) on all cores, same table, same publishes."""
    import ctypes as C

    from oracle import oracle as orc
    flags = deliver_flags(W["n_sub"], DELIVER_SECONDARY_V5)
    prng = np.random.default_rng(12)
    from_client = np.ascontiguousarray(prng.choice(client, size=n_t), dtype=np.uint32)
    qr = np.ascontiguousarray(prng.integers(0, 3, size=n_t) | (prng.integers(0, 2, size=n_t) << 2), dtype=np.uint8)
    g = L.hr_new(1, 0)
    t = time.time()
    assert L.hr_restore_bulk_ex(g, blob.ctypes.data, offs.ctypes.data, client.ctypes.data, qos.ctypes.data, flags.ctypes.data, W["n_sub"]) == 0
    log(f"forwards e2e config {cfg}: GpuRouter::restore of {W['n_sub']} relations (10 % v5) in {time.time() - t:.1f}s")
    rec = {"metric": f"Shared::forwards publishes/sec, delivery words -> sessions (config {cfg}, v5 fraction {DELIVER_SECONDARY_V5})", "unit": "publishes/s", "threads": cores,
           "config": {"workload": f"BASELINE.json configs[{cfg - 1}]: {W['n_sub']} subscriptions ({DELIVER_SECONDARY_V5:.0%} MQTT v5: No Local / RAP), {n_t} publishes cycled, "
                                  f"publishers drawn from the subscribers", "batcher": {"max_batch": 4096, "max_delay_us": 200}},
           "gpu_async": []}
    # (the shape that measured best at both configs, profiles/r06f_*: 4 submitters, 8 k publishes outstanding, 32 completion threads, 2 passes in flight — a
    # delivery pass carries 12 bytes per HIT to the host, so fewer, fuller passes and less outstanding work than the filter-id form of Router::matches)
    # (r7: once the publisher's owner id travels with its From and the completions take the table's lock once per run, FEWER threads measure best at both
    # configs — 4 submitters / 8 k outstanding / 32 completion threads / 2 passes in flight: 4.6 M publishes/s with p99 1.4 ms at config 2, against
    # 3.4-3.8 M with p99 20-50 ms for the larger shapes: profiles/r07t_*)
    best = (4, 8192, 32, 2)
    shapes = [best] if (args.e2e_submitters, args.e2e_outstanding, args.e2e_workers, args.e2e_passes) == (8, 16384, 0, 3) else \
             [(args.e2e_submitters, args.e2e_outstanding, args.e2e_workers or max(8, min(64, cores // 4)), args.e2e_passes)]
    if args.e2e_sweep:
        shapes += [x for x in [(4, 8192, 32, 2), (8, 16384, 64, 3), (8, 65536, 64, 4), (8, 16384, 128, 3), (12, 32768, 128, 4)] if x != best]
    for subm, outst, workers, passes in shapes:
        res = (C.c_uint64 * 13)()
        wall = C.c_double(0)
        lat = np.zeros(200_000, dtype=np.float32)
        nl = C.c_uint32(0)
        L.hr_forwards_run_async(g, tb.ctypes.data, to.ctypes.data, n_t, from_client.ctypes.data, qr.ctypes.data, subm, outst, workers, passes, 4096, 200, 1.0, res, C.byref(wall), None, 0, None)      # warm
        churn = start_churn(L, g) if getattr(args, "e2e_churn", False) else None
        L.hr_forwards_run_async(g, tb.ctypes.data, to.ctypes.data, n_t, from_client.ctypes.data, qr.ctypes.data, subm, outst, workers, passes, 4096, 200, 3.0 if quick else 5.0, res,
                                C.byref(wall), lat.ctypes.data, len(lat), C.byref(nl))
        churn_rec = stop_churn(churn)
        l = np.sort(lat[:nl.value])
        rec["gpu_async"].append({"submitters": subm, "outstanding": outst, "workers": workers, "passes_in_flight": passes,
                                 "value": round(res[0] / wall.value, 1), "recipients_per_s": round(res[1] / wall.value, 1), "device_passes": int(res[2]),
                                 "publishes_per_pass": round(res[0] / max(1, res[2]), 1), "errors": int(res[3]), "host_path_publishes": int(res[4]), "resubmitted_publishes": int(res[12]), "wall_s": round(wall.value, 2),
                                 "publishes": int(res[0]), "recipients": int(res[1]),
                                 "batcher_ms_per_pass": {"collect": round(res[6] / max(1, res[2]) / 1e6, 3), "device_pass": round(res[7] / max(1, res[2]) / 1e6, 3),
                                                         "dispatch": round(res[8] / max(1, res[2]) / 1e6, 3)},
                                 "worker_task_us": round(res[9] / max(1, res[10]) / 1e3, 1), "worker_tasks": int(res[10]), "max_task_queue": int(res[11]),
                                 "latency_us": {"p50": round(float(l[len(l) // 2]), 1), "p99": round(float(l[int(len(l) * 0.99)]), 1)} if len(l) else None})
        if churn_rec:
            rec["gpu_async"][-1]["subscribe_churn_under_load"] = churn_rec
        log(f"forwards e2e config {cfg}: {rec['gpu_async'][-1]}")
    L.hr_free(g)
    o = orc.DefaultRouter()
    o.add_bulk_ex(W["blob"], W["offs"], W["client"], W["qos"], flags)
    n_c = n_t if cfg == 2 else int(min(n_t, 60_000 * cores / 256 + 2000))
    idx = np.arange(n_c)
    cb, co = prefix(W, n_c)
    o.forwards_timed(cb, co, from_client[idx], qr[idx], cores)
    sec, rows, hits, reps = 0.0, 0, 0, 0
    while sec < 2.0 and reps < 400:
        s1, o1 = o.forwards_timed(cb, co, from_client[idx], qr[idx], cores)
        sec += s1; rows += o1["rows"]; hits += o1["hits"]; reps += 1
    rec["cpu_reference_port"] = {"value": round(n_c * reps / sec, 1), "recipients_per_s": round(rows / sec, 1), "relations_visited_per_s": round(hits / sec, 1), "threads": cores,
                                 "kind": "port", "sample": n_c, "sweeps": reps,
                                 "what": "oracle DefaultRouter::forwards_shaped: _matches with the v3 / v5 collector (router.rs:174-265, types.rs:510-540) + forwards_to's "
                                         "per-recipient transform (shared.rs:886-908); rows are built and dropped, no channel"}
    # the two sides reach the same recipients: per publish of the sample, rows delivered by the oracle == recipients counted by the device path
    rec["recipients_per_publish"] = {"gpu": round(rec["gpu_async"][0]["recipients"] / max(1, rec["gpu_async"][0]["publishes"]), 2),
                                     "cpu_port_sample": round(rows / max(1, reps * n_c), 2)}
    best = max(x["value"] for x in rec["gpu_async"])
    rec["value"] = best
    rec["value_async_submit"] = best
    rec["vs_cpu_port"] = round(best / rec["cpu_reference_port"]["value"], 2)
    del o
    return rec


def time_format(args):
    """--time-format NAME: build the config's table, then time `--steps` passes of ONE result format and nothing else (no parity, no
    baseline, no secondaries) — the quick A/B line for kernel geometry sweeps, and the command to put under `rocprofv3 --kernel-trace`
    (tools/trace_gaps.py turns the trace into busy time, idle gaps and per-kernel totals of a pass).
    --ab-env A=1,A=2,A=2+B=7: the same table and batch timed once per variant of environment switches the library reads per launch
    (RGR_COMPACT_LP, RGR_WINDOW_HITS); one JSON line per variant, then — unless --no-ab-check — one more line: per-topic digests of a
    full pass under the FASTEST variant against a full pass under the FIRST (every topic of the batch)."""
    import torch
    from rmqtt_amd import capi
    W = gen_workload(args.config, args.scale)
    names = args.time_format.split(",")
    deliver = any(nm in ("deliver", "deliver8") for nm in names)        # the delivery stage (delivery words + v5 dedup) as 12-byte tuples / 8-byte hits: `--deliver V5FRAC`, default 0.1
    if deliver and (any(nm not in ("deliver", "deliver8") for nm in names) or W["retain"]):
        raise SystemExit("--time-format deliver / deliver8 stand alone (the table carries the delivery flags) and need a router config")
    r = capi.Router(device=0, window_hits=args.window_hits, collect_walk_stats=False)
    v5 = (args.deliver if args.deliver >= 0 else DELIVER_SECONDARY_V5) if deliver else -1.0
    build_table(r, W, W["blob"], W["offs"], np.arange(W["n_sub"], dtype=np.uint32), W["qos"], deliver_frac=v5)
    batch = r.retain_batch(W["tb"], W["to"]) if W["retain"] else r.batch(W["tb"], W["to"])
    W["publish_attrs"] = None
    if deliver:
        pa = np.zeros(W["n_pub"], dtype=capi.PUBLISH_ATTR_DTYPE)
        prng = np.random.default_rng(12)
        pa["from_id"] = prng.choice(W["client"].astype(np.uint32), size=W["n_pub"])
        pa["qos_retain"] = prng.integers(0, 3, size=W["n_pub"]) | (prng.integers(0, 2, size=W["n_pub"]) << 2)
        batch.set_publish_attrs(pa)
        W["publish_attrs"] = pa
        W["v5_frac"] = v5
    # --ab-env "A=1,A=2,A=2+B=7": one variant per comma, a variant is one or more NAME=value joined by '+'; names a variant does not
    # set are unset for it
    ab_name, ab_values = None, [None]
    if args.ab_env:
        ab_values = [dict(kv.split("=", 1) for kv in item.split("+")) for item in args.ab_env.split(",")]
        ab_name = sorted({k for v in ab_values for k in v})
    for name in names:                                 # (several formats: one table build for all of them)
        _time_one_format(args, W, r, batch, name, ab_name, ab_values)
    batch.close(); r.close()
    return 0


def _set_variant(names, variant):
    """Environment of one --ab-env variant: its assignments, every other name of the sweep unset."""
    for k in names or ():
        os.environ.pop(k, None)
    for k, v in (variant or {}).items():
        os.environ[k] = v


def _time_one_format(args, W, r, batch, name, ab_name, ab_values):
    import torch
    from rmqtt_amd import capi
    deliver = name in ("deliver", "deliver8")
    positions = name == "positions"                    # retained path: tuples with positions of the preorder value array (rgr_batch_set_retain_positions)
    fmt = capi.RGR_FORMAT_DELIVER8 if name == "deliver8" else capi.RGR_FORMAT_TUPLE if (deliver or positions) else FORMAT_NAMES.index(name)
    want_order = args.topic_order == "walk" and not W["retain"] and name != "deliver"          # (12-byte delivery tuples: caller order only)
    if not W["retain"] and not want_order:
        batch.set_order(False)
    batch.set_format(fmt)
    if want_order:
        batch.set_order(True)
    if W["retain"]:
        batch.set_retain_positions(positions)
    bph = {"tuple": 12, "soa": 5, "packed": 4, "runs": 0, "ids24": 3, "deliver": 12, "deliver8": 8, "positions": 12}[name]
    results = []
    for val in ab_values:
        _set_variant(ab_name, val)
        for _ in range(args.warmup):
            batch.run()
        r.stats_reset()
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(args.steps):
            hits, nwin = batch.run()
        dt = time.time() - t
        st = r.stats()
        rec = {"format": name, "config": args.config, "scale": args.scale, "window_hits": args.window_hits or "default",
               "topic_order": "walk" if want_order else "caller",
               "value": round(W["n_pub"] * args.steps / dt, 1), "ms_per_step": round(dt * 1e3 / args.steps, 3), "windows_per_step": int(nwin),
               "hits_per_step": int(hits), "kernel_ms_per_step": {"walk": round(st["walk_ms"] / args.steps, 3), "scan_compact_tiles": round(st["scan_ms"] / args.steps, 3),
                                                                   "expand": round(st["expand_ms"] / args.steps, 3)},
               "expand_avg_launch_ms": round(st["expand_ms"] / max(1, st["expand_launches"]), 4),
               "expand_store_GBps": round(hits * args.steps * bph / max(1e-9, st["expand_ms"] / 1e3) / 1e9, 1),
               "extra_flags": os.environ.get("RGR_EXTRA_FLAGS", "")}
        if deliver:
            rec["v5_frac"] = W["v5_frac"]
            rec["kernel_ms_per_step"]["dedup"] = round(st["dedup_ms"] / args.steps, 3)
            rec["dedup_avg_launch_ms"] = round(st["dedup_ms"] / max(1, st["dedup_launches"]), 4)
        if ab_name:
            rec["env"] = val
        results.append(rec)
        print(json.dumps(rec), flush=True)
    if ab_name and len(ab_values) > 1 and not args.no_ab_check and fmt != capi.RGR_FORMAT_RUNS and not positions:
        best = max(range(len(results)), key=lambda i: results[i]["value"])
        if best == 0:
            best = max(range(1, len(results)), key=lambda i: results[i]["value"])
        if deliver:
            # the delivery words of whole windows (first, middle, last) of a pass under the fastest variant against the torch
            # restatement of the per-hit rules + the v5 first-hit-per-client rule (delivery_parity)
            t0 = time.time()
            _set_variant(ab_name, ab_values[best])
            dp = delivery_parity(batch, W, W["publish_attrs"], fmt=fmt)
            _set_variant(ab_name, None)
            print(json.dumps({"ab_check": [ab_values[0], ab_values[best]], "format": name, "ok": bool(dp["ok"]), "delivery_parity": dp, "seconds": round(time.time() - t0, 1)}), flush=True)
            batch.set_format(capi.RGR_FORMAT_TUPLE)
            return
        qos_by_sub = torch.as_tensor(np.ascontiguousarray(W["qos"]).astype(np.int64), device="cuda") if not W["retain"] else None
        t0 = time.time()
        _set_variant(ab_name, ab_values[0])
        d0, ok0, _ = gpu_digests(batch, W["n_pub"], W["retain"], fmt, qos_by_sub=qos_by_sub)
        _set_variant(ab_name, ab_values[best])
        d1, ok1, info = gpu_digests(batch, W["n_pub"], W["retain"], fmt, qos_by_sub=qos_by_sub)
        same = bool((d0 == d1).all())
        _set_variant(ab_name, None)
        print(json.dumps({"ab_check": [ab_values[0], ab_values[best]], "format": name, "topics": int(W["n_pub"]),
                          "hits": int(d1[:, 0].sum()), "windows": int(info["windows"]), "structure_ok": bool(ok0 and ok1), "digests_equal": same,
                          "ok": bool(same and ok0 and ok1), "seconds": round(time.time() - t0, 1),
                          "what": "per-topic digests (hits, sum, order-weighted sum, sum of squares) of every window of a full pass, reduced on the device, under both values"}),
              flush=True)
        del d0, d1
    if not W["retain"]:
        batch.set_order(False)
    batch.set_format(capi.RGR_FORMAT_TUPLE)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-execute this script under torch.distributed.run with one rank per GPU
    (exactly the command the driver uses for N > 1) and return its exit status; rank 0's JSON line goes to stdout unchanged."""
    import socket
    import torch
    ndev = torch.cuda.device_count()
    if args.dist_backend == "nccl" and ndev < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node exposes {ndev} GPU(s); RCCL needs one device per rank "
                         f"(logic runs with several ranks per GPU: add --dist-backend gloo)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("self-launch: " + " ".join(cmd))
    # stdout carries ONE JSON line (rank 0's); whatever else the ranks' libraries print there (gloo's connection banner) goes to stderr
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    for line in p.stdout:
        if line.startswith("{") and line.rstrip().endswith("}"):
            sys.stdout.write(line); sys.stdout.flush()
        else:
            sys.stderr.write(line)
    return p.wait()


# ---------------------------------------------------------------------------------------------------------------------------------
# The ONE stdout line.  Round 4's driver run (BENCH_r04.json) printed a 27 KB line — every record in full, with its prose notes — and
# the driver's parser returned `parsed: null` for it (round 3's 18.7 KB line still parsed).  The stdout line is now the contract's
# keys + the numbers of every record (a few KB); the full record (all notes, samples, per-format detail) goes to
# gpurun_out/bench_detail_n<N>.json and, as one `[bench detail]` line, to stderr.
# ---------------------------------------------------------------------------------------------------------------------------------
LINE_BUDGET = 8000

def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}

def _short(s, n=90):
    return s if not isinstance(s, str) or len(s) <= n else s[: n - 1] + "\u2026"

def compact_roofline(r):
    if not isinstance(r, dict):
        return r
    out = _pick(r, ["bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_ms", "alg_bytes_per_launch",
                    "alg_achieved", "alg_frac", "frac_stores_only", "hits_per_launch", "per_rank"])
    if "traffic_source" in r:
        out["traffic_source"] = _short(r["traffic_source"], 60)
    return out

def compact_cpu_baseline(c):
    if not isinstance(c, dict):
        return c
    out = _pick(c, ["value", "unit", "cores", "kind"])
    if "sample" in c:
        out["sample"] = _short(str(c["sample"]), 110)
    if isinstance(c.get("single_thread"), dict):
        out["single_thread"] = c["single_thread"].get("value")
    return out

def compact_parity(p):
    if not isinstance(p, dict):
        return p
    out = _pick(p, ["ok", "topics", "exhaustive", "hits", "formats", "oracle_s", "windows_checked", "of_windows", "mismatching_words", "every_topic_owned_exactly_once"])
    if isinstance(p.get("oracle_cross_check"), dict):
        out["oracle_cross_check"] = _pick(p["oracle_cross_check"], ["ok", "topics", "hits"])
    if isinstance(p.get("oracle"), dict):
        out["oracle"] = _pick(p["oracle"], ["ok", "topics", "hits", "share_of_hits"])
    return out

def compact_formats(fl):
    out = []
    for f in fl or []:
        e = {"format": str(f.get("format", "")).split(":")[0].split(" ")[0]}
        e.update(_pick(f, ["bytes_written_per_hit", "value", "ms_per_step", "expand_avg_launch_ms", "expand_store_GBps", "frac_of_hbm_peak_stores_kernel"]))
        out.append(e)
    return out

def compact_record(r, top):
    if "error" in r:
        return {"config": r.get("config"), "error": _short(r["error"], 200)}
    keys = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"]
    out = _pick(r, keys if top else ["metric", "value", "unit", "steps", "warmup", "ms_per_step"])
    cfg = r.get("config") or {}
    out["config"] = dict(cfg) if top else {"workload": _short(str(cfg.get("workload", "")), 60)}
    if isinstance(r.get("exchange"), dict):
        out["exchange"] = _pick(r["exchange"], ["transport", "ms_per_step", "share_of_step", "runs_all_ranks", "hits_described", "describes_every_hit",
                                               "bytes_received_per_rank_per_step", "bytes_all_ranks_per_step"])
        out["exchange"]["form"] = _short(str(r["exchange"].get("form", "")), 48)
    out.update(_pick(r, ["hits_per_step", "hits_per_s", "kernel_ms_per_step", "pcie_inclusive_matches_per_s", "shard_hits", "shard_imbalance_max_over_mean",
                         "value_blocking_callers", "value_async_submit", "vs_cpu_port", "threads"]))
    if top:
        out.update(_pick(r, ["table", "mean_hits_per_topic"]))
        if isinstance(r.get("h2d_inclusive"), dict):
            out["h2d_inclusive_matches_per_s"] = r["h2d_inclusive"].get("matches_per_s")
    if "roofline" in r:
        out["roofline"] = compact_roofline(r["roofline"])
        if not top:
            out["roofline"] = _pick(out["roofline"], ["bound", "kernel", "achieved", "peak", "unit", "frac", "frac_stores_only", "traffic", "launches", "avg_launch_ms", "alg_bytes_per_launch"])
    if "cpu_baseline" in r:
        out["cpu_baseline"] = compact_cpu_baseline(r["cpu_baseline"])
        if not top and isinstance(out["cpu_baseline"], dict):
            out["cpu_baseline"].pop("sample", None)
    if "cpu_reference_port" in r:
        out["cpu_baseline"] = _pick(r["cpu_reference_port"], ["value", "threads", "kind"])
    if "parity_sample" in r:
        out["parity_sample"] = compact_parity(r["parity_sample"])
    if r.get("compact_formats"):
        cf = compact_formats(r["compact_formats"])
        out["compact_formats"] = cf if top else {e["format"]: e.get("value") for e in cf}
    if isinstance(r.get("pcie_inclusive_ranges"), dict):
        out["pcie_inclusive_ranges_matches_per_s"] = r["pcie_inclusive_ranges"].get("matches_per_s")
    if isinstance(r.get("retain_positions"), dict):
        out["retain_positions"] = _pick(r["retain_positions"], ["value", "ms_per_step", "expand_avg_launch_ms", "frac_of_hbm_peak_stores_kernel"])
    if isinstance(r.get("delivery_stage"), dict):
        out["delivery_stage"] = _pick(r["delivery_stage"], ["v5_fraction", "bytes_written_per_hit", "dedup_ms_per_step", "expand_store_GBps"])
        if isinstance(r["delivery_stage"].get("tuple12"), dict):
            out["delivery_stage"]["tuple12"] = _pick(r["delivery_stage"]["tuple12"], ["value", "ms_per_step"])
    if isinstance(r.get("gpu_async"), list) and r["gpu_async"]:
        out["latency_us"] = r["gpu_async"][0].get("latency_us")
    return out

def compact_line(rec, detail_path=None):
    out = compact_record(rec, True)
    if rec.get("secondary"):
        out["secondary"] = [compact_record(s, False) for s in rec["secondary"]]
    if detail_path:
        out["detail"] = detail_path
    line = json.dumps(out, separators=(", ", ": "))
    if len(line) > LINE_BUDGET and "secondary" in out:        # never let a long line cost the headline: shed secondary detail first
        for s in out["secondary"]:
            for k in ("cpu_baseline", "compact_formats", "kernel_ms_per_step", "config", "latency_us", "delivery_stage"):
                s.pop(k, None)
        line = json.dumps(out, separators=(", ", ": "))
    if len(line) > LINE_BUDGET:
        out.pop("secondary", None); out.pop("table", None)
        line = json.dumps(out, separators=(", ", ": "))
    return line

def emit_line(rec, args):
    """Full record -> gpurun_out/bench_detail_n<N>.json + one stderr line; compact record -> the one stdout line."""
    detail_path = None
    try:
        ddir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpurun_out")
        os.makedirs(ddir, exist_ok=True)
        detail_path = os.path.join("gpurun_out", f"bench_detail_n{rec.get('n_gpus', 1)}.json")
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), detail_path), "w") as f:
            json.dump(rec, f)
    except OSError as e:
        log(f"bench detail file not written: {e!r}")
        detail_path = None
    print("[bench detail] " + json.dumps(rec), file=sys.stderr, flush=True)
    print(compact_line(rec, detail_path), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config number (1-based)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (non-headline runs only)")
    ap.add_argument("--gather", choices=["none", "counts", "tuples", "runs"], default=None,
                    help="N>1 exchange step per pass.  Default `runs`: BASELINE configs[3]'s all-gatherv of subscriber hits as 16-byte run descriptors "
                         "(every rank learns every rank's hits; subs[] replicated once).  `tuples`: all-gatherv of the 12-byte tuples themselves; "
                         "`counts`: per-rank hit counts only (16 bytes per step); `none`: no exchange")
    ap.add_argument("--key-levels", type=int, default=0, help="--group: leading topic levels hashed into the shard key (default 3; 1 = SURVEY 8(e))")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="queries in the CPU-baseline sample (0 = skip baseline and parity sample, -1 = auto)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--window-hits", type=int, default=0)
    ap.add_argument("--no-d2h", action="store_true", help="skip the PCIe-inclusive pass")
    ap.add_argument("--d2h", action="store_true", help="(kept for compatibility: the PCIe-inclusive pass now runs by default)")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity sample")
    ap.add_argument("--no-formats", action="store_true", help="skip the compact-result-format passes")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 PMC passes (roofline.frac/traffic stay null)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[1] / configs[4] secondary records")
    ap.add_argument("--secondary-steps", type=int, default=5)
    ap.add_argument("--pmc-topics", type=int, default=2_000_000, help="queries per phase replayed under the counters")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--pmc-phases", default="primary", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-world", type=int, default=1, help=argparse.SUPPRESS)
    ap.add_argument("--pmc-rank", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--deliver", type=float, default=-1.0, metavar="V5FRAC",
                    help="also run the delivery stage (SURVEY 8(f)-1): this fraction of the subscriptions is MQTT v5 "
                         "(No Local / RAP / per-client dedup); 0 = v3 only. Not the headline metric.")
    ap.add_argument("--topic-order", choices=["walk", "caller"], default="walk",
                    help="order in which the timed publish batch is walked: `walk` = sorted by the leading level tokens inside the library (rgr_batch_set_order; "
                         "results per topic are identical, windows enumerate walk positions), `caller` = as handed over (rounds 1-5)")
    ap.add_argument("--deliver-format", choices=["hits8", "tuple12"], default="hits8",
                    help="answer format of the delivery stage's records: 8-byte hits {sub_id, delivery word} (RGR_FORMAT_DELIVER8; the 12-byte form is timed "
                         "beside it) or 12-byte tuples only")
    ap.add_argument("--e2e-submitters", type=int, default=8)
    ap.add_argument("--e2e-outstanding", type=int, default=16384)
    ap.add_argument("--e2e-workers", type=int, default=0, help="completion pool threads (0 = cores / 4, at most 64)")
    ap.add_argument("--e2e-passes", type=int, default=3, help="device passes in flight")
    ap.add_argument("--e2e-churn", action="store_true", help="--router-e2e, forwards leg: a thread subscribing / unsubscribing beside the publishes; its add / remove latencies are reported")
    ap.add_argument("--e2e-sweep", action="store_true", help="--router-e2e: also run a few other (submitters, outstanding, workers, passes) shapes")
    ap.add_argument("--e2e-configs", default="2,3")
    ap.add_argument("--e2e-legs", default="matches,forwards", help="--router-e2e: which consumers to time: Router::matches (SubRelationsMap per publish) and / or Shared::forwards (delivery words -> sessions)")
    ap.add_argument("--time-format", default=None, help="time passes of ONE result format only (or several, comma-separated: one table build) and exit (sweeps, kernel traces): " + ", ".join(FORMAT_NAMES) + "; or `deliver` alone: the delivery stage (--deliver V5FRAC, default 0.1)")
    ap.add_argument("--ab-env", default=None, help="with --time-format: 'A=1,A=2,A=2+B=7': time the same batch once per variant (comma-separated; '+' joins assignments) of environment switches the library reads per launch")
    ap.add_argument("--no-ab-check", action="store_true", help="with --ab-env: skip the full-pass digest comparison of the fastest value against the first")
    ap.add_argument("--router-e2e", action="store_true", help="time Router::matches through the host Router mirror + batcher beside the CPU port (configs 2 and 3)")
    ap.add_argument("--group", type=int, default=0, metavar="SHARDS",
                    help="run the single-process sharded router (rgr_group_*) with this many shards on the visible GPUs instead of the N=1 bench")
    ap.add_argument("--torch-collectives", action="store_true", help="N>1: use torch.distributed collectives instead of the library's RCCL communicator")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) for real runs; gloo lets the N>1 logic be exercised on one GPU")
    args = ap.parse_args()
    if args.gather is None:
        args.gather = "runs"            # (only read at N > 1)

    if args.pmc_child:
        return pmc_child(args)
    if args.time_format:
        return time_format(args)
    if args.group > 0:
        return measure_group(args)
    if args.router_e2e:
        return measure_router_e2e(args)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)        # `python bench.py --gpus N` on its own: spawn the N ranks (one per GPU) and pass their line through
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started inside a {world}-rank job (WORLD_SIZE): launch it with --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.dist_backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()       # several ranks may share a GPU under gloo
    torch.cuda.set_device(local_rank)
    cdev = "cuda" if args.dist_backend == "nccl" else "cpu"        # device of the collective tensors
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)

    rec, phase = measure(args, args.config, args.scale, args.steps, args.warmup, True, rank, world, local_rank, dist, cdev)
    if rank != 0:
        return 0                           # (measure() left the process group together with every other rank)

    headline = world == 1 and args.deliver < 0
    secondary = []
    sec_phases = {}
    if headline and not args.no_secondary and args.config == 3 and args.scale == 1.0:
        for cfg in (2, 5):
            try:
                srec, sph = measure(args, cfg, 1.0, args.secondary_steps, 1, False)
                secondary.append(srec)
                sec_phases[f"config{cfg}"] = (srec, sph)
            except Exception as e:        # a secondary record never takes the headline line down
                log(f"secondary config {cfg} failed: {e!r}")
                secondary.append({"config": {"workload": f"BASELINE.json configs[{cfg - 1}]"}, "error": repr(e)})

    if headline and not args.no_secondary and args.config == 3 and args.scale == 1.0:
        # the delivery stage on the headline workload with 10 % MQTT v5 subscriptions: delivery words fused into the
        # expansion + per-client first-hit dedup; a record of its own so that the driver's run times it
        try:
            drec, dph = measure(args, 3, 1.0, max(2, args.secondary_steps // 2), 1, False, deliver_frac=DELIVER_SECONDARY_V5)
            secondary.append(drec)
            sec_phases["deliver"] = (drec, dph)
        except Exception as e:
            log(f"secondary delivery-stage record failed: {e!r}")
            secondary.append({"config": {"workload": "BASELINE.json configs[2] + delivery stage"}, "error": repr(e)})

    if headline and not args.no_secondary and args.config == 3 and args.scale == 1.0:
        # the drop-in boundary end to end: Router::matches through the C++ twin's batcher at config 2 — 256 blocked callers and
        # the asynchronous shape (a few submitters, 16 k publishes outstanding, 3 passes in flight) — beside the CPU port on the same host
        try:
            args.e2e_configs = "2"
            for erec in measure_router_e2e(args, quick=True):
                secondary.append(erec)
        except Exception as e:
            log(f"secondary router-e2e record failed: {e!r}")
            secondary.append({"config": {"workload": "BASELINE.json configs[1] through Router::matches"}, "error": repr(e)})

    if (headline or world > 1) and not args.no_pmc and args.deliver < 0:
        # N > 1: rank 0's child replays RANK 0's shard (table + topics under the same shard rule) on its GPU: a per-rank roofline
        cal = load_calibration()
        pmc = run_pmc_children(args, ["primary"] + list(sec_phases), world, 0)
        attach_traffic(rec, phase, pmc.get("primary") if pmc else None, cal)
        if world > 1 and rec["roofline"].get("traffic") is not None:
            rec["roofline"]["per_rank"] = f"rank 0 of {world}: its shard replayed under the counters; every rank runs the same kernels on its own shard"
        for name, (srec, sph) in sec_phases.items():
            attach_traffic(srec, sph, pmc.get(name) if pmc else None, cal)
    if secondary:
        rec["secondary"] = secondary

    emit_line(rec, args)
    bad = [r_ for r_ in [rec] + secondary if isinstance(r_.get("parity_sample"), dict) and not r_["parity_sample"]["ok"]]
    if bad:
        log("PARITY SAMPLE FAILED: the GPU's tuples differ from the oracle's")
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
