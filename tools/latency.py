#!/usr/bin/env python3
"""Per-call latency / throughput of the host-buffer entry points — the shape a broker's micro-batcher issues:
  rgr_match_batch     host blob in, host tuples out (pinned result block, expand / D2H pipelined)
  rgr_match_filters   host blob in, matched filter ids out (the host expands from its own relations)
timed around the C call itself (the result is freed, not copied into numpy: a consumer reads it in place),
and rgr_commit after SUBSCRIBE bursts.   python tools/latency.py [config=2] [scale=1.0]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from rmqtt_amd import capi, workload as wl

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = wl.CONFIGS[cfg]
n_sub = int(c["n_sub"] * scale)
blob, offs, client, qos = wl.gen_subs(n_sub, wl.SUB_SEED + cfg, c["p_plus"], c["p_hash"], c["p_sys"])
N = 200_000
tb, to = wl.gen_topics(N, wl.PUB_SEED + cfg, 0.01, c["p_blank"])
r = capi.Router(device=0)
r.subscribe_bulk(blob, offs, None, qos); r.commit()
L = capi.lib()


def call_match(b, o):
    res = capi.Result()
    o = np.ascontiguousarray(o, dtype=np.uint64)
    capi._check(L.rgr_match_batch(r._h, b.ctypes.data, o.ctypes.data, len(o) - 1, C.byref(res)))
    n = int(res.n_hits)
    L.rgr_result_free(C.byref(res))
    return n


def call_filters(b, o):
    res = capi.FiltersResult()
    o = np.ascontiguousarray(o, dtype=np.uint64)
    capi._check(L.rgr_match_filters(r._h, b.ctypes.data, o.ctypes.data, len(o) - 1, C.byref(res)))
    n = int(res.n_pairs)
    L.rgr_filters_result_free(C.byref(res))
    return n


for name, fn, unit, sizes in (("rgr_match_batch", call_match, "tuples", (1, 16, 256, 4096, 16384, 65536)),
                              ("rgr_match_filters", call_filters, "filter ids", (1, 256, 4096, 65536, 200_000))):
    for bs in sizes:
        batches = [wl.take(tb, to, np.arange(i * bs, (i + 1) * bs) % N) for i in range(4)]
        fn(*batches[0]); fn(*batches[1])                     # warm: workspace + pinned block of this size class
        reps = max(3, min(200, 300_000 // bs))
        t0 = time.time(); units = 0
        for i in range(reps):
            units += fn(*batches[i % len(batches)])
        dt = (time.time() - t0) / reps
        per = units / reps
        print(f"config {cfg} x{scale}: {name:18s} batch {bs:7d}: {dt * 1e3:9.3f} ms/call  {bs / dt:14.0f} topics/s  "
              f"{per / max(dt, 1e-9) / 1e6:10.1f} M {unit}/s" + (f"  {per * 12 / max(dt, 1e-9) / 1e9:6.2f} GB/s of tuples" if unit == "tuples" else ""), flush=True)

# ---- commit latency: a burst of SUBSCRIBEs followed by rgr_commit (delta path)
fb, fo, _, fq = wl.gen_subs(64 * 40, 777, c["p_plus"], c["p_hash"], c["p_sys"])
next_id = n_sub
for burst in (1, 64):
    ts, tc = [], []
    for it in range(20):
        idx = np.arange(it * burst, (it + 1) * burst)
        bb, bo = wl.take(fb, fo, idx)
        t0 = time.time()
        r.subscribe_bulk(bb, bo, np.arange(next_id, next_id + burst, dtype=np.uint32), fq[idx])
        t1 = time.time()
        r.commit()
        ts.append(t1 - t0); tc.append(time.time() - t1)
        next_id += burst
    st = r.stats()
    print(f"config {cfg} x{scale}: {burst:3d} new subscriptions: rgr_subscribe_bulk median {np.median(ts) * 1e3:8.3f} ms, rgr_commit median {np.median(tc) * 1e3:8.3f} ms  "
          f"(commits so far: {st['commits_full']} full, {st['commits_delta']} delta; table {st['n_subs']} subs)", flush=True)
