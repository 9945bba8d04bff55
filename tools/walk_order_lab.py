#!/usr/bin/env python3
"""Does the order of the publish topics inside a batch matter to the walk kernel?

Times the walk-dominated config (BASELINE configs[1]: 1 M subscriptions, 0.46 hits/topic) with the
batch in generator order, sorted by topic string (neighbouring lanes then share their top-of-trie
records) and randomly shuffled, through the product API only — no kernel change.  If sorted order
wins clearly, a device-side reorder (radix sort of (first tokens, topic index) before the walk,
results scattered back through the index) is worth building.

  python tools/walk_order_lab.py [config=2] [scale=1.0] [reps=5]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from rmqtt_amd import capi, workload as wl

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
c = wl.CONFIGS[cfg]
n_sub, n_pub = int(c["n_sub"] * scale), int(c["n_pub"] * scale)
blob, offs, client, qos = wl.gen_subs(n_sub, wl.SUB_SEED + cfg, c["p_plus"], c["p_hash"], c["p_sys"])
tb, to = wl.gen_topics(n_pub, wl.PUB_SEED + cfg, 0.01, c["p_blank"])
r = capi.Router(device=0, collect_walk_stats=True)
r.subscribe_bulk(blob, offs, None, qos)
r.commit()

strings = wl.strings(tb, to)
orders = {
    "generator order": np.arange(n_pub),
    "sorted by topic string": np.argsort(np.array(strings, dtype=object), kind="stable"),
    "shuffled": np.random.default_rng(1).permutation(n_pub),
}
ref_hits = None
for name, order in orders.items():
    b, o = wl.take(tb, to, order)
    batch = r.batch(b, o)
    batch.run()                                   # warm-up
    r.stats_reset()
    t0 = time.time()
    for _ in range(reps):
        hits, _ = batch.run()
    dt = (time.time() - t0) / reps
    st = r.stats()
    ref_hits = hits if ref_hits is None else ref_hits
    assert hits == ref_hits, "the hit count must not depend on the batch order"
    print(f"config {cfg} x{scale}  {name:24s}: {n_pub / dt / 1e6:9.1f} M topics/s  walk {st['walk_ms'] / reps:8.3f} ms  "
          f"scan/compact {st['scan_ms'] / reps:7.3f} ms  expand {st['expand_ms'] / reps:8.3f} ms  "
          f"visited/topic {st['visited_nodes'] / max(1, st['topics']):.1f}")
    batch.close()
