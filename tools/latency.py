#!/usr/bin/env python3
"""Per-call latency of the host-buffer entry point (rgr_match_batch) for small batches — the
shape a broker's micro-batcher would issue — and the PCIe-inclusive throughput for big ones."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rmqtt_amd import capi, workload as wl

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = wl.CONFIGS[cfg]
n_sub = int(c["n_sub"] * scale)
blob, offs, client, qos = wl.gen_subs(n_sub, wl.SUB_SEED + cfg, c["p_plus"], c["p_hash"], c["p_sys"])
tb, to = wl.gen_topics(200_000, wl.PUB_SEED + cfg, 0.01, c["p_blank"])
r = capi.Router(device=0)
r.subscribe_bulk(blob, offs, None, qos); r.commit()
for bs in (1, 16, 256, 4096, 65536, 200_000):
    reps = max(3, min(300, 200_000 // bs))
    batches = [wl.take(tb, to, np.arange(i * bs, (i + 1) * bs) % 200_000) for i in range(min(reps, 8))]
    r.match_batch(*batches[0])
    t0 = time.time(); hits = 0
    for i in range(reps):
        res = r.match_batch(*batches[i % len(batches)]); hits += len(res["tuples"])
    dt = (time.time() - t0) / reps
    print(f"config {cfg} x{scale}: batch {bs:7d}: {dt * 1e3:9.3f} ms/call  {bs / dt:14.0f} topics/s  {hits / reps / max(dt, 1e-9) / 1e6:10.1f} M tuples/s  (host blob in, host tuples out)")

# ---- the filter-run result form (rgr_match_filters): matched filter ids only, the host expands from its own relations
for bs in (1, 256, 4096, 65536, 200_000):
    reps = max(3, min(200, 400_000 // bs))
    batches = [wl.take(tb, to, np.arange(i * bs, (i + 1) * bs) % 200_000) for i in range(min(reps, 8))]
    r.match_filters(*batches[0])
    t0 = time.time(); pairs = 0
    for i in range(reps):
        res = r.match_filters(*batches[i % len(batches)]); pairs += len(res["filter_ids"])
    dt = (time.time() - t0) / reps
    print(f"config {cfg} x{scale}: batch {bs:7d}: {dt * 1e3:9.3f} ms/call  {bs / dt:14.0f} topics/s  {pairs / reps / max(dt, 1e-9) / 1e6:10.2f} M filter ids/s  (rgr_match_filters: host blob in, matched filter ids out)")

# ---- commit latency: a burst of SUBSCRIBEs followed by rgr_commit (delta path)
rng = np.random.default_rng(0)
fb, fo, _, fq = wl.gen_subs(64 * 40, 777, c["p_plus"], c["p_hash"], c["p_sys"])
next_id = n_sub
for burst in (1, 64):
    ts = []
    for it in range(20):
        idx = np.arange(it * burst, (it + 1) * burst)
        bb, bo = wl.take(fb, fo, idx)
        t0 = time.time()
        r.subscribe_bulk(bb, bo, np.arange(next_id, next_id + burst, dtype=np.uint32), fq[idx])
        r.commit()
        ts.append(time.time() - t0)
        next_id += burst
    st = r.stats()
    print(f"config {cfg} x{scale}: {burst:3d} new subscriptions + rgr_commit: median {np.median(ts) * 1e3:8.3f} ms  "
          f"(commits so far: {st['commits_full']} full, {st['commits_delta']} delta; table {st['n_subs']} subs)")
