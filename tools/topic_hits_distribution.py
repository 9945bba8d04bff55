"""Hits per publish topic at BASELINE config 3 (CPU only: seeded generator + oracle, full 10 M-subscription table, first N topics): how the
work items of the v5 dedup's topic pass (one block per topic that spans tiles) split by size, and how the hits of a topic split over its
runs — the numbers behind DESIGN 16 (r5).
  python tools/topic_hits_distribution.py [N=3000]  > profiles/r05n_config3_topic_hits_distribution.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oracle import oracle as orc  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
TILE = 2048
W = bench.gen_workload(3, 1.0)
t = time.time()
r = orc.DefaultRouter()
r.add_bulk(W["blob"], W["offs"], W["client"], W["qos"])
print(f"# oracle table built in {time.time() - t:.0f} s; first {N} publish topics", flush=True)
sb, so = bench.prefix(W, N)
res = r.match_flat(sb, so)
hit_off = np.asarray(res["hit_offsets"]).astype(np.int64)
f = np.asarray(res["filter_ids"]).astype(np.int64)
h = np.diff(hit_off)
H = int(h.sum())
print(f"topics {N}  hits {H}  mean {H / N:.0f}  median {int(np.median(h))}  max {int(h.max())}")
tiles = (hit_off[1:] - 1) // TILE - hit_off[:-1] // TILE + 1
tiles[h == 0] = 0
print("tiles spanned   topics   share   share of hits")
for lo, hi in ((0, 0), (1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 32), (33, 64), (65, 10 ** 9)):
    m = (tiles >= lo) & (tiles <= hi)
    print(f"{lo:>3}..{hi if hi < 10 ** 9 else 'inf':<6}  {int(m.sum()):8d}  {m.mean() * 100:6.2f} %  {h[m].sum() / H * 100:6.2f} %")
# a topic's largest run against the rest
big = np.zeros(N)
for k in range(N):
    a, b = hit_off[k], hit_off[k + 1]
    if b > a:
        ff = f[a:b]
        st = np.flatnonzero(np.r_[True, ff[1:] != ff[:-1]])
        big[k] = np.diff(np.r_[st, b - a]).max()
m = h > 0
print(f"largest run of a topic / its hits: hit-weighted mean {big[m].sum() / H:.3f}, plain mean {(big[m] / h[m]).mean():.3f}")
