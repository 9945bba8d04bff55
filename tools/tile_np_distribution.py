"""Pairs per expansion tile at BASELINE config 3 (CPU only: seeded generator + oracle, full 10 M-subscription table, first N publish topics).

A tile is 2048 consecutive output positions of a window; a pair is one (topic, subscriber run).  The compact expansions take a fast path
for tiles that lie inside ONE run (np == 1) and a staged path otherwise; this prints how the tiles (and the hits) of the workload split by
np, and how often all T consecutive tiles of a block are single-run — the numbers the round-4 kernel variants were designed from.
  python tools/tile_np_distribution.py [N=3000]  > profiles/r04n_config3_tile_np_distribution.txt"""
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
blob, offs = W["blob"], W["offs"]
t = time.time()
r = orc.DefaultRouter()
r.add_bulk(blob, offs, W["client"], W["qos"])
print(f"# oracle table built in {time.time() - t:.0f} s; first {N} publish topics", flush=True)
sb, so = bench.prefix(W, N)
res = r.match_flat(sb, so)
hit_off = np.asarray(res["hit_offsets"]).astype(np.int64)
f = np.asarray(res["filter_ids"]).astype(np.int64)
H = len(f)
# run starts: the filter changes, or a topic starts (two topics can end / start with the same filter)
start = np.ones(H, dtype=bool)
start[1:] = f[1:] != f[:-1]
if hit_off is not None:
    tb = hit_off[:-1]
    start[tb[tb < H]] = True
run_start = np.flatnonzero(start)
run_len = np.diff(np.append(run_start, H))
print(f"hits {H}  runs {len(run_start)}  mean run {H / len(run_start):.0f}  tiles {H // TILE}")
# np of tile k = runs that intersect [k*TILE, (k+1)*TILE)
ntiles = H // TILE
first_run = np.searchsorted(run_start, np.arange(ntiles) * TILE, side="right") - 1
last_run = np.searchsorted(run_start, (np.arange(ntiles) + 1) * TILE - 1, side="right") - 1
npairs = last_run - first_run + 1
print("np    tiles     share   cumulative")
tot = ntiles
cum = 0
for k in list(range(1, 17)) + [24, 32, 48, 64, 128, 256, 2048]:
    lo = k if k <= 16 else {24: 17, 32: 25, 48: 33, 64: 49, 128: 65, 256: 129, 2048: 257}[k]
    c = int(((npairs >= lo) & (npairs <= k)).sum())
    cum += c
    print(f"{('%d' % k) if lo == k else ('%d-%d' % (lo, k)):8s} {c:8d}  {c / tot * 100:6.2f} %  {cum / tot * 100:6.2f} %")
print(f"max np {int(npairs.max())}, mean np of multi-run tiles {npairs[npairs > 1].mean():.2f}")
for T in (2, 4, 8):
    nb = ntiles // T
    blk = npairs[:nb * T].reshape(nb, T)
    allone = (blk == 1).all(axis=1).mean()
    le8 = (blk <= 8).all(axis=1).mean()
    le16 = (blk <= 16).all(axis=1).mean()
    print(f"blocks of {T} tiles: all single-run {allone * 100:.1f} %, all np <= 8 {le8 * 100:.1f} %, all np <= 16 {le16 * 100:.1f} %, "
          f"mean multi-run tiles per block {(blk > 1).sum(axis=1).mean():.2f}")
# groups of four consecutive positions that straddle a run boundary
g = np.arange(H // 4) * 4
own0 = np.searchsorted(run_start, g, side="right")
own3 = np.searchsorted(run_start, g + 3, side="right")
print(f"groups of 4 positions that straddle a run boundary: {(own0 != own3).mean() * 100:.3f} %")
