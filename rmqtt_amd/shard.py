"""Multi-GPU sharding rule — host-side helpers over the C ABI.

owner(topic)  = H(first KEY_LEVELS levels) mod G
owner(filter) = the same, or -1 (replicate on every shard) when one of its first KEY_LEVELS
                levels is a wildcard.  A topic's full match set then lives on its owner alone, so the
                data path needs no collective; ranks only exchange hit counts (and,
                optionally, tuples: all-gatherv) afterwards.
"""
import ctypes as C

import numpy as np

from . import capi
from .workload import take  # noqa: F401  (re-export: gather a subset of a string batch)


KEY_LEVELS = 3


def assign(blob, offsets, n_shards, is_filter, key_levels=KEY_LEVELS):
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    out = np.zeros(n, dtype=np.int32)
    rc = capi.lib().rgr_shard_assign(blob.ctypes.data, offsets.ctypes.data, n, n_shards, int(is_filter), key_levels, out.ctypes.data)
    if rc != 0:
        raise capi.RgrError(rc, capi.lib().rgr_last_error().decode())
    return out


def allgatherv_rows(local, world, rank, dist, device):
    """all-gatherv of fixed-width rows: local is a tensor [n, k] ((topic_idx, sub_id, qos) tuples: k = 3; run descriptors
    {shard, src, len, topic}: k = 4).  RCCL has no native allgatherv: gather the counts, then one padded all_gather; returns the
    concatenation in rank order and the per-rank counts.  (world_size-2 gloo test: tests/test_distributed.py.)"""
    import torch
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt)
    counts = [int(x.item()) for x in counts]
    mx = max(counts) if counts else 0
    pad = torch.zeros((mx, local.shape[1]), dtype=local.dtype, device=device)
    pad[: local.shape[0]] = local
    bufs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0), counts


allgatherv_tuples = allgatherv_rows
