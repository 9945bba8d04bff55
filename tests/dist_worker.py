"""Worker of tests/test_distributed.py: one rank of the sharded N>1 path on CPU (gloo).

Each rank keeps the subscriptions it owns under the first-two-level hash (plus the replicated
wildcard-rooted ones), matches only the publish topics it owns (host emulator backend — the
sharding logic is backend-independent), then all ranks all-gatherv their tuples; rank 0 checks
the union bit-exactly against the oracle over the UNSHARDED table.

RMQTT_DIST_BACKEND=hip (the `-m gpu` variant): every rank is a process of its own that drives the PRODUCT library
(capi.Router, all ranks on GPU 0 of a one-GPU box) and the exchange goes through torch.distributed (gloo).  The library's
own exchange (rgr_comm_*: ncclCommInitRank + ncclAllGather + the send/recv group) cannot form a communicator with two
ranks on ONE device — RCCL refuses duplicate devices — so across processes it runs only on the multi-GPU node the driver
uses (bench.py --gpus N); its protocol (counts round, then exact-size payloads, failure flag) is exercised in-process by
tests/test_group_gpu.py over the same-device transport and with a world of one over RCCL."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
from rmqtt_amd import shard  # noqa: E402
from rmqtt_amd import workload as wl  # noqa: E402
from tests.emu import emu  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cfg = 3
    c = wl.CONFIGS[cfg]
    n_sub, n_pub = 20000, 3000
    blob, offs, client, qos = wl.gen_subs(n_sub, wl.SUB_SEED + cfg, c["p_plus"], c["p_hash"], c["p_sys"])
    tb, to = wl.gen_topics(n_pub, wl.PUB_SEED + cfg, 0.01, c["p_blank"])
    f_owner = shard.assign(blob, offs, world, is_filter=True)
    t_owner = shard.assign(tb, to, world, is_filter=False)
    keep_f = np.nonzero((f_owner == rank) | (f_owner < 0))[0]
    keep_t = np.nonzero(t_owner == rank)[0]
    fb, fo = shard.take(blob, offs, keep_f)
    pb, po = shard.take(tb, to, keep_t)
    if os.environ.get("RMQTT_DIST_BACKEND") == "hip":
        from rmqtt_amd import capi
        r = capi.Router(device=0, window_hits=50_000)
        assert r.subscribe_bulk(fb, fo, keep_f.astype(np.uint32), qos[keep_f]) == 0
        r.commit()
    else:
        r = emu.EmuRouter()
        assert r.subscribe_bulk(fb, fo, keep_f.astype(np.uint32), qos[keep_f]) == 0
    got = r.match_batch(pb, po)
    t = got["tuples"]
    local = torch.from_numpy(np.stack([keep_t[t["topic_idx"]].astype(np.int64), t["sub_id"].astype(np.int64),
                                       t["qos_flags"].astype(np.int64)], axis=1).reshape(-1, 3))
    allt, counts = shard.allgatherv_tuples(local, world, rank, dist, "cpu")
    assert sum(counts) == allt.shape[0]
    ok = 1
    if rank == 0:
        o = orc.DefaultRouter()
        assert o.add_bulk(blob, offs, client, qos) == 0
        exp = o.match_flat(tb, to)
        a = allt.numpy()
        # canonical order: by topic, preserving each rank's (already reference-ordered) run
        order = np.argsort(a[:, 0], kind="stable")
        a = a[order]
        n_exp = len(exp["sub_ids"])
        counts_t = np.diff(exp["hit_offsets"]).astype(np.int64)
        exp_topic = np.repeat(np.arange(n_pub, dtype=np.int64), counts_t)
        ok = int(a.shape[0] == n_exp and np.array_equal(a[:, 0], exp_topic) and np.array_equal(a[:, 1], exp["sub_ids"].astype(np.int64))
                 and np.array_equal(a[:, 2] & 0xFF, exp["qos"].astype(np.int64)))
        print(f"rank0: {n_exp} hits, per-rank hits {counts}, replicated filters {(f_owner < 0).sum()} of {n_sub}", flush=True)
    flag = torch.tensor([ok])
    dist.broadcast(flag, 0)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
