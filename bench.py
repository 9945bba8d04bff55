#!/usr/bin/env python3
"""bench.py — publish-topic matches/sec of the MI355X topic matcher (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" is one pass of the hot path (trie walk + subscriber expansion into
(topic_idx, sub_id, qos) tuples) over one batch of publish topics whose '/'-tokenised
form is already resident in HBM.  Workload at N=1: BASELINE.json configs[2] — 10 M
subscriptions with mixed '+'/'#' (seeded generator of SURVEY.md §8(d)), 10 M publishes,
1x MI355X.  N>1: the same table hash-sharded by the first three topic levels, publishes
routed to their owner rank (strong scaling: total work fixed); ranks exchange per-rank
hit counts (all-gather over RCCL), tuples stay on the owning GPU unless --gather tuples.

One JSON line on rank 0: value = whole-job publish-topic matches/s; `roofline` for the
dominant kernel (expand) from HIP events on the library's stream; `cpu_baseline` = the
oracle (C++ restatement of DefaultRouter, "port") timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def log(msg, rank=0):
    if rank == 0:
        print(f"[bench +{time.time() - T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


T0 = time.time()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config number (1-based)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (non-headline runs only)")
    ap.add_argument("--gather", choices=["none", "counts", "tuples"], default="counts")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="topics in the CPU-baseline sample (0 = skip, -1 = auto)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--window-hits", type=int, default=0)
    ap.add_argument("--d2h", action="store_true", help="also report the PCIe-inclusive rate (copies every window to host)")
    ap.add_argument("--deliver", type=float, default=-1.0, metavar="V5FRAC",
                    help="also run the delivery stage (SURVEY 8(f)-1): this fraction of the subscriptions is MQTT v5 "
                         "(No Local / RAP / per-client dedup); 0 = v3 only. Not the headline metric.")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) for real runs; gloo lets the N>1 logic be exercised on one GPU")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
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

    from rmqtt_amd import capi, shard
    from rmqtt_amd import workload as wl

    cfg = args.config
    c = wl.CONFIGS[cfg]
    n_sub = max(1, int(c["n_sub"] * args.scale))
    n_pub = max(1, int(c["n_pub"] * args.scale))

    retain = cfg == 5
    # ---- synthetic inputs (identical on every rank: seeded)
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

    sub_ids = np.arange(n_sub, dtype=np.uint32)
    if world > 1 and retain:
        raise SystemExit("config 5 (retained path) is a single-GPU config")
    if world > 1:
        f_owner = shard.assign(blob, offs, world, is_filter=True)
        t_owner = shard.assign(tb, to, world, is_filter=False)
        keep_f = np.nonzero((f_owner == rank) | (f_owner < 0))[0]
        keep_t = np.nonzero(t_owner == rank)[0]
        blob_r, offs_r = shard.take(blob, offs, keep_f)
        tb_r, to_r = shard.take(tb, to, keep_t)
        sub_ids_r, qos_r = sub_ids[keep_f], qos[keep_f]
    else:
        blob_r, offs_r, tb_r, to_r, sub_ids_r, qos_r = blob, offs, tb, to, sub_ids, qos
    my_topics = len(to_r) - 1

    # ---- table build + device-resident batch
    r = capi.Router(device=local_rank, window_hits=args.window_hits, collect_walk_stats=True)
    t = time.time()
    if retain:
        rej = r.retain_add_bulk(blob_r, offs_r)
        r.retain_commit()
    else:
        deliver = args.deliver >= 0 and world == 1
        flags_r = None
        if deliver:
            drng = np.random.default_rng(11)
            is5 = drng.random(n_sub) < args.deliver
            flags_r = (is5 * capi.RGR_SUB_V5 | (is5 & (drng.random(n_sub) < 0.3)) * capi.RGR_SUB_NO_LOCAL |
                       (is5 & (drng.random(n_sub) < 0.5)) * capi.RGR_SUB_RAP).astype(np.uint8)
        rej = r.subscribe_bulk(blob_r, offs_r, sub_ids_r, qos_r, flags_r)
        if deliver:
            r.sub_attrs_bulk(client.astype(np.uint32), client.astype(np.uint32))     # one Id per client
        r.commit()
    build_s = time.time() - t
    st0 = r.stats()
    if retain:
        log(f"retain table: {n_sub - rej} topics ({rej} names rejected by the parser), built in {build_s:.1f}s", rank)
    else:
        log(f"table: {st0['n_filters']} filters, {st0['n_subs']} subs, {st0['n_nodes']} trie nodes, "
            f"{st0['table_bytes_device'] / 2**30:.2f} GiB in HBM, built in {build_s:.1f}s (rejected {rej})", rank)
    t = time.time()
    batch = r.retain_batch(tb_r, to_r) if retain else r.batch(tb_r, to_r)
    log(f"batch: {my_topics} topics tokenised + uploaded in {time.time() - t:.1f}s", rank)
    if not retain and args.deliver >= 0 and world == 1:
        pa = np.zeros(my_topics, dtype=capi.PUBLISH_ATTR_DTYPE)
        prng = np.random.default_rng(12)
        pa["from_id"] = prng.choice(client.astype(np.uint32), size=my_topics)
        pa["qos_retain"] = prng.integers(0, 3, size=my_topics) | (prng.integers(0, 2, size=my_topics) << 2)
        batch.set_publish_attrs(pa)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    class _DevTuples:   # zero-copy torch view of a window's device tuples
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n, 3), "typestr": "<i4", "data": (ptr, False), "version": 2}

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
            local = torch.as_tensor(_DevTuples(w.d_tuples, n), device="cuda") if n else torch.zeros((0, 3), dtype=torch.int32, device="cuda")
            if cdev == "cpu":
                local = local.cpu()
            shard.allgatherv_tuples(local, world, rank, dist, cdev)
            hits += n
            nwin += 1
        return hits, nwin

    rank_hits = []          # per-rank hit counts of the last step (N>1, --gather counts): the shard imbalance

    def step():
        if world > 1 and args.gather == "tuples":
            return step_gather_tuples()
        hits, nwin = batch.run()      # synchronises the library's stream at the end of the pass
        if world > 1 and args.gather != "none":
            cnt = torch.tensor([hits], dtype=torch.int64, device=cdev)
            allc = [torch.zeros_like(cnt) for _ in range(world)]
            dist.all_gather(allc, cnt)
            rank_hits[:] = [int(x.item()) for x in allc]
        return hits, nwin

    for _ in range(args.warmup):
        step()
    r.stats_reset()
    barrier()
    t_start = time.time()
    hits = nwin = 0
    for _ in range(args.steps):
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

    pcie = None
    if args.d2h and rank == 0:
        batch.run_to_host()                      # warm the pinned staging ring
        t = time.time()
        batch.run_to_host()
        pcie = my_topics / (time.time() - t)

    if rank != 0:
        batch.close(); r.close()
        if world > 1:
            dist.destroy_process_group()
        return

    K = args.steps
    value = total_topics * K / elapsed
    # ---- roofline of the dominant kernel (expand): algorithmic bytes / HIP-event time
    exp_s = st["expand_ms"] / 1e3
    walk_s = st["walk_ms"] / 1e3
    exp_gbs = st["alg_bytes_expand"] / exp_s / 1e9 if exp_s > 0 else 0.0
    walk_gbs = st["alg_bytes_walk"] / walk_s / 1e9 if walk_s > 0 else 0.0
    dominant = "expand_kernel" if exp_s >= walk_s else "walk_kernel"
    ach = exp_gbs if dominant == "expand_kernel" else walk_gbs
    launches = st["expand_launches"] if dominant == "expand_kernel" else st["walk_launches"]
    dom_s = exp_s if dominant == "expand_kernel" else walk_s
    # HBM traffic per launch from the PMC passes of profiles/collect_pmc.sh (FETCH_SIZE / WRITE_SIZE are
    # collected in separate rocprofv3 runs, so they cannot be measured inside this process): measured
    # bytes per hit x hits per launch.  null until a calibration file for this kernel exists.
    traffic = None
    try:
        cal = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[dominant]
        if dominant == "expand_kernel" and launches:
            traffic = int((cal["write_bytes_per_hit"] + cal["fetch_bytes_per_hit"]) * st["hits"] / launches)
        elif launches:
            traffic = int(cal["bytes_per_topic"] * st["topics"] / launches)
    except (OSError, KeyError, ValueError):
        pass
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                "launches": int(launches), "avg_launch_ms": round(dom_s * 1e3 / max(1, launches), 4),
                "alg_bytes_per_launch": int((st["alg_bytes_expand"] if dominant == "expand_kernel" else st["alg_bytes_walk"]) / max(1, launches)),
                "walk_GBps": round(walk_gbs, 1), "expand_GBps": round(exp_gbs, 1)}

    # ---- CPU baseline: the oracle ("port"), bounded sample of the same workload, this host's cores (N=1 only)
    cpu = None
    if args.cpu_sample != 0 and world == 1:
        from oracle import oracle as orc
        cores = args.cpu_threads or os.cpu_count() or 1
        t = time.time()
        if retain:
            o = orc.RetainTree()
            o.insert_bulk(blob, offs)
        else:
            o = orc.DefaultRouter()
            o.add_bulk(blob, offs, client, qos)
        log(f"cpu_baseline: oracle table built in {time.time() - t:.1f}s; timing on {cores} threads", 0)
        hits_per_topic = max(1.0, total_hits / max(1, total_topics))
        # bounded sample: about 20 s of CPU work at the oracle's measured rates
        budget_hits = (3.5e7 if retain else 3.0e9) * cores / 256
        n_s = args.cpu_sample if args.cpu_sample > 0 else int(min(n_pub, max(200 if retain else 2000, budget_hits / hits_per_topic)))
        sb, so = shard.take(tb, to, np.arange(n_s))
        sec, ost = o.match_timed(sb, so, cores)
        cpu = {"value": round(n_s / sec, 1), "unit": "publish-topic matches/s", "cores": cores, "kind": "port",
               "sample": f"first {n_s} publish topics of the same batch against the full {n_sub}-subscription table, "
                         f"{ost['hits']} hits, {sec:.2f}s wall",
               "hits_per_s": round(ost["hits"] / sec, 1)}
        # SURVEY 8(d) also asks for the single-thread figure: a ~3 s prefix of the same sample on one core
        if cores > 1:
            n1 = max(20, min(n_s, int(n_s / cores * 0.15)))
            s1b, s1o = shard.take(tb, to, np.arange(n1))
            sec1, ost1 = o.match_timed(s1b, s1o, 1)
            cpu["single_thread"] = {"value": round(n1 / sec1, 1), "hits_per_s": round(ost1["hits"] / sec1, 1),
                                    "sample": f"first {n1} topics, {sec1:.2f}s wall"}

    out = {
        "metric": f"publish-topic matches/sec with delivery stage (config {cfg}, scale {args.scale}, v5 fraction {args.deliver})"
                  if (args.deliver >= 0 and not retain and world == 1) else
                  "publish-topic matches/sec @10M subs" if cfg in (3, 4) and args.scale == 1.0 else
                  (f"retained-path SUBSCRIBE-filter matches/sec (config 5, scale {args.scale})" if retain else
                   f"publish-topic matches/sec (config {cfg}, scale {args.scale})"),
        "value": round(value, 1), "unit": "SUBSCRIBE-filter matches/s" if retain else "publish-topic matches/s",
        "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / K, 3),
        "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[{cfg - 1}]: {n_sub} subscriptions (p_plus/level {c['p_plus']}, p_hash {c['p_hash']}, "
                               f"Zipf tokens s=1.1, Zipf clients s=1.0), {n_pub} publish topics, seeds 0x{wl.SUB_SEED + cfg:X}/0x{wl.PUB_SEED + cfg:X}",
                   "subscriptions": n_sub, "publishes": n_pub, "sharding": f"hash of the first {shard.KEY_LEVELS} levels x{world}" if world > 1 else "none",
                   "gather": args.gather if world > 1 else "n/a", "windows_per_step": int(nwin)},
        "hits_per_step": int(total_hits), "hits_per_s": round(total_hits * K / elapsed, 1),
        "mean_hits_per_topic": round(total_hits / max(1, total_topics), 2),
        "mean_visited_nodes_per_topic": round(st["visited_nodes"] / max(1, st["topics"]), 2),
        "kernel_ms_per_step": {"walk": round(st["walk_ms"] / K, 3), "scan_compact_tiles": round(st["scan_ms"] / K, 3),
                               "expand": round(st["expand_ms"] / K, 3)},
        "alg_bytes_per_step": {"walk": int(st["alg_bytes_walk"] / K), "expand": int(st["alg_bytes_expand"] / K)},
        "table": {"filters": int(st0["n_filters"]), "subs": int(st0["n_subs"]), "trie_nodes": int(st0["n_nodes"]),
                  "hbm_bytes": int(st0["table_bytes_device"]), "host_build_s": round(build_s, 1)},
        "roofline": roofline, "cpu_baseline": cpu,
    }
    try:
        if world > 1 and rank_hits and sum(rank_hits) > 0:
            out["shard_hits"] = rank_hits
            out["shard_imbalance_max_over_mean"] = round(max(rank_hits) * len(rank_hits) / sum(rank_hits), 3)
    except Exception as e:      # reporting only: never fail the bench line over it
        log(f"shard imbalance not reported: {e}", 0)
    if args.deliver >= 0 and not retain and world == 1:
        out["delivery_stage"] = {"v5_fraction": args.deliver, "dedup_ms_per_step": round(st["dedup_ms"] / K, 3),
                                 "dedup_candidates_per_step": int(st["dedup_candidates"] / K),
                                 "dedup_launches_per_step": int(st["dedup_launches"] / K)}
    if pcie is not None:
        out["pcie_inclusive_matches_per_s"] = round(pcie, 1)
    print(json.dumps(out), flush=True)
    batch.close(); r.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
