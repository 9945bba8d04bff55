"""Where a SMALL host-staged delivery pass spends its time (the pass GpuShared's batcher issues: a few thousand publishes of config 2 per call).
rgr_match_batch (plain) and rgr_match_batch_deliver on the same topics, wall clock per call beside the library's own stage clocks (rgr_stats).
   python3 tools/deliver_pass_profile.py [publishes per call] [calls] [threads]
With threads > 1 the delivery pass is also timed from that many host threads at once (ctypes releases the GIL): what one pass costs when the
batcher keeps several in flight."""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rmqtt_amd import capi, workload as wl

n_call = int(sys.argv[1]) if len(sys.argv) > 1 else 2600
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n_threads = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n_subs = 1_000_000
blob, offs, client, qos = wl.gen_subs(n_subs, wl.SUB_SEED + 2, p_plus=0.028, p_hash=0.0, n_clients=n_subs // 10)
rng = np.random.default_rng(7)
flags = np.where(rng.random(n_subs) < 0.1, capi.RGR_SUB_V5 | np.where(rng.random(n_subs) < 0.5, capi.RGR_SUB_NO_LOCAL, 0) | np.where(rng.random(n_subs) < 0.5, capi.RGR_SUB_RAP, 0), 0).astype(np.uint8)
r = capi.Router(device=0)
r.subscribe_bulk(blob, offs, sub_ids=np.arange(n_subs, dtype=np.uint32), qos=qos.astype(np.uint8), flags=flags)
r.sub_attrs_bulk(client.astype(np.uint32), client.astype(np.uint32))
r.commit()
tb, to = wl.gen_topics(200_000, wl.PUB_SEED + 2)[:2]
def batch(k):
    lo = (k * n_call) % (200_000 - n_call)
    o = to[lo:lo + n_call + 1]
    return tb[int(o[0]):int(o[-1])].copy(), (o - o[0]).astype(np.uint64)
pa = np.zeros(n_call, dtype=capi.PUBLISH_ATTR_DTYPE)
pa["from_id"] = rng.choice(client.astype(np.uint32), size=n_call)
pa["qos_retain"] = rng.integers(0, 3, size=n_call) | (rng.integers(0, 2, size=n_call) << 2)
KEYS = ("tokenize_ms", "h2d_ms", "walk_ms", "scan_ms", "expand_ms", "dedup_ms", "d2h_ms")
for name in ("plain", "deliver", "plain", "deliver"):
    bs = [batch(k) for k in range(calls)]
    r.stats_reset()
    t0 = time.perf_counter()
    hits = 0
    for b, o in bs:
        res = r.match_batch(b, o) if name == "plain" else r.match_batch_deliver(b, o, pa)
        hits += len(res["tuples"])
    wall = (time.perf_counter() - t0) / calls * 1e3
    st = r.stats()
    print(f"{name:8s} {n_call} publishes per call: {wall:.3f} ms per call (python included); library clocks per call: " +
          ", ".join(f"{k} {st[k] / calls:.3f}" for k in KEYS) + f"; launches per call: walk {st['walk_launches'] / calls:.1f} expand {st['expand_launches'] / calls:.1f} dedup {st['dedup_launches'] / calls:.1f}; hits per call {st['hits'] / calls:.0f}")

if n_threads > 1:
    import threading
    for nt in range(1, n_threads + 1):
        bs = [batch(k) for k in range(calls)]
        walls = [0.0] * nt
        def work(i):
            t0 = time.perf_counter()
            for b, o in bs[i::nt]:
                r.match_batch_deliver(b, o, pa)
            walls[i] = (time.perf_counter() - t0) / max(1, len(bs[i::nt])) * 1e3
        th = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        tot = time.perf_counter() - t0
        print(f"{nt} host threads: {sum(walls) / nt:.3f} ms per delivery pass of {n_call} publishes seen by a thread, {calls / tot:.0f} passes/s = {calls * n_call / tot / 1e6:.2f} M publishes/s in all")
