# GPU session r5q: delivery windows of 2^27 / 2^28 / 2^29 hits under the lean expansion; the N = 2 self-launch (gloo, one GPU, 1/10 scale) on this tree
set -u
O=gpurun_out/r5q
mkdir -p $O
timeout 700 python bench.py --time-format deliver --steps 3 --warmup 1 --ab-env "X=0,RGR_DELIVER_WINDOW_HITS=134217728,RGR_DELIVER_WINDOW_HITS=536870912" > $O/ab_deliver_window_hits.jsonl 2> $O/ab_deliver_window_hits.err; echo "deliver rc=$?"
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r5q/ab_*.jsonl")):
    print(f)
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if "ab_check" in d: print("  CHECK", d["format"], d["ab_check"], "ok" if d["ok"] else "MISMATCH", d.get("delivery_parity", {}).get("mismatching_words"))
        else: print("  ", d["format"], d.get("env"), d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
( time timeout 900 python bench.py --gpus 2 --dist-backend gloo --scale 0.1 --steps 5 --warmup 2 --no-secondary > $O/bench_2rank_gloo_scale0.1.json 2> $O/bench_2rank_gloo_scale0.1.err ) 2> $O/bench_2rank_time.txt; echo "2-rank rc=$?"; tail -3 $O/bench_2rank_time.txt
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r5q/bench_2rank_gloo_scale0.1.json").read().strip().splitlines()[-1])
    print(d["value"], d["n_gpus"], d["ms_per_step"], d["config"].get("rccl_ranks"), d.get("shard_hits"), (d.get("parity_sample") or {}).get("ok"), (d.get("parity_sample") or {}).get("topics"), (d.get("roofline") or {}).get("frac"))
except Exception as e: print("parse failed", e)
PY
