# GPU session r5m: the v5 dedup's load balance — topic pass with items taken from a counter (RGR_DEDUP_PROBE=7), tile pass with more blocks
set -u
O=gpurun_out/r5m
mkdir -p $O
( RGR_DEDUP_PROBE=7 RGR_DEDUP_TILE_GRID=8192 timeout 600 python -m pytest tests/test_deliver_parity.py tests/test_properties_gpu.py -k "deliver or delivery or dedup" -m gpu -q -x --timeout 300 > $O/pytest_deliver_probe7.log 2>&1; echo "pytest rc=$?" >> $O/pytest_deliver_probe7.log ); tail -3 $O/pytest_deliver_probe7.log | cut -c1-300
timeout 700 python bench.py --time-format deliver --steps 3 --warmup 1 --ab-env "X=0,RGR_DEDUP_PROBE=7,RGR_DEDUP_TILE_GRID=8192,RGR_DEDUP_TILE_GRID=32768,RGR_DEDUP_PROBE=7+RGR_DEDUP_TILE_GRID=8192" > $O/ab_dedup_balance.jsonl 2> $O/ab_dedup_balance.err; echo "deliver rc=$?"
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r5m/ab_*.jsonl")):
    print(f)
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if "ab_check" in d: print("  CHECK", d["format"], d["ab_check"], "ok" if d["ok"] else "MISMATCH", d.get("delivery_parity", {}).get("mismatching_words"))
        else: print("  ", d["format"], d.get("env"), d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
