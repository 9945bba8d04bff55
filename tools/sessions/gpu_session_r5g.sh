# GPU session r5g: the delivery expansion that compacts every wave's v5 hits (RGR_DELIVER_LEAN=1) and the topic pass with the flat probe
# loop (RGR_DEDUP_PROBE=7) — parity worlds under both switches, then the A/B on one table with the whole-window delivery check
set -u
O=gpurun_out/r5g
mkdir -p $O
( RGR_DELIVER_LEAN=1 RGR_DEDUP_PROBE=7 timeout 600 python -m pytest tests/test_deliver_parity.py tests/test_properties_gpu.py -k "deliver or delivery or dedup" -m gpu -q -x --timeout 300 > $O/pytest_deliver_lean_probe7.log 2>&1; echo "pytest rc=$?" >> $O/pytest_deliver_lean_probe7.log ); tail -3 $O/pytest_deliver_lean_probe7.log | cut -c1-300
timeout 700 python bench.py --time-format deliver --steps 3 --warmup 1 --ab-env "X=0,RGR_DELIVER_LEAN=1,RGR_DEDUP_PROBE=7,RGR_DELIVER_LEAN=1+RGR_DEDUP_PROBE=7" > $O/ab_deliver_lean_probe7.jsonl 2> $O/ab_deliver_lean_probe7.err; echo "deliver rc=$?"
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r5g/ab_*.jsonl")):
    print(f)
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if "ab_check" in d: print("  CHECK", d["format"], d["ab_check"], "ok" if d["ok"] else "MISMATCH", d.get("delivery_parity", {}).get("mismatching_words"))
        else: print("  ", d["format"], d.get("env"), d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
tail -5 $O/ab_deliver_lean_probe7.err | cut -c1-400
