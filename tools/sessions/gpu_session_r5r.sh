# GPU session r5r: tile pass + classification in one launch (no per-window memset): parity worlds, then delivery windows of 2^28 / 2^27 / 2^26 hits
set -u
O=gpurun_out/r5r
mkdir -p $O
( timeout 600 python -m pytest tests/test_deliver_parity.py tests/test_properties_gpu.py tests/test_host_router.py -k "deliver or delivery or dedup or router" -m gpu -q -x --timeout 300 > $O/pytest_deliver_fused_classify.log 2>&1; echo "pytest rc=$?" >> $O/pytest_deliver_fused_classify.log ); tail -3 $O/pytest_deliver_fused_classify.log | cut -c1-300
timeout 700 python bench.py --time-format deliver --steps 3 --warmup 1 --ab-env "X=0,RGR_DELIVER_WINDOW_HITS=134217728,RGR_DELIVER_WINDOW_HITS=67108864" > $O/ab_deliver_window_hits_fused_classify.jsonl 2> $O/ab_deliver_window_hits_fused_classify.err; echo "deliver rc=$?"
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r5r/ab_*.jsonl")):
    print(f)
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if "ab_check" in d: print("  CHECK", d["format"], d["ab_check"], "ok" if d["ok"] else "MISMATCH", d.get("delivery_parity", {}).get("mismatching_words"))
        else: print("  ", d["format"], d.get("env"), d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
