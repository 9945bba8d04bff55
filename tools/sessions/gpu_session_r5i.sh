# GPU session r5i: lean delivery expansion, 512 x 4 (trimmed v5 loop) vs 256 x 8 positions per lane, and 2^30-hit delivery windows under it
set -u
O=gpurun_out/r5i
mkdir -p $O
( RGR_DELIVER_LEAN=2 timeout 600 python -m pytest tests/test_deliver_parity.py tests/test_properties_gpu.py -k "deliver or delivery or dedup" -m gpu -q -x --timeout 300 > $O/pytest_deliver_lean2.log 2>&1; echo "pytest rc=$?" >> $O/pytest_deliver_lean2.log ); tail -3 $O/pytest_deliver_lean2.log | cut -c1-300
( RGR_DELIVER_LEAN=1 timeout 600 python -m pytest tests/test_deliver_parity.py tests/test_properties_gpu.py -k "deliver or delivery or dedup" -m gpu -q -x --timeout 300 > $O/pytest_deliver_lean1.log 2>&1; echo "pytest rc=$?" >> $O/pytest_deliver_lean1.log ); tail -3 $O/pytest_deliver_lean1.log | cut -c1-300
timeout 700 python bench.py --time-format deliver --steps 3 --warmup 1 --ab-env "X=0,RGR_DELIVER_LEAN=1,RGR_DELIVER_LEAN=2,RGR_DELIVER_LEAN=1+RGR_DELIVER_WINDOW_HITS=1073741824,RGR_DELIVER_LEAN=2+RGR_DELIVER_WINDOW_HITS=1073741824" > $O/ab_deliver_lean_512x4_256x8.jsonl 2> $O/ab_deliver_lean_512x4_256x8.err; echo "deliver rc=$?"
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r5i/ab_*.jsonl")):
    print(f)
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if "ab_check" in d: print("  CHECK", d["format"], d["ab_check"], "ok" if d["ok"] else "MISMATCH", d.get("delivery_parity", {}).get("mismatching_words"))
        else: print("  ", d["format"], d.get("env"), d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
tail -3 $O/ab_deliver_lean_512x4_256x8.err | cut -c1-400
