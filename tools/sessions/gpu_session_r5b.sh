# GPU session r5b: the v5 dedup of window k on a second stream beside the expansion of window k + 1 (RGR_DELIVER_OVERLAP, default on in
# device-resident delivery passes), with r5a's winners as defaults (RGR_PREP_BATCH, RGR_DELIVER_EARLY).  Parity first, then the A/B.
set -u
O=gpurun_out/r5b
mkdir -p $O
( timeout 600 python -m pytest tests/test_properties_gpu.py tests/test_deliver_parity.py tests/test_formats_gpu.py -m gpu -q -x --timeout 300 > $O/pytest_deliver_formats_properties.log 2>&1; echo "pytest rc=$?" >> $O/pytest_deliver_formats_properties.log ); tail -5 $O/pytest_deliver_formats_properties.log | cut -c1-300
timeout 400 python bench.py --time-format deliver --steps 3 --warmup 1 --ab-env "RGR_DELIVER_OVERLAP=0,RGR_DELIVER_OVERLAP=1,RGR_DELIVER_OVERLAP=1+RGR_DELIVER_WINDOW_HITS=134217728,RGR_DELIVER_OVERLAP=1+RGR_DELIVER_WINDOW_HITS=536870912" > $O/ab_deliver_overlap.jsonl 2> $O/ab_deliver_overlap.err; echo "deliver rc=$?"
tail -3 $O/ab_deliver_overlap.err | cut -c1-400
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r5b/ab_*.jsonl")):
    print(f)
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if "ab_check" in d: print("  CHECK", d["format"], d["ab_check"], "ok" if d["ok"] else "MISMATCH")
        else: print("  ", d["format"], d.get("env"), d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
