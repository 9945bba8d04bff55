# GPU session r5f: topic pass of the v5 dedup — double hashing (bit 0) and 16-byte table clears (bit 1) behind RGR_DEDUP_PROBE
set -u
O=gpurun_out/r5f
mkdir -p $O
( RGR_DEDUP_PROBE=7 timeout 600 python -m pytest tests/test_deliver_parity.py tests/test_properties_gpu.py -k "deliver or delivery or dedup" -m gpu -q -x --timeout 300 > $O/pytest_deliver_probe3.log 2>&1; echo "pytest rc=$?" >> $O/pytest_deliver_probe3.log ); tail -3 $O/pytest_deliver_probe3.log | cut -c1-300
timeout 500 python bench.py --time-format deliver --steps 3 --warmup 1 --ab-env "RGR_DEDUP_PROBE=3,RGR_DEDUP_PROBE=7" > $O/ab_dedup_probe.jsonl 2> $O/ab_dedup_probe.err; echo "deliver rc=$?"
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r5f/ab_*.jsonl")):
    print(f)
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if "ab_check" in d: print("  CHECK", d["format"], d["ab_check"], "ok" if d["ok"] else "MISMATCH")
        else: print("  ", d["format"], d.get("env"), d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
