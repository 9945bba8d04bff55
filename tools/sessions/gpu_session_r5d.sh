# GPU session r5d: the v5 dedup with its loads in batches — tile pass over contiguous tile ranges (one coalesced read of the count words),
# topic pass with one round trip per item (dedup_topic_batch_kernel; RGR_DEDUP_BATCH=0 = the one-load-at-a-time kernel).  Parity first.
set -u
O=gpurun_out/r5d
mkdir -p $O
( timeout 600 python -m pytest tests/test_deliver_parity.py tests/test_properties_gpu.py -k "deliver or delivery or dedup" -m gpu -q -x --timeout 300 > $O/pytest_deliver.log 2>&1; echo "pytest rc=$?" >> $O/pytest_deliver.log ); tail -4 $O/pytest_deliver.log | cut -c1-300
timeout 400 python bench.py --time-format deliver --steps 3 --warmup 1 --ab-env "RGR_DEDUP_BATCH=0,RGR_DEDUP_BATCH=1" > $O/ab_dedup_batch.jsonl 2> $O/ab_dedup_batch.err; echo "deliver rc=$?"
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r5d/ab_*.jsonl")):
    print(f)
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if "ab_check" in d: print("  CHECK", d["format"], d["ab_check"], "ok" if d["ok"] else "MISMATCH")
        else: print("  ", d["format"], d.get("env"), d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
