# GPU session r5o: topic pass of the v5 dedup with a two-level Bloom filter in front of a small exact table (RGR_DEDUP_PROBE=7)
set -u
O=gpurun_out/r5o
mkdir -p $O
( RGR_DEDUP_PROBE=7 timeout 600 python -m pytest tests/test_deliver_parity.py tests/test_properties_gpu.py -k "deliver or delivery or dedup" -m gpu -q -x --timeout 300 > $O/pytest_deliver_probe7.log 2>&1; echo "pytest rc=$?" >> $O/pytest_deliver_probe7.log ); tail -3 $O/pytest_deliver_probe7.log | cut -c1-300
timeout 700 python bench.py --time-format deliver --steps 3 --warmup 1 --ab-env "X=0,RGR_DEDUP_PROBE=7" > $O/ab_dedup_bloom.jsonl 2> $O/ab_dedup_bloom.err; echo "deliver rc=$?"
timeout 400 python bench.py --time-format deliver --deliver 0.3 --steps 2 --warmup 1 --ab-env "X=0,RGR_DEDUP_PROBE=7" > $O/ab_dedup_bloom_v5frac0.3.jsonl 2> $O/ab_dedup_bloom_v5frac0.3.err; echo "deliver 0.3 rc=$?"
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r5o/ab_*.jsonl")):
    print(f)
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if "ab_check" in d: print("  CHECK", d["format"], d["ab_check"], "ok" if d["ok"] else "MISMATCH", d.get("delivery_parity", {}).get("mismatching_words"), d.get("delivery_parity", {}).get("v5_duplicates_flagged"))
        else: print("  ", d["format"], d.get("env"), d.get("v5_frac"), d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
