# GPU session r5aa: topic-major slots — the walk collects eight matched filter ids in registers and writes whole 32-byte sectors of the topic's
# row (RGR_SLOTS_TOPIC_MAJOR=1): whole GPU suite under the switch, then the A/B on one table (ids24, runs, tuple)
set -u
O=gpurun_out/r5aa
mkdir -p $O
( RGR_SLOTS_TOPIC_MAJOR=1 timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 > $O/pytest_gpu_slots_topic_major.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_slots_topic_major.log ); tail -4 $O/pytest_gpu_slots_topic_major.log | cut -c1-300
timeout 700 python bench.py --time-format ids24,runs,tuple --steps 4 --warmup 1 --ab-env "X=0,RGR_SLOTS_TOPIC_MAJOR=1" > $O/ab_slots_topic_major.jsonl 2> $O/ab.err; echo "rc=$?"
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r5aa/ab_*.jsonl")):
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        if "ab_check" in d: print("  CHECK", d["format"], d["ab_check"], "ok" if d["ok"] else "MISMATCH")
        else: print("  ", d["format"], d.get("env"), d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"])
PY
