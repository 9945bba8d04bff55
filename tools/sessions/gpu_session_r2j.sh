# GPU session r2j: overflow re-walk with one wave per block + run descriptors kept by the count step — full suite, default bench, A/B
set -u
O=gpurun_out/r2j
mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -q --durations=3 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log )
tail -3 $O/pytest_gpu.log
( timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/bench_default.err )
tail -1 $O/bench_default.err
B="--steps 5 --warmup 2 --config 3 --no-pmc --no-secondary --cpu-sample 0 --no-d2h"
( RGR_NO_SLOT_DESC=1 timeout 400 python bench.py $B > $O/bench_no_slot_desc.json 2> $O/bench_no_slot_desc.err )
python - <<PY
import json
for g in ("bench_default","bench_no_slot_desc"):
    try:
        d=json.load(open("$O/%s.json" % g)); r=d["roofline"]
        print(g, d["value"], d["ms_per_step"], d["kernel_ms_per_step"], r["avg_launch_ms"], r.get("frac"), d.get("overflow_topics_per_step"), [(f["format"][:6], f["value"]) for f in d["compact_formats"]], [(s["value"], s["parity_sample"]["ok"]) for s in d.get("secondary",[])], d.get("parity_sample",{}).get("ok"))
    except Exception as e: print(g, "failed", e)
PY
