# GPU session r3f: whole suite (no -x), delivery stage with the topic pass keeping its first tile in registers,
# Router::matches end to end after the batcher got per-request wake-ups and the rows ref-counted strings
set -u
O=gpurun_out/r3f
mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -8 $O/pytest_gpu.log | cut -c1-300
( timeout 400 python bench.py --config 3 --steps 3 --warmup 1 --no-pmc --no-secondary --no-d2h --deliver 0.1 > $O/bench_cfg3_deliver0.1.json 2> $O/bench_cfg3_deliver0.1.err ); tail -1 $O/bench_cfg3_deliver0.1.err | cut -c1-300
( timeout 900 python bench.py --router-e2e > $O/router_e2e.jsonl 2> $O/router_e2e.err ); tail -3 $O/router_e2e.err | cut -c1-600
python - <<PY
import json
try:
    d=json.load(open("$O/bench_cfg3_deliver0.1.json"))
    print("deliver0.1", d["value"], d["ms_per_step"], d.get("kernel_ms_per_step"), d.get("delivery_stage"))
except Exception as e:
    print("unreadable", e)
for l in open("$O/router_e2e.jsonl"):
    d=json.loads(l); print(d["metric"][-10:], [(g["mode"][:7], g["value"], g["latency_us"]) for g in d["gpu"]], d["cpu_reference_port"]["value"], d["vs_cpu_port"])
PY
