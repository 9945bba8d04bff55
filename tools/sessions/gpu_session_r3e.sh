# GPU session r3e: full suite (ctypes argtypes of the new entry points fixed), delivery stage after moving the candidate bookkeeping
# behind the stores + 16-byte candidates, Router::matches end to end through the host mirror beside the CPU port
set -u
O=gpurun_out/r3e
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -8 $O/pytest_gpu.log | cut -c1-300
( timeout 400 python bench.py --config 3 --steps 3 --warmup 1 --no-pmc --no-secondary --no-d2h --deliver 0.1 > $O/bench_cfg3_deliver0.1.json 2> $O/bench_cfg3_deliver0.1.err ); tail -1 $O/bench_cfg3_deliver0.1.err | cut -c1-300
( timeout 900 python bench.py --router-e2e > $O/router_e2e.jsonl 2> $O/router_e2e.err ); tail -6 $O/router_e2e.err | cut -c1-600
python - <<PY
import json
try:
    d=json.load(open("$O/bench_cfg3_deliver0.1.json"))
    print("deliver0.1", d["value"], d["ms_per_step"], d.get("kernel_ms_per_step"), d.get("delivery_stage"))
except Exception as e:
    print("unreadable", e)
PY
cat $O/router_e2e.jsonl | cut -c1-1500
