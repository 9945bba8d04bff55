# GPU session r3g: delivery variant with its attribute loads issued together with the subscriber entries (A/B against r3e: 1.01 ms per window)
set -u
O=gpurun_out/r3g
mkdir -p $O
( timeout 600 python -m pytest tests/test_deliver_parity.py tests/test_host_router.py -m gpu -q --timeout 300 > $O/pytest_deliver.log 2>&1 ); tail -3 $O/pytest_deliver.log | cut -c1-200
( timeout 400 python bench.py --config 3 --steps 3 --warmup 1 --no-pmc --no-secondary --no-d2h --deliver 0.1 > $O/bench_cfg3_deliver0.1.json 2> $O/bench_cfg3_deliver0.1.err )
python - <<PY
import json
d=json.load(open("$O/bench_cfg3_deliver0.1.json"))
print("deliver0.1", d["value"], d["ms_per_step"], d.get("kernel_ms_per_step"), d.get("delivery_stage"))
PY
