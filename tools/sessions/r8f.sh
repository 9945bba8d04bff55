# GPU session r8f: Shared::forwards at config 3 with the slab slots and relations of a publish's hits prefetched ahead in the completion loop
set -u
O=$PWD/gpurun_out/r8f
mkdir -p $O
( timeout 900 python3 -m pytest tests/test_host_router.py -m gpu -q -x > $O/pytest_host_router.log 2>&1 ); echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_host_router.log | tail -2
timeout 1500 python3 bench.py --router-e2e --e2e-configs 3 --e2e-legs forwards > $O/e2e3.jsonl 2> $O/e2e3.err; echo "rc=$?"
grep -E "e2e config" $O/e2e3.err | cut -c1-700
python3 - <<PY
import json
for ln in open("$O/e2e3.jsonl"):
    try: d = json.loads(ln)
    except Exception: continue
    for x in (d if isinstance(d, list) else [d]):
        if isinstance(x, dict) and "metric" in x: print(x["metric"][:80], x.get("value"), x.get("vs_cpu_port"), (x.get("cpu_reference_port") or {}).get("value"))
PY
