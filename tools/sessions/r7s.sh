# GPU session r7s: the publisher's owner id looked up where the publish is submitted (epoch-checked hint) instead of inside the pass — host-router tests,
# Shared::forwards through the boundary at config 2 over the batcher's shapes, config 3 once
set -u
O=$PWD/gpurun_out/r7s
mkdir -p $O
( timeout 1200 python3 -m pytest tests/test_host_router.py -m gpu -q -x > $O/pytest_host_router.log 2>&1 ); echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_host_router.log | tail -3
timeout 1500 python3 bench.py --router-e2e --e2e-configs 2 --e2e-legs forwards --e2e-sweep > $O/e2e.jsonl 2> $O/e2e.err; echo "rc=$?"
grep -E "e2e config" $O/e2e.err | cut -c1-560
timeout 1500 python3 bench.py --router-e2e --e2e-configs 3 --e2e-legs forwards > $O/e2e3.jsonl 2> $O/e2e3.err; echo "rc=$?"
grep -E "e2e config" $O/e2e3.err | cut -c1-560
python3 - <<PY
import json
for f in ("$O/e2e.jsonl", "$O/e2e3.jsonl"):
    for ln in open(f):
        try: d = json.loads(ln)
        except Exception: continue
        for x in (d if isinstance(d, list) else [d]):
            if isinstance(x, dict) and "metric" in x: print(x["metric"][:80], x.get("value"), x.get("vs_cpu_port"), (x.get("cpu_reference_port") or {}).get("value"))
PY
