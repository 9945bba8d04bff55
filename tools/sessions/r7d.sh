# GPU session r7d: Shared::forwards through the boundary after (1) the publishers' owner ids looked up by Id instead of by a joined string key
# (1.1 ms of a 1.5 ms pass of 2 763 publishes) and (2) one shared-lock acquisition per worker run instead of one per publish.  Host-router tests first.
set -u
O=$PWD/gpurun_out/r7d
mkdir -p $O
( timeout 1200 python3 -m pytest tests/test_host_router.py tests/test_rust_ffi_consistency.py -q -x > $O/pytest_host_router.log 2>&1 ); echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_host_router.log | tail -3
timeout 1500 python3 bench.py --router-e2e --e2e-configs 2,3 --e2e-legs forwards,matches --e2e-sweep > $O/e2e.jsonl 2> $O/e2e.err; echo "rc=$?"
grep -E "e2e config" $O/e2e.err | cut -c1-700
python3 - <<PY
import json
for ln in open("$O/e2e.jsonl"):
    try: d = json.loads(ln)
    except Exception: continue
    for x in (d if isinstance(d, list) else [d]):
        if isinstance(x, dict) and "metric" in x: print(x["metric"][:80], x.get("value"), x.get("vs_cpu_port"), (x.get("cpu_reference_port") or {}).get("value"))
PY
