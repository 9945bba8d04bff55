# GPU session r8j: filter passes survive removals too (a slot keeps its filter's entry after its relation is removed; leases + limbo as for delivery passes) —
# host-router tests, the two consumers at config 2 with a subscriber thread beside them and without
set -u
O=$PWD/gpurun_out/r8j
mkdir -p $O
( timeout 1500 python3 -m pytest tests/test_host_router.py -m gpu -q -x > $O/pytest_host_router.log 2>&1 ); echo "pytest rc=$?"; grep -E "passed|failed|error|assert" $O/pytest_host_router.log | tail -5
timeout 1200 python3 bench.py --router-e2e --e2e-configs 2 --e2e-legs forwards,matches --e2e-churn > $O/e2e_churn.jsonl 2> $O/e2e_churn.err; echo "rc=$?"
grep -E "e2e config.*async|forwards e2e config 2: \{" $O/e2e_churn.err | cut -c1-1300
timeout 1200 python3 bench.py --router-e2e --e2e-configs 2 --e2e-legs matches > $O/e2e.jsonl 2> $O/e2e.err; echo "rc=$?"
grep -E "e2e config.*async" $O/e2e.err | cut -c1-600
