# GPU session r8a: r7z + a delivery pass on a dirty table commits under the exclusive lock and then RUNS under the shared one (as filter passes do) —
# host-router tests, Shared::forwards at config 2 with and without a subscriber thread beside it
set -u
O=$PWD/gpurun_out/r8a
mkdir -p $O
( timeout 1500 python3 -m pytest tests/test_host_router.py -m gpu -q -x > $O/pytest_host_router.log 2>&1 ); echo "pytest rc=$?"; grep -E "passed|failed|error|assert" $O/pytest_host_router.log | tail -5
timeout 1200 python3 bench.py --router-e2e --e2e-configs 2 --e2e-legs forwards --e2e-churn > $O/e2e_churn.jsonl 2> $O/e2e_churn.err; echo "rc=$?"
grep -E "e2e config" $O/e2e_churn.err | cut -c1-1500
timeout 1200 python3 bench.py --router-e2e --e2e-configs 2 --e2e-legs forwards,matches > $O/e2e.jsonl 2> $O/e2e.err; echo "rc=$?"
grep -E "e2e config" $O/e2e.err | cut -c1-600
