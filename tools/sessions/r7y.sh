# GPU session r7y: (1) the table's lock lets a waiting writer through (new readers yield), (2) a publish whose pass a removal overtook joins another batch
# instead of the host path — host-router tests (incl. the churn tests), then Shared::forwards at config 2 with and without a subscriber thread beside it
set -u
O=$PWD/gpurun_out/r7y
mkdir -p $O
( timeout 1500 python3 -m pytest tests/test_host_router.py -m gpu -q -x > $O/pytest_host_router.log 2>&1 ); echo "pytest rc=$?"; grep -E "passed|failed|error|assert" $O/pytest_host_router.log | tail -5
timeout 1200 python3 bench.py --router-e2e --e2e-configs 2 --e2e-legs forwards --e2e-churn > $O/e2e_churn.jsonl 2> $O/e2e_churn.err; echo "rc=$?"
grep -E "e2e config" $O/e2e_churn.err | cut -c1-1500
timeout 1200 python3 bench.py --router-e2e --e2e-configs 2 --e2e-legs forwards,matches > $O/e2e.jsonl 2> $O/e2e.err; echo "rc=$?"
grep -E "e2e config" $O/e2e.err | cut -c1-600
