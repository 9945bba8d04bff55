# GPU session r8k: the owner index's epoch per bucket of the Id's hash (a subscriber coming and going no longer sends every publisher back to the locked
# lookup) — host-router tests, the two consumers at config 2 with a subscriber thread beside them and without
set -u
O=$PWD/gpurun_out/r8k
mkdir -p $O
( timeout 1500 python3 -m pytest tests/test_host_router.py -m gpu -q -x > $O/pytest_host_router.log 2>&1 ); echo "pytest rc=$?"; grep -E "passed|failed|error|assert" $O/pytest_host_router.log | tail -5
timeout 1200 python3 bench.py --router-e2e --e2e-configs 2 --e2e-legs forwards,matches --e2e-churn > $O/e2e_churn.jsonl 2> $O/e2e_churn.err; echo "rc=$?"
grep -E "e2e config.*async|forwards e2e config 2: \{" $O/e2e_churn.err | cut -c1-1300
timeout 1200 python3 bench.py --router-e2e --e2e-configs 2 --e2e-legs forwards,matches > $O/e2e.jsonl 2> $O/e2e.err; echo "rc=$?"
grep -E "e2e config.*async|forwards e2e config 2: \{" $O/e2e.err | cut -c1-600
