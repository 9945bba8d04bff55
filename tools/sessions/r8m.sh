# GPU session r8m: host-router tests after the readers' back-off behind a long writer; the consumers at config 2 beside the subscriber thread once more
set -u
O=$PWD/gpurun_out/r8m
mkdir -p $O
( timeout 1500 python3 -m pytest tests/test_host_router.py -m gpu -q -x > $O/pytest_host_router.log 2>&1 ); echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_host_router.log | tail -2
timeout 1200 python3 bench.py --router-e2e --e2e-configs 2 --e2e-legs forwards,matches --e2e-churn > $O/e2e_churn.jsonl 2> $O/e2e_churn.err; echo "rc=$?"
grep -E "e2e config.*async|forwards e2e config 2: \{" $O/e2e_churn.err | cut -c1-330
