# GPU session r7u: host-router tests with persistent From objects (cached owner ids across the publisher's own subscribes / unsubscribes and recycled owner
# ids), then the consumers at config 2 with the default shapes
set -u
O=$PWD/gpurun_out/r7u
mkdir -p $O
( timeout 1200 python3 -m pytest tests/test_host_router.py -m gpu -q -x > $O/pytest_host_router.log 2>&1 ); echo "pytest rc=$?"; grep -E "passed|failed|error|assert" $O/pytest_host_router.log | tail -6
timeout 1500 python3 bench.py --router-e2e --e2e-configs 2 --e2e-legs forwards,matches > $O/e2e.jsonl 2> $O/e2e.err; echo "rc=$?"
grep -E "e2e config" $O/e2e.err | cut -c1-560
