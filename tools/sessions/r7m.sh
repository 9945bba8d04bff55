# GPU session r7m: small chunks walked by every 4th lane (16 walks per wave instead of 64; RGR_WALK_LANE_SHIFT overrides) — parity under both mappings,
# the small delivery pass with and without, the two consumers through the boundary at config 2
set -u
O=$PWD/gpurun_out/r7m
mkdir -p $O
( timeout 2400 python3 -m pytest tests/test_parity.py tests/test_golden_fixtures.py tests/test_publish_packets.py tests/test_max_sizes.py tests/test_group_gpu.py tests/test_host_router.py tests/test_deliver_parity.py tests/test_hypothesis_parity.py tests/test_formats_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 ); echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest.log | tail -3
( RGR_WALK_LANE_SHIFT=1 timeout 1200 python3 -m pytest tests/test_parity.py tests/test_golden_fixtures.py tests/test_hypothesis_parity.py -m gpu -x -q > $O/pytest_shift1.log 2>&1 ); echo "pytest shift1 rc=$?"; grep -E "passed|failed|error" $O/pytest_shift1.log | tail -3
for sh in 0 1 2; do RGR_WALK_LANE_SHIFT=$sh timeout 600 python3 tools/deliver_pass_profile.py 2600 300 > $O/profile_2600_shift$sh.txt 2> $O/profile_2600_shift$sh.err; echo "shift $sh rc=$?"; tail -1 $O/profile_2600_shift$sh.txt | cut -c1-330; done
timeout 1500 python3 bench.py --router-e2e --e2e-configs 2 --e2e-legs forwards,matches > $O/e2e.jsonl 2> $O/e2e.err; echo "rc=$?"
grep -E "e2e config" $O/e2e.err | cut -c1-620
