# GPU session r7o: the sparse walk chosen by chunk size (up to 2 048 waves; 2 600 topics -> 2 walks per wave), parity under the rule and with one walk
# per wave forced, small delivery pass (2 600 and 300 and 20 000 publishes per call), the consumers through the boundary at config 2
set -u
O=$PWD/gpurun_out/r7o
mkdir -p $O
( timeout 2400 python3 -m pytest tests/test_parity.py tests/test_golden_fixtures.py tests/test_publish_packets.py tests/test_max_sizes.py tests/test_group_gpu.py tests/test_host_router.py tests/test_deliver_parity.py tests/test_hypothesis_parity.py tests/test_formats_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 ); echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest.log | tail -3
( RGR_WALK_LANE_SHIFT=6 timeout 1200 python3 -m pytest tests/test_parity.py tests/test_golden_fixtures.py tests/test_hypothesis_parity.py -m gpu -x -q > $O/pytest_shift6.log 2>&1 ); echo "pytest shift6 rc=$?"; grep -E "passed|failed|error" $O/pytest_shift6.log | tail -3
for sh in 4 5 6; do RGR_WALK_LANE_SHIFT=$sh timeout 600 python3 tools/deliver_pass_profile.py 2600 300 > $O/profile_2600_shift$sh.txt 2> $O/profile_2600_shift$sh.err; echo "shift $sh rc=$?"; tail -1 $O/profile_2600_shift$sh.txt | cut -c1-250; done
for n in 300 2600 20000; do timeout 600 python3 tools/deliver_pass_profile.py $n 200 > $O/profile_${n}_auto.txt 2> $O/profile_${n}_auto.err; echo "auto $n rc=$?"; tail -1 $O/profile_${n}_auto.txt | cut -c1-250; RGR_WALK_LANE_SHIFT=0 timeout 600 python3 tools/deliver_pass_profile.py $n 200 > $O/profile_${n}_dense.txt 2> $O/profile_${n}_dense.err; tail -1 $O/profile_${n}_dense.txt | cut -c1-250; done
timeout 1500 python3 bench.py --router-e2e --e2e-configs 2 --e2e-legs forwards,matches > $O/e2e.jsonl 2> $O/e2e.err; echo "rc=$?"
grep -E "e2e config" $O/e2e.err | cut -c1-520
