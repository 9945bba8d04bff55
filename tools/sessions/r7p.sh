# GPU session r7p: the tokeniser's two kernels also spread a small batch over more waves (same rule as the walk) — parity (rule, dense forced, one item
# per wave forced), small delivery passes of 300 / 2 600 / 20 000 publishes with the rule and dense
set -u
O=$PWD/gpurun_out/r7p
mkdir -p $O
( timeout 2400 python3 -m pytest tests/test_parity.py tests/test_golden_fixtures.py tests/test_publish_packets.py tests/test_max_sizes.py tests/test_retain_parity.py tests/test_host_router.py tests/test_hypothesis_parity.py -m gpu -x -q > $O/pytest.log 2>&1 ); echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest.log | tail -3
for sh in 0 6; do ( RGR_WALK_LANE_SHIFT=$sh timeout 1200 python3 -m pytest tests/test_parity.py tests/test_golden_fixtures.py tests/test_publish_packets.py tests/test_hypothesis_parity.py -m gpu -x -q > $O/pytest_shift$sh.log 2>&1 ); echo "pytest shift $sh rc=$?"; grep -E "passed|failed|error" $O/pytest_shift$sh.log | tail -2; done
for n in 300 2600 20000; do timeout 600 python3 tools/deliver_pass_profile.py $n 200 > $O/profile_${n}_auto.txt 2> $O/profile_${n}_auto.err; echo "auto $n rc=$?"; tail -1 $O/profile_${n}_auto.txt | cut -c1-250; RGR_WALK_LANE_SHIFT=0 timeout 600 python3 tools/deliver_pass_profile.py $n 200 > $O/profile_${n}_dense.txt 2> $O/profile_${n}_dense.err; tail -1 $O/profile_${n}_dense.txt | cut -c1-250; done
