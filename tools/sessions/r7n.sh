# GPU session r7n: the sparse walk with 8 and 4 walks per wave (RGR_WALK_LANE_SHIFT=3, 4) beside 16 (=2), small delivery pass
set -u
O=$PWD/gpurun_out/r7n
mkdir -p $O
for sh in 2 3 4 2 3; do RGR_WALK_LANE_SHIFT=$sh timeout 600 python3 tools/deliver_pass_profile.py 2600 300 > $O/profile_2600_shift$sh.txt 2> $O/profile_2600_shift$sh.err; echo "shift $sh rc=$?"; tail -1 $O/profile_2600_shift$sh.txt | cut -c1-250; done
( RGR_WALK_LANE_SHIFT=3 timeout 1200 python3 -m pytest tests/test_parity.py tests/test_golden_fixtures.py tests/test_hypothesis_parity.py -m gpu -x -q > $O/pytest_shift3.log 2>&1 ); echo "pytest shift3 rc=$?"; grep -E "passed|failed|error" $O/pytest_shift3.log | tail -3
