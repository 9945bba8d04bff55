# GPU session r7w: the full-size property tests (RMQTT_TEST_SCALE=1.0: BASELINE sizes) on the last tree of the round
set -u
O=$PWD/gpurun_out/r7w
mkdir -p $O
( time RMQTT_TEST_SCALE=1.0 timeout 3000 python3 -m pytest tests/test_properties_gpu.py -m gpu -x -q > $O/pytest_full_size.log 2>&1 ) 2> $O/time.txt; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_full_size.log | tail -3; tail -3 $O/time.txt
