# GPU session r4m: the full-size property test (10 M x 10 M, config 3; 5 M x 1 M, config 5) on the final tree
set -u
O=gpurun_out/r4m
mkdir -p $O
( RMQTT_TEST_SCALE=1.0 timeout 800 python -m pytest tests/test_properties_gpu.py -m gpu -q --timeout 700 > $O/pytest_properties_fullscale.log 2>&1; echo "pytest rc=$?" >> $O/pytest_properties_fullscale.log ); tail -4 $O/pytest_properties_fullscale.log | cut -c1-300
