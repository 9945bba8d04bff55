# GPU session r5v: the full-size property tests (config 3 at 10 M x 10 M, config 5 at 5 M x 1 M, the delivery properties) on the round's final tree
set -u
O=gpurun_out/r5v
mkdir -p $O
( RMQTT_TEST_SCALE=1.0 timeout 1200 python -m pytest tests/test_properties_gpu.py -m gpu -q --timeout 900 > $O/pytest_properties_fullscale.log 2>&1; echo "pytest rc=$?" >> $O/pytest_properties_fullscale.log ); tail -4 $O/pytest_properties_fullscale.log | cut -c1-300
