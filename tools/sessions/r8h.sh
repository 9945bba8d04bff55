# GPU session r8h: the permanent lane-mapping parity test (all 64 lanes / every 8th / one item per wave on small batches)
set -u
O=$PWD/gpurun_out/r8h
mkdir -p $O
( timeout 1200 python3 -m pytest tests/test_parity.py -m gpu -q -x > $O/pytest_parity.log 2>&1 ); echo "pytest rc=$?"; grep -E "passed|failed|error|assert" $O/pytest_parity.log | tail -4
