# GPU session r6f: Shared::forwards through the delivery stage (rmqtt_amd/host/gpu_shared.*): the three-way parity test, then the end-to-end leg at
# configs 2 and 3 beside the oracle's forwards pass on the same host
set -u
O=$PWD/gpurun_out/r6f
mkdir -p $O
( time timeout 1200 python3 -m pytest tests/test_host_router.py tests/test_retain_parity.py -m gpu -x -q > $O/pytest.log 2>&1 ) 2> $O/pytest_time.txt; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest.log | tail -3
( time timeout 1800 python3 bench.py --router-e2e --e2e-legs forwards --e2e-configs 2,3 --e2e-sweep > $O/forwards_e2e.jsonl 2> $O/forwards_e2e.err ) 2> $O/e2e_time.txt; echo "e2e rc=$?"
cut -c1-1800 $O/forwards_e2e.jsonl
tail -3 $O/e2e_time.txt
