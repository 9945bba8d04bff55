# GPU session r6e: (1) delivery + retained-path parity on the device (8-byte hits; positions / packed reads of the retained path);
# (2) config 5, one table: tuples from 8-byte entries vs from the packed side array vs positions; (3) rocprofv3 kernel stats of the config-5 pass
set -u
O=$PWD/gpurun_out/r6e
mkdir -p $O
( time timeout 1500 python3 -m pytest tests/test_deliver_parity.py tests/test_retain_parity.py tests/test_retain_tiers.py tests/test_formats_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 ) 2> $O/pytest_time.txt; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest.log | tail -3
( time timeout 900 python3 bench.py --config 5 --time-format tuple,positions --steps 5 --warmup 1 --ab-env RGR_RETAIN_PACKED_READS=0,RGR_RETAIN_PACKED_READS=1 > $O/ab_retain_id_source.jsonl 2> $O/ab_retain_id_source.err ) 2> $O/ab_time.txt; echo "ab rc=$?"
cut -c1-600 $O/ab_retain_id_source.jsonl
