# GPU session r7k: micro-batch passes with fewer host round trips — one shard of a group answers without shard keys / copies (as the filter form always
# did), count -> scan -> fill of the tokeniser back to back on bound-sized arrays, no synchronisation after the blob's upload.  Parity, the small pass,
# the two consumers through the boundary at config 2.
set -u
O=$PWD/gpurun_out/r7k
mkdir -p $O
( timeout 2400 python3 -m pytest tests/test_parity.py tests/test_golden_fixtures.py tests/test_publish_packets.py tests/test_max_sizes.py tests/test_group_gpu.py tests/test_host_router.py tests/test_deliver_parity.py tests/test_hypothesis_parity.py -m gpu -x -q > $O/pytest.log 2>&1 ); echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest.log | tail -3
timeout 900 python3 tools/deliver_pass_profile.py 2600 300 > $O/profile_2600.txt 2> $O/profile_2600.err; echo "rc=$?"; cut -c1-330 $O/profile_2600.txt
timeout 1500 python3 bench.py --router-e2e --e2e-configs 2 --e2e-legs forwards,matches --e2e-sweep > $O/e2e.jsonl 2> $O/e2e.err; echo "rc=$?"
grep -E "e2e config" $O/e2e.err | cut -c1-620
