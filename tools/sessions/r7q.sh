# GPU session r7q: two more host round trips out of a micro-batch pass (per-topic offsets fetched with the totals; publish attributes staged in pinned
# memory, nobody waits for their upload) — delivery / host-router / group / publish-packet parity, small passes, the consumers at config 2
set -u
O=$PWD/gpurun_out/r7q
mkdir -p $O
( timeout 2400 python3 -m pytest tests/test_parity.py tests/test_publish_packets.py tests/test_group_gpu.py tests/test_host_router.py tests/test_deliver_parity.py tests/test_formats_gpu.py tests/test_retain_parity.py tests/test_bench_line.py -m gpu -x -q > $O/pytest.log 2>&1 ); echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest.log | tail -3
for n in 300 2600 20000; do timeout 600 python3 tools/deliver_pass_profile.py $n 300 > $O/profile_${n}.txt 2> $O/profile_${n}.err; echo "$n rc=$?"; tail -2 $O/profile_${n}.txt | cut -c1-250; done
timeout 1500 python3 bench.py --router-e2e --e2e-configs 2 --e2e-legs forwards,matches > $O/e2e.jsonl 2> $O/e2e.err; echo "rc=$?"
grep -E "e2e config" $O/e2e.err | cut -c1-520
