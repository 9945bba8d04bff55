# GPU session r7h: r7g + the tokeniser scans reading their bytes through aligned 8-byte words (one load per eight bytes) — parity (tokeniser edge
# cases, golden vectors, PUBLISH packets, maximum sizes, retained path), the small delivery pass again, tokenisation of 10 M topics
set -u
O=$PWD/gpurun_out/r7h
mkdir -p $O
( timeout 1500 python3 -m pytest tests/test_parity.py tests/test_golden_fixtures.py tests/test_publish_packets.py tests/test_max_sizes.py tests/test_retain_parity.py tests/test_properties_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 ); echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest.log | tail -3
timeout 900 python3 tools/deliver_pass_profile.py 2600 300 > $O/profile_2600.txt 2> $O/profile_2600.err; echo "rc=$?"; cut -c1-330 $O/profile_2600.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $GRAFT_REPO_ROOT/tools/deliver_pass_profile.py 2600 200 > $O/profile_under_rocprofv3.txt 2> $O/profile_under_rocprofv3.err; echo "prof rc=$?"
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/deliver_pass_kernel_stats.csv && head -6 "$f" | cut -c1-80,150-260; rm -rf $O/trace
cd $GRAFT_REPO_ROOT
timeout 900 python3 bench.py --gpus 1 --steps 3 --warmup 1 --no-pmc --no-secondary --no-formats --cpu-sample 0 > $O/bench_quick.json 2> $O/bench_quick.err; echo "bench rc=$?"
python3 -c "
import json; d=json.loads(open('$O/bench_quick.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('h2d_inclusive'), d.get('tokenize'))" 2>&1 | cut -c1-600
grep -i "tokeni" $O/bench_quick.err | tail -5 | cut -c1-300
