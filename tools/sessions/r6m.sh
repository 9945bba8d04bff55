# GPU session r6m: (1) the group tests incl. the retained path over the group; (2) rocprofv3 kernel stats of the round's tree: the headline program,
# the delivery pass (8-byte hits, walk order), the config-5 pass; (3) the full-size property tests (RMQTT_TEST_SCALE=1.0)
set -u
O=$PWD/gpurun_out/r6m
R=$PWD
mkdir -p $O
( time timeout 1200 python3 -m pytest tests/test_group_gpu.py tests/test_distributed.py -m gpu -x -q > $O/pytest_group.log 2>&1 ) 2> $O/pytest_group_time.txt; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_group.log | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --steps 5 --warmup 2 --no-secondary --no-pmc --cpu-sample 0 --no-d2h --no-formats > $O/bench_config3_under_rocprofv3.json 2> $O/bench_config3_under_rocprofv3.err; echo "prof1 rc=$?"
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp "$f" $O/bench_config3_tuple_kernel_stats_rocprofv3.csv 2>/dev/null; head -6 "$f" | cut -c1-170; rm -rf $O/trace
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace2 -o t -- python3 $R/bench.py --time-format deliver8 --steps 3 --warmup 1 > $O/deliver8_under_rocprofv3.jsonl 2> $O/deliver8_under_rocprofv3.err; echo "prof2 rc=$?"
f=$(find $O/trace2 -name "*kernel_stats.csv" | head -1); cp "$f" $O/deliver8_walk_order_kernel_stats_rocprofv3.csv 2>/dev/null; head -8 "$f" | cut -c1-170; rm -rf $O/trace2
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace3 -o t -- python3 $R/bench.py --config 5 --time-format tuple --steps 5 --warmup 1 > $O/config5_under_rocprofv3.jsonl 2> $O/config5_under_rocprofv3.err; echo "prof3 rc=$?"
f=$(find $O/trace3 -name "*kernel_stats.csv" | head -1); cp "$f" $O/config5_tuple_kernel_stats_rocprofv3.csv 2>/dev/null; head -8 "$f" | cut -c1-170; rm -rf $O/trace3
cd $R
( time RMQTT_TEST_SCALE=1.0 timeout 1500 python3 -m pytest tests/test_properties_gpu.py -m gpu -x -q > $O/pytest_properties_fullscale.log 2>&1 ) 2> $O/pytest_properties_time.txt; echo "props rc=$?"; grep -E "passed|failed|error" $O/pytest_properties_fullscale.log | tail -3
