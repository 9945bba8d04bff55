# GPU session r4o (4 GPU-minutes left): the tree with IDS24 on the lane-held kernel by default — the retained path (config 5: its
# "pairs" are ranges of the preorder value array) digested over every filter against the tile-per-block kernel, the format / retain GPU
# tests, smoke, and the rocprofv3 kernel stats of a default ids24 pass at config 3.
set -u
O=gpurun_out/r4o
mkdir -p $O
timeout 120 python bench.py --config 5 --time-format ids24 --steps 3 --warmup 1 --ab-env "RGR_COMPACT_LP=0,RGR_COMPACT_LP=1" > $O/ab_config5_ids24.jsonl 2> $O/ab_config5.err
echo "config5 rc=$?"; cut -c1-400 $O/ab_config5_ids24.jsonl; tail -2 $O/ab_config5.err | cut -c1-200
( timeout 150 python -m pytest tests/test_formats_gpu.py tests/test_retain_parity.py -q -x --timeout 120 > $O/pytest_formats_retain.log 2>&1; echo "pytest rc=$?" >> $O/pytest_formats_retain.log ); tail -3 $O/pytest_formats_retain.log | cut -c1-300
( timeout 100 python __graft_entry__.py smoke > $O/smoke.log 2>&1 ); tail -1 $O/smoke.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_ids24 -o t -- python $GRAFT_REPO_ROOT/bench.py --time-format ids24 --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/$O/prof_ids24.json 2> $GRAFT_REPO_ROOT/$O/prof_ids24.err
cd $GRAFT_REPO_ROOT
cut -c1-330 $O/prof_ids24.json
f=$(find $O/prof_ids24 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_ids24.csv && head -5 "$f" | cut -c1-200
python tools/trace_gaps.py $O/prof_ids24 > $O/trace_gaps_ids24.txt 2>&1; head -12 $O/trace_gaps_ids24.txt
find $O -name "*kernel_trace.csv" -delete
du -sh $O
