# GPU session r6b: exempt runs of the v5 dedup (kernels.hpp kExemptMinRun): the delivery parity worlds on the device, then the A/B on one table at
# full size (config 3, 10 % v5) with the whole-window parity check under the faster value, then kernel stats of a delivery pass
set -u
O=$PWD/gpurun_out/r6b
mkdir -p $O
( time timeout 1200 python3 -m pytest tests/test_deliver_parity.py -m gpu -x -q > $O/pytest_deliver.log 2>&1 ) 2> $O/pytest_deliver_time.txt; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_deliver.log | tail -3
( time timeout 1200 python3 bench.py --time-format deliver --steps 3 --warmup 1 --ab-env RGR_DELIVER_EXEMPT=0,RGR_DELIVER_EXEMPT=1 > $O/ab_deliver_exempt.jsonl 2> $O/ab_deliver_exempt.err ) 2> $O/ab_time.txt; echo "ab rc=$?"
cat $O/ab_deliver_exempt.jsonl | cut -c1-900
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o deliver -- python3 $GRAFT_REPO_ROOT/bench.py --time-format deliver --steps 2 --warmup 1 > $O/prof_run.jsonl 2> $O/prof_run.err; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/deliver_kernel_stats.csv
head -12 $O/deliver_kernel_stats.csv | cut -c1-200
rm -rf $O/prof
