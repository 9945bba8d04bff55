# GPU session r6r: r6q + the candidate statistic per block of the tile pass instead of 2 048 atomics on one address
# of its list requested together.  Delivery parity, then the delivery pass timed (8-byte hits, walk order) with its whole-window parity
set -u
O=$PWD/gpurun_out/r6r
mkdir -p $O
( time timeout 1200 python3 -m pytest tests/test_deliver_parity.py -m gpu -x -q > $O/pytest.log 2>&1 ) 2> $O/pytest_time.txt; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest.log | tail -3
timeout 900 python3 bench.py --time-format deliver8 --steps 3 --warmup 1 --ab-env RGR_DEDUP_PROBE=3,RGR_DEDUP_PROBE=0 > $O/deliver8.jsonl 2> $O/deliver8.err; echo "rc=$?"
python3 - <<PY
import json
for ln in open("$O/deliver8.jsonl"):
    d = json.loads(ln)
    if "ab_check" in d: print("ab_check", d["ok"], d["delivery_parity"]["mismatching_words"]); continue
    print(d["env"], d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace2 -o t -- python3 $GRAFT_REPO_ROOT/bench.py --time-format deliver8 --steps 3 --warmup 1 > $O/deliver8_under_rocprofv3.jsonl 2> $O/deliver8_under_rocprofv3.err; echo "prof rc=$?"
f=$(find $O/trace2 -name "*kernel_stats.csv" | head -1); cp "$f" $O/deliver8_kernel_stats_rocprofv3.csv 2>/dev/null; head -5 "$f" | cut -c1-60,300-420; rm -rf $O/trace2
