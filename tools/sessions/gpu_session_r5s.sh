# GPU session r5s: the round's final tree — whole GPU suite, smoke, the driver's own command (roofline from the PMC children again),
# rocprofv3 kernel trace of the headline program and of the delivery pass
set -u
O=$PWD/gpurun_out/r5s
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -4 $O/pytest_gpu.log | cut -c1-300
( timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1 ); tail -1 $O/smoke.log
( time timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err ) 2> $O/bench_driver_cmd_time.txt; echo "bench rc=$?"; tail -3 $O/bench_driver_cmd_time.txt
wc -c $O/bench_driver_cmd.json; wc -l $O/bench_driver_cmd.json
cp gpurun_out/bench_detail_n1.json $O/bench_detail_n1.json 2>/dev/null
grep "pmc:" $O/bench_driver_cmd.err | cut -c1-300
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_driver_cmd.json").read().strip().splitlines()[-1])
    print("default:", d["value"], d["ms_per_step"], d.get("roofline"))
    for f in d.get("compact_formats", []): print("   fmt", f.get("format"), f.get("value"), f.get("ms_per_step"), f.get("expand_avg_launch_ms"))
    for x in d.get("secondary", []): print("   sec", str(x.get("metric"))[:80], x.get("value"), x.get("ms_per_step"), (x.get("roofline") or {}).get("frac"), (x.get("parity_sample") or {}).get("ok"))
except Exception as e: print("parse failed", e)
PY
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 5 --warmup 2 --no-secondary --no-pmc --cpu-sample 0 --no-d2h --no-formats > $O/bench_under_rocprofv3.json 2> $O/bench_under_rocprofv3.err
echo "trace rc=$?"
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp "$f" $O/bench_config3_kernel_stats_rocprofv3.csv 2>/dev/null; head -6 "$f" | cut -c1-160; rm -rf $O/trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace2 -o t -- python $R/bench.py --time-format deliver --steps 3 --warmup 1 > $O/deliver_under_rocprofv3.jsonl 2> $O/deliver_under_rocprofv3.err
echo "trace2 rc=$?"
f=$(find $O/trace2 -name "*kernel_stats.csv" | head -1); cp "$f" $O/deliver_kernel_stats_rocprofv3.csv 2>/dev/null; head -8 "$f" | cut -c1-160; rm -rf $O/trace2
cat $O/deliver_under_rocprofv3.jsonl | cut -c1-400
du -sh $O
