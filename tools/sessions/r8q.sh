# GPU session r8q: the driver's own command on the FINAL tree of round 6 (python3 bench.py --gpus 1 --steps 20 --warmup 5), the whole GPU suite, kernel traces
set -u
O=$PWD/gpurun_out/r8q
mkdir -p $O
( time timeout 2400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err ) 2> $O/bench_driver_cmd_time.txt; echo "bench rc=$?"; tail -3 $O/bench_driver_cmd_time.txt
cp gpurun_out/bench_detail_n1.json $O/bench_detail_n1.json 2>/dev/null
python3 - <<PY
import json
d=json.loads(open("$O/bench_driver_cmd.json").read().strip().splitlines()[-1])
print("line bytes", len(open("$O/bench_driver_cmd.json").read().strip().splitlines()[-1]))
print("default:", d["value"], d["ms_per_step"], {k: d["roofline"].get(k) for k in ("frac","frac_stores_only","traffic","avg_launch_ms","kernel")}, d["config"].get("topic_order","")[:20])
print("  cpu", d.get("cpu_baseline")); print("  parity", d.get("parity_sample"))
for f in d.get("compact_formats", []): print("   fmt", f.get("format"), f.get("value"))
for x in d.get("secondary", []): print("   sec", str(x.get("metric"))[:90], x.get("value"), (x.get("roofline") or {}).get("frac"), (x.get("roofline") or {}).get("kernel"), (x.get("parity_sample") or {}).get("ok"), x.get("retain_positions"), x.get("delivery_stage"), x.get("vs_cpu_port"))
PY
( time timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 ) 2> $O/pytest_gpu_time.txt; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
# the kernel trace of the same command (short form), for the average launch durations the line's roofline is checked against
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 3 --warmup 1 --no-pmc --no-secondary --no-formats --no-d2h --cpu-sample 0 > $O/bench_under_rocprofv3.json 2> $O/bench_under_rocprofv3.err; echo "prof rc=$?"
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp "$f" $O/bench_kernel_stats_rocprofv3.csv 2>/dev/null; head -6 "$f" | cut -c1-70,300-420; rm -rf $O/trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace2 -o t -- python3 $GRAFT_REPO_ROOT/bench.py --time-format deliver8 --steps 3 --warmup 1 > $O/deliver8_under_rocprofv3.jsonl 2> $O/deliver8_under_rocprofv3.err; echo "prof2 rc=$?"
f=$(find $O/trace2 -name "*kernel_stats.csv" | head -1); cp "$f" $O/deliver8_kernel_stats_rocprofv3.csv 2>/dev/null; head -6 "$f" | cut -c1-70,300-420; rm -rf $O/trace2
