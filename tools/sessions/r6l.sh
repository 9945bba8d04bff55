# GPU session r6l: the driver's own command on the round's tree (python3 bench.py --gpus 1 --steps 20 --warmup 5), then the whole GPU suite
set -u
O=$PWD/gpurun_out/r6l
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
