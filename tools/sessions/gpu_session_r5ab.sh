# GPU session r5ab: a second sample of the driver's own command on the round's final tree (ISA-identical to what r5s measured)
set -u
O=$PWD/gpurun_out/r5ab
mkdir -p $O
( time timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err ) 2> $O/bench_driver_cmd_time.txt; echo "bench rc=$?"; tail -3 $O/bench_driver_cmd_time.txt
cp gpurun_out/bench_detail_n1.json $O/bench_detail_n1.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("$O/bench_driver_cmd.json").read().strip().splitlines()[-1])
print("default:", d["value"], d["ms_per_step"], {k: d["roofline"].get(k) for k in ("frac","traffic","avg_launch_ms","kernel")})
for f in d.get("compact_formats", []): print("   fmt", f.get("format"), f.get("value"))
for x in d.get("secondary", []): print("   sec", str(x.get("metric"))[:70], x.get("value"), (x.get("roofline") or {}).get("frac"), (x.get("roofline") or {}).get("kernel"), (x.get("parity_sample") or {}).get("ok"))
PY
