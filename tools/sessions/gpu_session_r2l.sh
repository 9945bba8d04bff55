# GPU session r2l: last check of the committed tree — GPU suite, smoke, the default bench exactly as the driver runs it
set -u
O=gpurun_out/r2l
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -3 $O/pytest_gpu.log
( timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1 ); tail -1 $O/smoke.log
( timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/bench_default.err ); tail -1 $O/bench_default.err
python - <<PY
import json
d=json.load(open("$O/bench_default.json")); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["frac"], d["parity_sample"]["ok"], [(f["format"][:6], f["value"]) for f in d["compact_formats"]], [(s["value"], s["roofline"]["frac"], s["parity_sample"]["ok"]) for s in d["secondary"]])
PY
