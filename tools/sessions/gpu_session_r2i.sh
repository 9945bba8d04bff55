# GPU session r2i: final validation of the committed tree — full GPU suite, smoke, driver-style default bench
set -u
O=gpurun_out/r2i
mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log )
tail -4 $O/pytest_gpu.log
( timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1 ); tail -2 $O/smoke.log
( timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/bench_default.err )
tail -2 $O/bench_default.err
python - <<PY
import json
d=json.load(open("$O/bench_default.json")); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["alg_frac"], d["parity_sample"]["ok"], [(f["format"][:6], f["value"]) for f in d["compact_formats"]], [(s["value"], s["parity_sample"]["ok"]) for s in d["secondary"]])
PY
