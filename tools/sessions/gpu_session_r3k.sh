# GPU session r3k: the driver-style default bench line again, after the parity strata were capped by the oracle's budget
# (r3i: 18 m 52 s, 14 of them in config 5's uncapped last-window stratum), with the delivery-stage parity in its secondary record
set -u
O=gpurun_out/r3k
mkdir -p $O
( time timeout 1000 python bench.py > $O/bench_default_final.json 2> $O/bench_default_final.err ) 2> $O/bench_default_time.txt; tail -3 $O/bench_default_time.txt
python - <<PY
import json
d=json.load(open("$O/bench_default_final.json"))
print("default:", d["value"], d["ms_per_step"], d["kernel_ms_per_step"], {k: d["roofline"].get(k) for k in ("frac","frac_stores_only","alg_frac","avg_launch_ms","traffic")}, d["parity_sample"]["ok"], d["parity_sample"]["formats"], d["parity_sample"]["strata"])
for s in d.get("secondary", []):
    print("  sec:", s.get("metric","?")[:70], s.get("value"), s.get("ms_per_step"), (s.get("parity_sample") or {}), s.get("delivery_stage"))
PY
grep "bench +" $O/bench_default_final.err | cut -c1-120
