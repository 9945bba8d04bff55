# GPU session r4i: final tree — GPU suite, smoke, the driver-style default line, the compact formats after the short first chunk
set -u
O=gpurun_out/r4i
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -4 $O/pytest_gpu.log | cut -c1-300
( timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1 ); tail -1 $O/smoke.log
for f in ids24 packed tuple; do timeout 300 python bench.py --time-format $f --steps 5 --warmup 2 >> $O/formats_final.jsonl 2>> $O/formats.err; done; cut -c1-330 $O/formats_final.jsonl
( time timeout 1100 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default_time.txt; echo "bench rc=$?"; tail -3 $O/bench_default_time.txt
python - <<PY
import json
try:
    d=json.load(open("$O/bench_default.json"))
    print("default:", d["value"], d["ms_per_step"], d["kernel_ms_per_step"], {k: d["roofline"].get(k) for k in ("frac","frac_stores_only","alg_frac","avg_launch_ms","traffic")})
    ps=d["parity_sample"]; print("  parity:", ps["ok"], ps["topics"], ps["formats"], ps["oracle_s"], ps["gpu_digest_s"], ps["oracle_cross_check"]["ok"])
    print("  cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
    for f in d.get("compact_formats", []): print("   fmt", f.get("format","")[:14], f.get("value"), f.get("ms_per_step"), f.get("expand_avg_launch_ms"), f.get("expand_store_GBps"))
    for s in d.get("secondary", []):
        ps=s.get("parity_sample") or {}
        print("  sec:", s.get("metric","?")[:72], s.get("value"), s.get("ms_per_step"), "parity", ps.get("ok"), ps.get("topics"), "frac", (s.get("roofline") or {}).get("frac"), "cpu", (s.get("cpu_baseline") or {}).get("value"), s.get("pcie_inclusive_ranges",{}).get("matches_per_s"), s.get("value_async_submit"), (s.get("cpu_reference_port") or {}).get("value"))
except Exception as e: print("parse failed", e)
PY
du -sh $O
