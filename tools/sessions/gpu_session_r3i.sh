# GPU session r3i: the round's final evidence on the committed tree — whole GPU suite + smoke, the driver-style default bench line
# (PMC roofline, parity in every format, secondaries incl. the delivery stage), rocprofv3 kernel trace of a config-3 run, Router e2e
set -u
O=gpurun_out/r3i
mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -4 $O/pytest_gpu.log | cut -c1-200
( timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1 ); tail -1 $O/smoke.log
( time timeout 1500 python bench.py > $O/bench_default_final.json 2> $O/bench_default_final.err ) 2> $O/bench_default_time.txt; tail -3 $O/bench_default_time.txt
cd /tmp && export TMPDIR=/tmp
( timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o cfg3 -- python $GRAFT_REPO_ROOT/bench.py --config 3 --steps 5 --warmup 2 --no-pmc --no-secondary --no-d2h --no-formats --cpu-sample 0 > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_under_rocprof.err )
cd $GRAFT_REPO_ROOT
python profiles/summarize_kernel_trace.py $O/prof > $O/bench_config3_kernel_stats_rocprofv3.txt 2>&1; head -12 $O/bench_config3_kernel_stats_rocprofv3.txt
find $O/prof -type f -size +1M -delete
( timeout 900 python bench.py --router-e2e > $O/router_e2e.jsonl 2> $O/router_e2e.err )
python - <<PY
import json
d=json.load(open("$O/bench_default_final.json"))
print("default:", d["value"], d["ms_per_step"], d["kernel_ms_per_step"], {k: d["roofline"].get(k) for k in ("frac","frac_stores_only","alg_frac","avg_launch_ms","traffic")}, d["parity_sample"]["ok"], d["parity_sample"]["formats"])
for s in d.get("secondary", []):
    print("  sec:", s.get("metric","?")[:70], s.get("value"), s.get("ms_per_step"), (s.get("parity_sample") or {}).get("ok"), s.get("delivery_stage"))
print("  formats:", [(c["format"][:6], c["value"]) for c in d.get("compact_formats", [])], "cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
for l in open("$O/router_e2e.jsonl"):
    e=json.loads(l); print(e["metric"][-10:], [(g["mode"][:7], g["value"], g["latency_us"]) for g in e["gpu"]], e["cpu_reference_port"]["value"], e["vs_cpu_port"])
PY
