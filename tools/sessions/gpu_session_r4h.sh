# GPU session r4h: the tree as it will be judged — whole GPU suite, smoke, the driver-style default bench line (all secondaries + PMC),
# rocprofv3 kernel stats of the headline and of the compact formats, the 2-rank self-launched line
set -u
O=gpurun_out/r4h
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -4 $O/pytest_gpu.log | cut -c1-300
( timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1 ); tail -1 $O/smoke.log
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
grep "bench +" $O/bench_default.err | cut -c1-150 | tail -30
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_tuple -o t -- python $GRAFT_REPO_ROOT/bench.py --time-format tuple --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/$O/prof_tuple.json 2> $GRAFT_REPO_ROOT/$O/prof_tuple.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_ids24 -o t -- python $GRAFT_REPO_ROOT/bench.py --time-format ids24 --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/$O/prof_ids24.json 2> $GRAFT_REPO_ROOT/$O/prof_ids24.err
cd $GRAFT_REPO_ROOT
for n in tuple ids24; do f=$(find $O/prof_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$n.csv && head -6 "$f" | cut -c1-230; python tools/trace_gaps.py $O/prof_$n > $O/trace_gaps_$n.txt 2>&1; head -3 $O/trace_gaps_$n.txt; done
find $O -name "*kernel_trace.csv" -delete
for f in packed ids24; do timeout 300 python bench.py --time-format $f --steps 5 --warmup 2 >> $O/formats_final.jsonl 2>> $O/formats.err; done; cut -c1-330 $O/formats_final.jsonl
( timeout 600 python bench.py --gpus 2 --dist-backend gloo --scale 0.1 --steps 3 --warmup 1 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err ); echo "2rank rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_2rank_gloo.json').read().strip().splitlines()[-1]); print(d['value'], d['n_gpus'], d['parity_sample']['ok'], d['parity_sample']['topics'], d['roofline'].get('frac'), d.get('shard_hits'))"
du -sh $O
