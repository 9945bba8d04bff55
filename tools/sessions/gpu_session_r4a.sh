# GPU session r4a: first of round 4 — (1) GPU suite on the new tree, (2) the self-launching N>1 bench on one GPU (2 ranks, gloo) with
# exhaustive parity against the unsharded oracle, (3) exhaustive parity at full size for configs 3 / 5 (timing of the fast oracle digests),
# (4) packed format vs window size, and a kernel trace of a packed pass (where the non-expansion time goes)
set -u
O=gpurun_out/r4a
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -3 $O/pytest_gpu.log | cut -c1-200
( time timeout 600 python bench.py --gpus 2 --dist-backend gloo --scale 0.1 --steps 3 --warmup 1 > $O/bench_2rank_gloo_selflaunch.json 2> $O/bench_2rank_gloo_selflaunch.err ) 2> $O/t2.txt; echo "2rank rc=$?"; tail -c 1500 $O/bench_2rank_gloo_selflaunch.json; tail -5 $O/bench_2rank_gloo_selflaunch.err | cut -c1-300
( time timeout 900 python bench.py --steps 5 --warmup 2 --no-secondary --no-pmc > $O/bench_cfg3_exhaustive.json 2> $O/bench_cfg3_exhaustive.err ) 2> $O/t3.txt; echo "cfg3 rc=$?"; grep real $O/t3.txt
python - <<PY
import json
try:
    d=json.load(open("$O/bench_cfg3_exhaustive.json"))
    print("cfg3:", d["value"], d["ms_per_step"], d["kernel_ms_per_step"], {k:v for k,v in d["parity_sample"].items() if k not in ("full_pass","digest")})
    for f in d.get("compact_formats", []): print("   ", f.get("format","")[:20], f.get("value"), f.get("ms_per_step"), f.get("expand_avg_launch_ms"), f.get("expand_store_GBps"))
except Exception as e: print("cfg3 parse failed", e)
PY
grep "bench +" $O/bench_cfg3_exhaustive.err | cut -c1-160
( time timeout 600 python bench.py --config 5 --steps 3 --warmup 1 --no-secondary --no-pmc --no-formats --no-d2h > $O/bench_cfg5_exhaustive.json 2> $O/bench_cfg5_exhaustive.err ) 2> $O/t5.txt; echo "cfg5 rc=$?"; grep real $O/t5.txt
python - <<PY
import json
try:
    d=json.load(open("$O/bench_cfg5_exhaustive.json"))
    print("cfg5:", d["value"], d["ms_per_step"], {k:v for k,v in d["parity_sample"].items() if k not in ("full_pass","digest")})
except Exception as e: print("cfg5 parse failed", e)
PY
grep "bench +" $O/bench_cfg5_exhaustive.err | tail -4 | cut -c1-200
for wh in 268435456 536870912 1073741824; do
  timeout 300 python bench.py --time-format packed --steps 5 --warmup 2 --window-hits $wh >> $O/packed_vs_window.jsonl 2>> $O/packed_vs_window.err
done
cat $O/packed_vs_window.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_packed -o t -- python $GRAFT_REPO_ROOT/bench.py --time-format packed --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$O/trace_packed.json 2> $GRAFT_REPO_ROOT/$O/trace_packed.err
cd $GRAFT_REPO_ROOT
python tools/trace_gaps.py $O/trace_packed | tee $O/trace_packed_gaps.txt
find $O/trace_packed -name "*.csv" -size +20M -delete
du -sh $O
