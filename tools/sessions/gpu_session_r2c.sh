# GPU session r2c: full GPU test suite, commit-latency probe, host-out latency (C calls), walk A/B lab, compact-kernel geometry sweep
set -u
O=gpurun_out/r2c
mkdir -p $O
R=$(pwd)
( timeout 1200 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log )
tail -6 $O/pytest_gpu.log
( RGR_COMMIT_PROFILE=1 timeout 600 python tools/latency.py 3 1.0 > $O/latency_cfg3_full.txt 2> $O/latency_cfg3_full.err ); cat $O/latency_cfg3_full.txt; grep "\[commit\]" $O/latency_cfg3_full.err | tail -12
( timeout 200 tools/walk_lab 1000000 1000000 0.028 0.0 5 > $O/walk_lab.txt 2>&1; timeout 200 tools/walk_lab 1000000 1000000 0.2 0.0 5 >> $O/walk_lab.txt 2>&1; timeout 300 tools/walk_lab 1000000 1000000 0.2 0.1 5 >> $O/walk_lab.txt 2>&1 ); cat $O/walk_lab.txt
for T in 128 256 512; do
  ( RGR_EXTRA_FLAGS="-DRGR_COMPACT_THREADS=$T" python -c "from rmqtt_amd import build as b; b.build_gpu(force=True)" > /dev/null 2>&1
    timeout 400 python bench.py --steps 5 --warmup 2 --config 3 --scale 0.5 --no-pmc --no-secondary --cpu-sample 0 --no-d2h > $O/bench_compact_threads_$T.json 2> $O/bench_compact_threads_$T.err )
  python - <<PY
import json
d=json.load(open("$O/bench_compact_threads_$T.json"))
print("compact threads $T:", d["value"], [ (f["format"][:6], f["value"], f["expand_avg_launch_ms"]) for f in d["compact_formats"]])
PY
done
python -c "from rmqtt_amd import build as b; b.build_gpu(force=True)" > /dev/null 2>&1
