# GPU session r2g: the final tree — full GPU suite, driver-style default bench, the same command under rocprofv3 --kernel-trace,
# delivery-stage regression runs, latency probes
set -u
O=gpurun_out/r2g
mkdir -p $O
R=$(pwd)
( timeout 1200 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log )
tail -4 $O/pytest_gpu.log
( timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/bench_default.err )
tail -2 $O/bench_default.err
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r2g -- python $R/bench.py --steps 5 --warmup 1 --no-pmc --no-secondary --cpu-sample 0 --no-d2h --no-formats > $R/$O/bench_under_rocprof.json 2> $R/$O/bench_under_rocprof.err )
python profiles/summarize_kernel_trace.py $O/prof > $O/kernel_stats.txt 2>&1; head -8 $O/kernel_stats.txt
B="--steps 5 --warmup 2 --config 3 --no-pmc --no-secondary --cpu-sample 0 --no-d2h"
( timeout 400 python bench.py $B --deliver 0 > $O/bench_deliver_v3only.json 2> $O/bench_deliver_v3only.err )
( timeout 400 python bench.py $B --deliver 0.1 > $O/bench_deliver_v5frac0.1.json 2> $O/bench_deliver_v5frac0.1.err )
python - <<PY
import json
for g in ("deliver_v3only","deliver_v5frac0.1"):
    try:
        d=json.load(open("$O/bench_%s.json" % g)); print(g, d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d.get("delivery_stage"))
    except Exception as e: print(g, "failed", e)
PY
( timeout 300 python tools/latency.py 2 1.0 > $O/latency_cfg2.txt 2>&1 ); head -6 $O/latency_cfg2.txt
