# GPU session r2f: full suite on the prefetch / TileRec / runs-format tree, then tuple-kernel geometry with and without chunk prefetch
set -u
O=gpurun_out/r2f
mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log )
tail -5 $O/pytest_gpu.log
B="--steps 5 --warmup 2 --config 3 --no-pmc --no-secondary --cpu-sample 0 --no-d2h"
( timeout 400 python bench.py $B > $O/bench_512x4.json 2> $O/bench_512x4.err )
( RGR_NO_PREFETCH=1 timeout 400 python bench.py $B > $O/bench_512x4_noprefetch.json 2> $O/bench_512x4_noprefetch.err )
for G in "1024 2" "1024 4"; do
  set -- $G
  ( RGR_EXTRA_FLAGS="-DRGR_EXPAND_THREADS=$1 -DRGR_EXPAND_PER_THREAD=$2" python -c "from rmqtt_amd import build as b; b.build_gpu(force=True)" > /dev/null 2>&1
    timeout 400 python bench.py $B > $O/bench_$1x$2.json 2> $O/bench_$1x$2.err )
done
python -c "from rmqtt_amd import build as b; b.build_gpu(force=True)" > /dev/null 2>&1
python - <<PY
import json
for g in ("512x4","512x4_noprefetch","1024x2","1024x4"):
    try:
        d=json.load(open("$O/bench_%s.json" % g))
        print(g, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["kernel_ms_per_step"], [(f["format"][:6], f["value"], f["expand_avg_launch_ms"]) for f in d["compact_formats"]])
    except Exception as e:
        print(g, "failed", e)
PY
