# GPU session r2e: TileRec fast path (parity first), then geometry A/B of the tuple kernel, then the default bench
set -u
O=gpurun_out/r2e
mkdir -p $O
( timeout 900 python -m pytest tests/test_formats_gpu.py tests/test_parity.py tests/test_deliver_parity.py tests/test_retain_parity.py tests/test_properties_gpu.py -m gpu -q -x > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log )
tail -3 $O/pytest_subset.log
B="--steps 5 --warmup 2 --config 3 --no-pmc --no-secondary --cpu-sample 0 --no-d2h"
( timeout 400 python bench.py $B > $O/bench_512x4.json 2> $O/bench_512x4.err )
( RGR_EXTRA_FLAGS="-DRGR_EXPAND_THREADS=256 -DRGR_EXPAND_PER_THREAD=8" python -c "from rmqtt_amd import build as b; b.build_gpu(force=True)" > /dev/null 2>&1
  timeout 400 python bench.py $B > $O/bench_256x8.json 2> $O/bench_256x8.err )
( RGR_EXTRA_FLAGS="-DRGR_EXPAND_THREADS=1024 -DRGR_EXPAND_PER_THREAD=2" python -c "from rmqtt_amd import build as b; b.build_gpu(force=True)" > /dev/null 2>&1
  timeout 400 python bench.py $B > $O/bench_1024x2.json 2> $O/bench_1024x2.err )
python -c "from rmqtt_amd import build as b; b.build_gpu(force=True)" > /dev/null 2>&1
python - <<PY
import json
for g in ("512x4","256x8","1024x2"):
    try:
        d=json.load(open("$O/bench_%s.json" % g))
        print(g, d["value"], d["roofline"]["avg_launch_ms"], [(f["format"][:6], f["value"], f["expand_avg_launch_ms"]) for f in d["compact_formats"]])
    except Exception as e:
        print(g, "failed", e)
PY
