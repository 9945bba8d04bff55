# GPU session r3j: A/B of the delivery stage — topic-pass tables sized from an upper bound of a topic's candidates (no per-pair counts, no
# per-topic atomics in the expansion; the default build) against exact counts (-DRGR_DEDUP_EXACT_COUNTS, round 3's state until r3i)
set -u
O=gpurun_out/r3j
mkdir -p $O
( timeout 600 python -m pytest tests/test_deliver_parity.py tests/test_host_router.py tests/test_properties_gpu.py -m gpu -q --timeout 300 > $O/pytest_deliver.log 2>&1 ); tail -3 $O/pytest_deliver.log | cut -c1-200
for v in BOUNDS EXACT; do
  if [ $v = BOUNDS ]; then export RGR_EXTRA_FLAGS=""; else export RGR_EXTRA_FLAGS="-DRGR_DEDUP_EXACT_COUNTS"; python -c "from rmqtt_amd import build as b; b.build_gpu(force=True)" > $O/build_$v.log 2>&1; fi
  ( timeout 300 python bench.py --config 3 --steps 3 --warmup 1 --no-pmc --no-secondary --no-d2h --deliver 0.1 > $O/deliver_$v.json 2> $O/deliver_$v.err )
  python - <<PY
import json
try:
    d=json.load(open("$O/deliver_$v.json")); k=d["kernel_ms_per_step"]; w=d["config"]["windows_per_step"]
    print("$v", "expand ms/window", round(k["expand"]/w,3), "dedup ms/window", round(d["delivery_stage"]["dedup_ms_per_step"]/w,3), "matches/s", d["value"])
except Exception as e:
    print("$v", "failed", e)
PY
done 2>&1 | tee $O/deliver_ab.txt
