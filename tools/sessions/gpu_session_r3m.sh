# GPU session r3m: dedup_topic_kernel with 256-thread blocks (5 per CU) against 512 (4 per CU)
set -u
O=gpurun_out/r3m
mkdir -p $O
for v in 512 256; do
  export RGR_EXTRA_FLAGS="-DRGR_DEDUP_TOPIC_THREADS=$v"
  python -c "from rmqtt_amd import build as b; b.build_gpu(force=True)" > $O/build_$v.log 2>&1
  if [ $v = 256 ]; then ( timeout 300 python -m pytest tests/test_deliver_parity.py -m gpu -q --timeout 200 > $O/pytest_deliver_256.log 2>&1 ); tail -1 $O/pytest_deliver_256.log; fi
  ( timeout 300 python bench.py --config 3 --steps 3 --warmup 1 --no-pmc --no-secondary --no-d2h --no-parity --deliver 0.1 > $O/deliver_$v.json 2> $O/deliver_$v.err )
  python - <<PY
import json
try:
    d=json.load(open("$O/deliver_$v.json")); k=d["kernel_ms_per_step"]; w=d["config"]["windows_per_step"]
    print("threads $v", "expand ms/window", round(k["expand"]/w,3), "dedup ms/window", round(d["delivery_stage"]["dedup_ms_per_step"]/w,3), "matches/s", d["value"])
except Exception as e:
    print("$v", "failed", e)
PY
done 2>&1 | tee $O/dedup_topic_threads_ab.txt
