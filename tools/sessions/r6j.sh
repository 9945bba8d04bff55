# GPU session r6j: the delivery stage in walk order (8-byte hits): parity tests, then caller vs walk order on one full-size delivery pass each
set -u
O=$PWD/gpurun_out/r6j
mkdir -p $O
( time timeout 1200 python3 -m pytest tests/test_formats_gpu.py tests/test_deliver_parity.py -m gpu -x -q > $O/pytest.log 2>&1 ) 2> $O/pytest_time.txt; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest.log | tail -3
for ord in caller walk; do
  timeout 900 python3 bench.py --time-format deliver,deliver8 --steps 3 --warmup 1 --topic-order $ord > $O/deliver_order_$ord.jsonl 2> $O/deliver_order_$ord.err; echo "$ord rc=$?"
done
python3 - <<PY
import json
for f in ("deliver_order_caller", "deliver_order_walk"):
    for ln in open("$O/" + f + ".jsonl"):
        d = json.loads(ln)
        print(f, d["format"], d.get("topic_order"), d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
