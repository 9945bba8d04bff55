# GPU session r6h: walk order inside the library (rgr_batch_set_order), two-level keys: the GPU test, then caller order vs walk order on full-size passes
# (config 3: tuples, runs, ids24, packed; config 2: tuples), each with the digest check of a whole pass against the other order's where the tool offers it
set -u
O=$PWD/gpurun_out/r6h
mkdir -p $O
( time timeout 1200 python3 -m pytest tests/test_formats_gpu.py tests/test_parity.py -m gpu -x -q > $O/pytest.log 2>&1 ) 2> $O/pytest_time.txt; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest.log | tail -3
for ord in caller walk; do
  timeout 900 python3 bench.py --time-format tuple,runs,ids24,packed --steps 5 --warmup 2 --topic-order $ord > $O/config3_formats_order_$ord.jsonl 2> $O/config3_formats_order_$ord.err; echo "c3 $ord rc=$?"
  timeout 600 python3 bench.py --config 2 --time-format tuple,runs --steps 20 --warmup 3 --topic-order $ord > $O/config2_order_$ord.jsonl 2> $O/config2_order_$ord.err; echo "c2 $ord rc=$?"
done
python3 - <<PY
import json
for f in ("config3_formats_order_caller", "config3_formats_order_walk", "config2_order_caller", "config2_order_walk"):
    for ln in open("$O/" + f + ".jsonl"):
        d = json.loads(ln)
        print(f, d["format"], d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"])
PY
