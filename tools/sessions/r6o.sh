# GPU session r6o: count_big / compact_big with eight descriptors per thread and step: retained-path and router parity, then config 5 timed (digests of
# a whole pass against the positions form, which shares the descriptors) and config 3 for regression
set -u
O=$PWD/gpurun_out/r6o
mkdir -p $O
( time timeout 1200 python3 -m pytest tests/test_retain_parity.py tests/test_retain_tiers.py tests/test_parity.py tests/test_max_sizes.py tests/test_properties_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 ) 2> $O/pytest_time.txt; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest.log | tail -3
timeout 600 python3 bench.py --config 5 --time-format tuple,positions --steps 5 --warmup 1 > $O/config5.jsonl 2> $O/config5.err; echo "c5 rc=$?"
timeout 900 python3 bench.py --time-format tuple --steps 5 --warmup 2 > $O/config3.jsonl 2> $O/config3.err; echo "c3 rc=$?"
python3 - <<PY
import json
for f in ("config5", "config3"):
    for ln in open(f"$O/{f}.jsonl"):
        d = json.loads(ln)
        print(f, d["format"], d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"])
PY
