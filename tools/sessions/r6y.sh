# GPU session r6y: the lean delivery expansion (8-byte hits) reading its entries from the 4-byte delivery-packed side array — parity (all delivery
# tests; the table's life cycle: widths chosen at the build, appended runs, a rebuild, a table that cannot be packed), then the A/B on one table
set -u
O=$PWD/gpurun_out/r6y
mkdir -p $O
( time timeout 1500 python3 -m pytest tests/test_deliver_parity.py tests/test_formats_gpu.py tests/test_host_router.py -m gpu -x -q > $O/pytest.log 2>&1 ) 2> $O/pytest_time.txt; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest.log | tail -3
timeout 1200 python3 bench.py --time-format deliver8 --steps 3 --warmup 1 --ab-env "RGR_DELIVER_PACKED_READS=0,RGR_DELIVER_PACKED_READS=1" > $O/deliver8.jsonl 2> $O/deliver8.err; echo "rc=$?"
python3 - <<PY
import json
for ln in open("$O/deliver8.jsonl"):
    d = json.loads(ln)
    if "ab_check" in d: print("ab_check", d["ok"], d["delivery_parity"]["mismatching_words"]); continue
    print(d["env"], d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
