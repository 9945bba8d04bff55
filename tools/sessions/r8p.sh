# GPU session r8p: 2^28-hit windows as the default of device-resident delivery passes in 8-byte hits — delivery / property tests, the pass timed in both
# delivery formats with its whole-window parity
set -u
O=$PWD/gpurun_out/r8p
mkdir -p $O
( timeout 1500 python3 -m pytest tests/test_deliver_parity.py tests/test_properties_gpu.py tests/test_formats_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 ); echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest.log | tail -3
timeout 1200 python3 bench.py --time-format deliver8,deliver --steps 3 --warmup 1 > $O/deliver.jsonl 2> $O/deliver.err; echo "rc=$?"
python3 - <<PY
import json
for ln in open("$O/deliver.jsonl"):
    d = json.loads(ln)
    print(d.get("format"), d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"), (d.get("delivery_parity") or {}).get("windows_checked"), (d.get("delivery_parity") or {}).get("of_windows"), (d.get("delivery_parity") or {}).get("mismatching_words"))
PY
