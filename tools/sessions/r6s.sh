# GPU session r6s: the topic pass's candidate lists fetched ahead (RGR_DEDUP_PROBE=7: 256 entries per tile in one go; =b: (tile, quarter) units
# with the next unit in flight), the walk's statistic with one global atomic per block.  Parity first, then the A/B on one table.
set -u
O=$PWD/gpurun_out/r6s
mkdir -p $O
( time timeout 1500 python3 -m pytest tests/test_parity.py tests/test_deliver_parity.py tests/test_formats_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 ) 2> $O/pytest_time.txt; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest.log | tail -3
for m in 7 b; do ( RGR_DEDUP_PROBE=$m timeout 900 python3 -m pytest tests/test_deliver_parity.py -m gpu -x -q > $O/pytest_probe_$m.log 2>&1 ); echo "probe $m rc=$?"; grep -E "passed|failed|error" $O/pytest_probe_$m.log | tail -2; done
timeout 1200 python3 bench.py --time-format deliver8 --steps 3 --warmup 1 --ab-env RGR_DEDUP_PROBE=3,RGR_DEDUP_PROBE=7,RGR_DEDUP_PROBE=b > $O/deliver8.jsonl 2> $O/deliver8.err; echo "rc=$?"
python3 - <<PY
import json
for ln in open("$O/deliver8.jsonl"):
    d = json.loads(ln)
    if "ab_check" in d: print("ab_check", d["ok"], d["delivery_parity"]["mismatching_words"]); continue
    print(d["env"], d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
timeout 600 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-pmc --no-secondary --no-formats --no-d2h > $O/headline.json 2> $O/headline.err; echo "headline rc=$?"
python3 -c "
import json; d=json.loads(open('$O/headline.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('parity'))" 2>&1 | cut -c1-400
