# GPU session r6t: the topic pass with the next item fetched ahead and two barriers per item (RGR_DEDUP_PROBE=a), emptier tables
# (RGR_DEDUP_SLOT_FACTOR=4) — parity under each switch, then the A/B on one table against the 256-entry list heads (=7)
set -u
O=$PWD/gpurun_out/r6t
mkdir -p $O
for m in a; do ( RGR_DEDUP_PROBE=$m timeout 900 python3 -m pytest tests/test_deliver_parity.py -m gpu -x -q > $O/pytest_probe_$m.log 2>&1 ); echo "probe $m rc=$?"; grep -E "passed|failed|error" $O/pytest_probe_$m.log | tail -2; done
( RGR_DEDUP_PROBE=a RGR_DEDUP_SLOT_FACTOR=4 timeout 900 python3 -m pytest tests/test_deliver_parity.py -m gpu -x -q > $O/pytest_probe_a_f4.log 2>&1 ); echo "probe a f4 rc=$?"; grep -E "passed|failed|error" $O/pytest_probe_a_f4.log | tail -2
timeout 1500 python3 bench.py --time-format deliver8 --steps 3 --warmup 1 --ab-env "RGR_DEDUP_PROBE=7,RGR_DEDUP_PROBE=a,RGR_DEDUP_PROBE=7+RGR_DEDUP_SLOT_FACTOR=4,RGR_DEDUP_PROBE=a+RGR_DEDUP_SLOT_FACTOR=4,RGR_DEDUP_PROBE=a+RGR_DEDUP_SLOT_FACTOR=8" > $O/deliver8.jsonl 2> $O/deliver8.err; echo "rc=$?"
python3 - <<PY
import json
for ln in open("$O/deliver8.jsonl"):
    d = json.loads(ln)
    if "ab_check" in d: print("ab_check", d["ok"], d["delivery_parity"]["mismatching_words"]); continue
    print(d["env"], d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
