# GPU session r8o: delivery windows of 2^26 / 2^27 / 2^28 hits with the round's dedup passes and 8-byte hits (2^27 was chosen in round 5)
set -u
O=$PWD/gpurun_out/r8o
mkdir -p $O
timeout 1500 python3 bench.py --time-format deliver8 --steps 3 --warmup 1 --ab-env "RGR_DELIVER_WINDOW_HITS=268435456,RGR_DELIVER_WINDOW_HITS=536870912,RGR_DELIVER_WINDOW_HITS=1073741824,RGR_DELIVER_WINDOW_HITS=134217728" > $O/deliver8.jsonl 2> $O/deliver8.err; echo "rc=$?"
python3 - <<PY
import json
for ln in open("$O/deliver8.jsonl"):
    d = json.loads(ln)
    if "ab_check" in d: print("ab_check", d["ok"], d["delivery_parity"]["mismatching_words"]); continue
    print(d["env"], d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
