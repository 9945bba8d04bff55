# GPU session r6k: 8-byte delivery hits in walk order: window size and expansion geometry once more (the optimum of the 12-byte form was 2^27 hits, 512 x 4)
set -u
O=$PWD/gpurun_out/r6k
mkdir -p $O
timeout 1200 python3 bench.py --time-format deliver8 --steps 3 --warmup 1 --ab-env RGR_DELIVER_WINDOW_HITS=134217728,RGR_DELIVER_WINDOW_HITS=67108864,RGR_DELIVER_WINDOW_HITS=268435456,RGR_DELIVER_WINDOW_HITS=134217728+RGR_DELIVER_LEAN=2 > $O/ab_deliver8_window_geometry.jsonl 2> $O/ab.err; echo "rc=$?"
python3 - <<PY
import json
for ln in open("$O/ab_deliver8_window_geometry.jsonl"):
    d = json.loads(ln)
    if "ab_check" in d: print("ab_check", d["ok"], d["ab_check"][1]); continue
    print(d["env"], d["value"], d["ms_per_step"], d["windows_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
