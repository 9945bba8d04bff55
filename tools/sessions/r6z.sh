# GPU session r6z: the lean delivery expansion with the window's tiles shared between two launches (RGR_DELIVER_LEAN=3): tiles of up to 510 pairs by
# 256 x 8 blocks with 6.7 KB of pair staging (five blocks per CU by registers: 160 positions in flight per lane slot instead of 128), denser tiles by the
# 512 x 4 kernel.  Parity under the switch, then the A/B on one table (8-byte and 4-byte entries).
set -u
O=$PWD/gpurun_out/r6z
mkdir -p $O
( RGR_DELIVER_LEAN=3 timeout 1500 python3 -m pytest tests/test_deliver_parity.py tests/test_formats_gpu.py -m gpu -x -q > $O/pytest_lean3.log 2>&1 ); echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_lean3.log | tail -3
timeout 1500 python3 bench.py --time-format deliver8 --steps 3 --warmup 1 --ab-env "RGR_DELIVER_LEAN=1,RGR_DELIVER_LEAN=3,RGR_DELIVER_LEAN=3+RGR_DELIVER_PACKED_READS=0,RGR_DELIVER_LEAN=2" > $O/deliver8.jsonl 2> $O/deliver8.err; echo "rc=$?"
python3 - <<PY
import json
for ln in open("$O/deliver8.jsonl"):
    d = json.loads(ln)
    if "ab_check" in d: print("ab_check", d["ok"], d["delivery_parity"]["mismatching_words"]); continue
    print(d["env"], d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
PY
