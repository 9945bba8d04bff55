# GPU session r6x: the lean delivery expansion with its words stored BEFORE the v5 path (the attribute gather and the stores' back-pressure overlap; a hit
# the dense path changes stores again) against a build that stores behind it as until now (-DRGR_LEAN_STORE_LATE=1).  Parity first.
set -u
O=$PWD/gpurun_out/r6x
mkdir -p $O
( time timeout 1500 python3 -m pytest tests/test_deliver_parity.py tests/test_formats_gpu.py -m gpu -x -q > $O/pytest.log 2>&1 ) 2> $O/pytest_time.txt; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest.log | tail -3
for v in EARLY LATE; do
  if [ $v = EARLY ]; then export RGR_EXTRA_FLAGS=""; else export RGR_EXTRA_FLAGS="-DRGR_LEAN_STORE_LATE=1"; fi
  python3 -c "from rmqtt_amd import build as b; b.build_gpu(force=True)" > $O/build_$v.log 2>&1 || { echo "$v build failed"; tail -5 $O/build_$v.log; continue; }
  timeout 600 python3 bench.py --time-format deliver8,deliver --steps 3 --warmup 1 > $O/deliver_$v.jsonl 2> $O/deliver_$v.err
  python3 - <<PY
import json
for ln in open("$O/deliver_$v.jsonl"):
    try:
        d = json.loads(ln)
        print("$v", d.get("format"), d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"), d.get("delivery_parity", {}).get("mismatching_words"))
    except Exception as e:
        print("$v failed", e)
PY
done 2>&1 | tee $O/sweep.txt
export RGR_EXTRA_FLAGS=""
python3 -c "from rmqtt_amd import build as b; b.build_gpu(force=True)" > $O/build_restore.log 2>&1
