# GPU session r6v: what the two dependent gathers of the lean delivery expansion's v5 path cost (synthetic attributes: upper bound of what
# removing them can give) and whether touching the attribute lines early helps — diagnostic builds, results wrong on purpose, product build restored
set -u
O=$PWD/gpurun_out/r6v
mkdir -p $O
for v in LEAN_NO_ATTR LEAN_WARM; do
  tag=$(echo "$v" | tr -d ' ' | sed 's/-DRGR_DIAG_/+/g')
  if [ "$v" = BASE ]; then export RGR_EXTRA_FLAGS=""; else export RGR_EXTRA_FLAGS="-DRGR_DIAG_BUILD -DRGR_DIAG_$v"; fi
  python3 -c "from rmqtt_amd import build as b; b.build_gpu(force=True)" > $O/build_$tag.log 2>&1 || { echo "$tag build failed"; tail -5 $O/build_$tag.log; continue; }
  timeout 600 python3 bench.py --time-format deliver8 --steps 2 --warmup 1 > $O/deliver8_$tag.jsonl 2> $O/deliver8_$tag.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$O/deliver8_$tag.jsonl").read().strip().splitlines()[0])
    print("$tag", d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"], d.get("dedup_avg_launch_ms"))
except Exception as e:
    print("$tag failed", e)
PY
done 2>&1 | tee $O/sweep.txt
export RGR_EXTRA_FLAGS=""
python3 -c "from rmqtt_amd import build as b; b.build_gpu(force=True)" > $O/build_restore.log 2>&1
