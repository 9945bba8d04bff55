# GPU session r6n: how many leading levels should order the batch?  (RGR_ORDER_LEVELS = 2 / 4 / 6 / 8: one stable sort per pair of levels)
set -u
O=$PWD/gpurun_out/r6n
mkdir -p $O
for lv in 4 6 8; do
  RGR_ORDER_LEVELS=$lv timeout 900 python3 bench.py --time-format tuple,ids24,runs --steps 5 --warmup 2 > $O/config3_order_levels_$lv.jsonl 2> $O/err_$lv.txt; echo "c3 $lv rc=$?"
  RGR_ORDER_LEVELS=$lv timeout 600 python3 bench.py --config 2 --time-format tuple --steps 20 --warmup 3 > $O/config2_order_levels_$lv.jsonl 2>> $O/err_$lv.txt; echo "c2 $lv rc=$?"
done
python3 - <<PY
import json
for lv in (4, 6, 8):
    for f in ("config3", "config2"):
        for ln in open(f"$O/{f}_order_levels_{lv}.jsonl"):
            d = json.loads(ln)
            print(lv, f, d["format"], d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["expand_avg_launch_ms"])
PY
