# GPU session r2h: sharded single-process router at config-4 layout on one GPU; delivery stage on its own geometry
set -u
O=gpurun_out/r2h
mkdir -p $O
( timeout 600 python -m pytest tests/test_deliver_parity.py tests/test_parity.py tests/test_properties_gpu.py tests/test_formats_gpu.py -m gpu -q -x > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log ); tail -3 $O/pytest_subset.log
( timeout 600 python bench.py --group 8 --steps 3 --warmup 1 > $O/bench_group8_config3.json 2> $O/bench_group8_config3.err ); tail -2 $O/bench_group8_config3.err; head -c 1500 $O/bench_group8_config3.json; echo
( timeout 300 python bench.py --group 8 --scale 0.1 --gather tuples --window-hits 4194304 --steps 2 --warmup 1 > $O/bench_group8_scale0.1_allgatherv.json 2> $O/bench_group8_scale0.1_allgatherv.err ); head -c 1200 $O/bench_group8_scale0.1_allgatherv.json; echo
( timeout 300 python bench.py --group 2 --steps 3 --warmup 1 > $O/bench_group2_config3.json 2> $O/bench_group2_config3.err ); head -c 900 $O/bench_group2_config3.json; echo
B="--steps 5 --warmup 2 --config 3 --no-pmc --no-secondary --cpu-sample 0 --no-d2h"
( timeout 400 python bench.py $B --deliver 0.1 > $O/bench_deliver_v5frac0.1.json 2> $O/bench_deliver_v5frac0.1.err )
python - <<PY
import json
d=json.load(open("$O/bench_deliver_v5frac0.1.json")); print("deliver 0.1:", d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d.get("delivery_stage"))
PY
