# GPU session r3d: full suite with the round's new entry points (run gather, grouped delivery, filter-subs Router path, churn test),
# delivery stage after the tile-pass skip, BASELINE configs[3] layout on one GPU with key_levels 1 vs 3 and the run gather
set -u
O=gpurun_out/r3d
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -12 $O/pytest_gpu.log
( timeout 400 python bench.py --config 3 --steps 3 --warmup 1 --no-pmc --no-secondary --no-d2h --deliver 0.1 > $O/bench_cfg3_deliver0.1.json 2> $O/bench_cfg3_deliver0.1.err ); tail -2 $O/bench_cfg3_deliver0.1.err | cut -c1-300
( timeout 400 python bench.py --config 3 --steps 3 --warmup 1 --no-pmc --no-secondary --no-d2h --deliver 0.0 > $O/bench_cfg3_deliver0.0.json 2> $O/bench_cfg3_deliver0.0.err )
for kl in 1 3; do
  ( timeout 500 python bench.py --group 8 --key-levels $kl --steps 3 --warmup 1 --gather runs > $O/bench_group8_keylevels$kl.json 2> $O/bench_group8_keylevels$kl.err ); tail -2 $O/bench_group8_keylevels$kl.err | cut -c1-300
done
python - <<PY
import json
for f in ("bench_cfg3_deliver0.1","bench_cfg3_deliver0.0","bench_group8_keylevels1","bench_group8_keylevels3"):
    try:
        d=json.load(open("$O/"+f+".json"))
        print(f, d["value"], d["ms_per_step"], d.get("kernel_ms_per_step"), d.get("delivery_stage"), d.get("shard_imbalance_max_over_mean"), d.get("replicated_subs"), d.get("run_gather"))
    except Exception as e:
        print(f, "unreadable", e)
PY
