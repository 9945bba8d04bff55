# GPU session r3b: validate the round-3 kernel changes and the new full-size parity sample
#   LDS-staged compaction, slot capacity 64, walk with ds_read (two instances), LDS v5 dedup (tile + topic tables),
#   bench.py parity_sample = full-pass digests in every format + stratified oracle sample
set -u
O=gpurun_out/r3b
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -15 $O/pytest_gpu.log
( timeout 300 python bench.py --config 2 --steps 10 --warmup 3 --no-pmc --no-secondary --no-d2h > $O/bench_cfg2.json 2> $O/bench_cfg2.err ); tail -3 $O/bench_cfg2.err
( timeout 600 python bench.py --config 3 --steps 5 --warmup 2 --no-pmc --no-secondary --no-d2h --deliver 0.1 > $O/bench_cfg3_deliver0.1.json 2> $O/bench_cfg3_deliver0.1.err ); tail -2 $O/bench_cfg3_deliver0.1.err
( timeout 900 python bench.py --config 3 --steps 10 --warmup 3 --no-pmc --no-secondary --no-d2h > $O/bench_cfg3.json 2> $O/bench_cfg3.err ); tail -4 $O/bench_cfg3.err
python - <<PY
import json
for f in ("bench_cfg2","bench_cfg3_deliver0.1","bench_cfg3"):
    try:
        d=json.load(open("$O/"+f+".json"))
        print(f, d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d.get("delivery_stage"), d.get("parity_sample"), [ (c.get("format","")[:6], c.get("value")) for c in d.get("compact_formats",[])])
    except Exception as e:
        print(f, "unreadable", e)
PY
