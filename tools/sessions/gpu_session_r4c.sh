# GPU session r4c: (1) GPU tests of the retained range answer, the delivery digest, the async batcher; (2) ids24 after the dwordx3 fix,
# (3) Router::matches e2e async after the cheap submit path, (4) the delivery record with its oracle sample + CPU baseline,
# (5) config 5 with the range answer's PCIe-inclusive rate
set -u
O=gpurun_out/r4c
mkdir -p $O
( timeout 900 python -m pytest tests/test_retain_parity.py tests/test_retain_tiers.py tests/test_deliver_parity.py tests/test_host_router.py tests/test_formats_gpu.py -m gpu -q -x --timeout 300 > $O/pytest_gpu_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_subset.log ); tail -4 $O/pytest_gpu_subset.log | cut -c1-300
for f in ids24 packed; do timeout 300 python bench.py --time-format $f --steps 5 --warmup 2 --window-hits 1073741824 >> $O/formats_2e30.jsonl 2>> $O/formats.err; done
timeout 300 python bench.py --time-format ids24 --steps 5 --warmup 2 >> $O/formats_2e30.jsonl 2>> $O/formats.err
cut -c1-330 $O/formats_2e30.jsonl
( timeout 400 python bench.py --router-e2e --e2e-configs 2 --e2e-sweep > $O/router_e2e_cfg2.jsonl 2> $O/router_e2e_cfg2.err ); echo "e2e rc=$?"; grep "router e2e" $O/router_e2e_cfg2.err | cut -c1-420
( time timeout 600 python bench.py --deliver 0.1 --steps 3 --warmup 1 --no-secondary --no-pmc > $O/bench_deliver.json 2> $O/bench_deliver.err ) 2> $O/t_deliver.txt; echo "deliver rc=$?"; grep real $O/t_deliver.txt
python - <<PY
import json
try:
    d=json.load(open("$O/bench_deliver.json"))
    print("deliver:", d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d.get("delivery_stage"), d.get("cpu_baseline"))
    print("   parity:", {k:v for k,v in d["parity_sample"].items() if k!="what"})
except Exception as e: print("deliver parse failed", e)
PY
grep "bench +" $O/bench_deliver.err | cut -c1-200 | tail -8
( time timeout 600 python bench.py --config 5 --steps 3 --warmup 1 --no-secondary --no-pmc --no-formats > $O/bench_cfg5.json 2> $O/bench_cfg5.err ) 2> $O/t5.txt; echo "cfg5 rc=$?"; grep real $O/t5.txt
python - <<PY
import json
try:
    d=json.load(open("$O/bench_cfg5.json"))
    print("cfg5:", d["value"], d["pcie_inclusive_matches_per_s"], d.get("pcie_inclusive_ranges"), d["parity_sample"]["ok"])
except Exception as e: print("cfg5 parse failed", e)
PY
tail -3 $O/bench_cfg5.err | cut -c1-300
du -sh $O
