# GPU session r4e: e2e async after request recycling / function-pointer completions (configs 2 and 3), delivery after the tile-range fix
set -u
O=gpurun_out/r4e
mkdir -p $O
( timeout 600 python -m pytest tests/test_host_router.py tests/test_deliver_parity.py tests/test_group_gpu.py tests/test_properties_gpu.py -m gpu -q -x --timeout 300 > $O/pytest_gpu_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_subset.log ); tail -3 $O/pytest_gpu_subset.log | cut -c1-300
( timeout 400 python bench.py --router-e2e --e2e-configs 2 --e2e-sweep > $O/router_e2e_cfg2.jsonl 2> $O/router_e2e_cfg2.err ); echo "e2e rc=$?"; grep "router e2e" $O/router_e2e_cfg2.err | cut -c1-600
grep -o '"cpu_reference_port": {[^}]*}' $O/router_e2e_cfg2.jsonl | cut -c1-300
( time timeout 600 python bench.py --deliver 0.1 --steps 3 --warmup 1 --no-secondary --no-pmc --cpu-sample 0 > $O/bench_deliver.json 2> $O/bench_deliver.err ) 2> $O/t_deliver.txt; echo "deliver rc=$?"
python - <<PY
import json
try:
    d=json.load(open("$O/bench_deliver.json"))
    print("deliver:", d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d.get("delivery_stage"))
    print("   parity:", {k:v for k,v in d["parity_sample"].items() if k!="what"})
except Exception as e: print("deliver parse failed", e)
PY
( timeout 600 python bench.py --router-e2e --e2e-configs 3 > $O/router_e2e_cfg3.jsonl 2> $O/router_e2e_cfg3.err ); echo "e2e3 rc=$?"; grep "router e2e" $O/router_e2e_cfg3.err | cut -c1-600
grep -o '"cpu_reference_port": {[^}]*}' $O/router_e2e_cfg3.jsonl | cut -c1-300
du -sh $O
