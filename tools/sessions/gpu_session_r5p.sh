# GPU session r5p: the tree after the lean delivery expansion — whole GPU suite, smoke, the driver's own command (python3 bench.py --gpus 1
# --steps 20 --warmup 5: is the ONE stdout line the compact one?), rocprofv3 kernel trace of a shorter run of the same program
set -u
O=$PWD/gpurun_out/r5p
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -4 $O/pytest_gpu.log | cut -c1-300
( timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1 ); tail -1 $O/smoke.log
( time timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err ) 2> $O/bench_driver_cmd_time.txt; echo "bench rc=$?"; tail -3 $O/bench_driver_cmd_time.txt
wc -c $O/bench_driver_cmd.json; wc -l $O/bench_driver_cmd.json
cp gpurun_out/bench_detail_n1.json $O/bench_detail_n1.json 2>/dev/null
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_driver_cmd.json").read().strip().splitlines()[-1])
    print("line keys:", list(d.keys()))
    print("default:", d["value"], d["ms_per_step"], d.get("roofline"), d.get("cpu_baseline"))
    print(json.dumps(d)[:3000])
except Exception as e: print("parse failed", e)
PY
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 5 --warmup 2 --no-secondary --no-pmc --cpu-sample 0 --no-d2h > $O/bench_under_rocprofv3.json 2> $O/bench_under_rocprofv3.err
echo "trace rc=$?"
f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
cp "$f" $O/bench_config3_kernel_stats_rocprofv3.csv 2>/dev/null
head -12 "$f" | cut -c1-200
rm -rf $O/trace
du -sh $O
