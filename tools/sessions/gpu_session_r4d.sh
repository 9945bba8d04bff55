# GPU session r4d: (1) whole GPU suite (default window now 2^30, 8-byte candidates + single-pass topic dedup), (2) compact kernels with more
# positions in flight per lane (they are latency-bound: ids24 and packed take the same time), (3) e2e async with the batcher's timing,
# (4) the delivery stage after the dedup rewrite
set -u
O=gpurun_out/r4d
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -4 $O/pytest_gpu.log | cut -c1-300
run() { for f in packed ids24; do echo -n "$1 " >> $O/sweep.jsonl; env $2 timeout 300 python bench.py --time-format $f --steps 5 --warmup 2 >> $O/sweep.jsonl 2>> $O/sweep.err; done; }
run "base(256x2)" "X=1"
for flags in "-DRGR_COMPACT_THREADS=128" "-DRGR_COMPACT_THREADS=64"; do
  RGR_EXTRA_FLAGS="$flags" python -c "from rmqtt_amd import build; build.build_gpu(force=True)" 2>> $O/sweep.err
  run "$flags" "RGR_EXTRA_FLAGS=$flags"
done
python -c "from rmqtt_amd import build; build.build_gpu(force=True)" 2>> $O/sweep.err
cut -c1-330 $O/sweep.jsonl
( timeout 400 python bench.py --router-e2e --e2e-configs 2 --e2e-sweep > $O/router_e2e_cfg2.jsonl 2> $O/router_e2e_cfg2.err ); echo "e2e rc=$?"; grep "router e2e" $O/router_e2e_cfg2.err | cut -c1-600
( time timeout 600 python bench.py --deliver 0.1 --steps 3 --warmup 1 --no-secondary --no-pmc --cpu-sample 0 > $O/bench_deliver.json 2> $O/bench_deliver.err ) 2> $O/t_deliver.txt; echo "deliver rc=$?"; grep real $O/t_deliver.txt
python - <<PY
import json
try:
    d=json.load(open("$O/bench_deliver.json"))
    print("deliver:", d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d.get("delivery_stage"))
    print("   parity:", {k:v for k,v in d["parity_sample"].items() if k!="what"})
except Exception as e: print("deliver parse failed", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_deliver -o t -- python $GRAFT_REPO_ROOT/bench.py --deliver 0.1 --steps 2 --warmup 1 --no-secondary --no-pmc --cpu-sample 0 --no-parity > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof_deliver.err
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_deliver -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200 | tee $O/deliver_kernel_stats.txt
find $O/prof_deliver -name "*.csv" -size +5M -delete
du -sh $O
