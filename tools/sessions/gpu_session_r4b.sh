# GPU session r4b: (1) the changed GPU tests (formats incl. IDS24, host router incl. async batcher / shared-lock passes, snapshot v1),
# (2) packed / ids24 at 2^30-hit windows: per-launch event cost, tiles per block, nontemporal vs plain stores,
# (3) Router::matches end to end with asynchronous submit + pipelined passes (config 2)
set -u
O=gpurun_out/r4b
mkdir -p $O
( timeout 600 python -m pytest tests/test_formats_gpu.py tests/test_host_router.py tests/test_snapshot.py tests/test_capi_cpu.py -m gpu -q -x --timeout 300 > $O/pytest_gpu_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_subset.log ); tail -3 $O/pytest_gpu_subset.log | cut -c1-300
WH=1073741824
run() { # label, extra env
  for f in packed ids24; do
    echo -n "$1 " >> $O/sweep.jsonl
    env $2 timeout 300 python bench.py --time-format $f --steps 5 --warmup 2 --window-hits $WH >> $O/sweep.jsonl 2>> $O/sweep.err
  done
}
run "base" "X=1"
run "nospans" "RGR_SPAN_SAMPLE=0"
for flags in "-DRGR_COMPACT_TILES=2" "-DRGR_COMPACT_TILES=4" "-DRGR_COMPACT_NT=0" "-DRGR_COMPACT_THREADS=512"; do
  RGR_EXTRA_FLAGS="$flags" python -c "from rmqtt_amd import build; build.build_gpu(force=True)" 2>> $O/sweep.err
  run "$flags" "RGR_EXTRA_FLAGS=$flags"
done
python -c "from rmqtt_amd import build; build.build_gpu(force=True)" 2>> $O/sweep.err
echo "window 2^28 ids24:" >> $O/sweep.jsonl
timeout 300 python bench.py --time-format ids24 --steps 5 --warmup 2 >> $O/sweep.jsonl 2>> $O/sweep.err
cut -c1-330 $O/sweep.jsonl
( timeout 500 python bench.py --router-e2e --e2e-configs 2 --e2e-sweep > $O/router_e2e_cfg2.jsonl 2> $O/router_e2e_cfg2.err ); echo "e2e rc=$?"; grep "router e2e" $O/router_e2e_cfg2.err | cut -c1-400
du -sh $O
