# GPU session r3a: first session of round 3 — validate on hardware what ended round 2 CPU-only.
#   1. full GPU suite + smoke of the tree with the bitmap miss filter merged (tests/test_max_sizes.py is now [emu, hip])
#   2. walk: hash edge table vs CSR children lists at config-3 size (neither cache-resident)
#   3. v5 dedup: global table vs LDS tables (tools/dedup_lab.hip), config-3 shape and a low-fan-out shape
#   4. bench config 2 and config 3 (walk ms with the bitmap; before: 0.591 ms per 1 M topics at config 2, 18.9 ms per pass at config 3)
#   5. config-3 property test at full scale
set -u
O=gpurun_out/r3a
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -3 $O/pytest_gpu.log
( timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1 ); tail -1 $O/smoke.log
( timeout 300 tools/walk_lab 10000000 2000000 0.028 0.1 5 ) > $O/walk_lab_config3_size.txt 2>&1; tail -12 $O/walk_lab_config3_size.txt
( timeout 120 tools/dedup_lab 26; timeout 120 tools/dedup_lab 26 50 0.1 2500000 5 1.0; timeout 120 tools/dedup_lab 26 14800 0.5 ) > $O/dedup_lab.txt 2>&1; cat $O/dedup_lab.txt
for cfg in 2 3; do
  ( timeout 600 python bench.py --config $cfg --steps 10 --warmup 3 --no-pmc --no-secondary --no-d2h --cpu-sample 0 \
      > $O/bench_cfg${cfg}.json 2> $O/bench_cfg${cfg}.err )
  python - <<PY
import json
d=json.load(open("$O/bench_cfg${cfg}.json"))
print("cfg$cfg:", d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d.get("compact_formats"))
PY
done
( RMQTT_TEST_SCALE=1.0 timeout 900 python -m pytest tests/test_properties_gpu.py -m gpu -q -k "config3_windows" > $O/pytest_config3_fullscale.log 2>&1 ); tail -2 $O/pytest_config3_fullscale.log
