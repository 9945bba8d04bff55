# GPU session r3a (prepared at the end of round 2, NOT yet run): the measurements DESIGN §12 names first.
#   1. GPU suite + smoke of the tree as it stands
#   2. walk: hash edge table vs CSR children lists at config-3 size (neither cache-resident), and edge-table density
#      through the PRODUCT (RGR_EDGE_SLOTS_PER_NODE is read by HostTable::materialize_edges) at config 2 and config 3
#   3. gather rate vs table size (is the walk's 38.9 G gathers/s ceiling a property of HBM or of the 16 GiB table the
#      calibration used?)
#   0. (before this session) `git merge next/bitmap-miss-filter`: the 64-bit child-token bitmap miss filter, CPU-verified only;
#      step 1 is then its GPU validation and step 2's config-2 line its measurement (walk 0.585 ms per 1 M topics before)
set -u
O=gpurun_out/r3a
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -3 $O/pytest_gpu.log
( timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1 ); tail -1 $O/smoke.log
# maximum MQTT sizes (65 535-byte levels, 32 768-level topics, 2^14-way forks) through the HIP path: so far emulator only
( RMQTT_MAX_SIZES_BACKEND=hip timeout 300 python -m pytest tests/test_max_sizes_cpu.py -q -m gpu > $O/pytest_max_sizes_hip.log 2>&1 ); tail -2 $O/pytest_max_sizes_hip.log
[ -x tools/walk_lab ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rmqtt_amd/csrc -I include tools/walk_lab.hip rmqtt_amd/csrc/table.cpp \
    rmqtt_amd/csrc/workload.cpp -o tools/walk_lab -pthread
( timeout 300 tools/walk_lab 10000000 2000000 0.028 0.1 5 ) > $O/walk_lab_config3_size.txt 2>&1; cat $O/walk_lab_config3_size.txt
#   4. v5 dedup: global table vs LDS tables (tools/dedup_lab.hip), config-3 shape and a low-fan-out shape
[ -x tools/dedup_lab ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rmqtt_amd/csrc -I include tools/dedup_lab.hip -o tools/dedup_lab
( timeout 120 tools/dedup_lab 26; timeout 120 tools/dedup_lab 26 50 0.1 2500000 5 1.0; timeout 120 tools/dedup_lab 26 14800 0.5 ) > $O/dedup_lab.txt 2>&1; cat $O/dedup_lab.txt
for x in 4 8; do
  for cfg in 2 3; do
    ( RGR_EDGE_SLOTS_PER_NODE=$x timeout 600 python bench.py --config $cfg --steps 10 --warmup 3 --no-pmc --no-secondary --no-formats --no-d2h --cpu-sample 0 \
        > $O/bench_cfg${cfg}_slots${x}.json 2> $O/bench_cfg${cfg}_slots${x}.err )
    python - <<PY
import json
d=json.load(open("$O/bench_cfg${cfg}_slots${x}.json"))
print("cfg$cfg slots/node $x:", d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["table"]["hbm_bytes"])
PY
  done
done
