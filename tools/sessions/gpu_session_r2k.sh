# GPU session r2k: expand_tuple4_kernel (four consecutive positions per lane, dwordx4 loads and stores) against expand_kernel<false>
set -u
O=gpurun_out/r2k
mkdir -p $O
( RGR_TUPLE4=1 timeout 600 python -m pytest tests/test_parity.py tests/test_formats_gpu.py tests/test_properties_gpu.py tests/test_hypothesis_parity.py tests/test_golden_fixtures.py tests/test_retain_parity.py tests/test_snapshot.py tests/test_publish_packets.py tests/test_group_gpu.py -m gpu -q -x -k "not rccl" > $O/pytest_tuple4.log 2>&1; echo "pytest rc=$?" >> $O/pytest_tuple4.log )
tail -3 $O/pytest_tuple4.log
B="--steps 8 --warmup 2 --no-pmc --no-secondary --cpu-sample 0 --no-d2h --no-formats"
for V in 1 0; do
  ( RGR_TUPLE4=$V timeout 400 python bench.py $B --config 3 > $O/bench_cfg3_tuple4_$V.json 2> $O/bench_cfg3_tuple4_$V.err )
done
( RGR_TUPLE4=1 timeout 200 python bench.py $B --config 5 > $O/bench_cfg5_tuple4_1.json 2> $O/bench_cfg5_tuple4_1.err )
( RGR_TUPLE4=1 timeout 200 python bench.py $B --config 2 > $O/bench_cfg2_tuple4_1.json 2> $O/bench_cfg2_tuple4_1.err )
python - <<PY
import json
for g in ("cfg3_tuple4_1","cfg3_tuple4_0","cfg5_tuple4_1","cfg2_tuple4_1"):
    try:
        d=json.load(open("$O/bench_%s.json" % g)); r=d["roofline"]
        print(g, d["value"], d["ms_per_step"], d["kernel_ms_per_step"], r["kernel"], r["avg_launch_ms"])
    except Exception as e: print(g, "failed", e)
PY
