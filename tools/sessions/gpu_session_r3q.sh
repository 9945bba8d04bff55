# GPU session r3q (last GPU minutes of the round): tile records of all windows of a chunk in one launch — parity subset + timing
set -u
O=gpurun_out/r3q
mkdir -p $O
( timeout 150 python -m pytest tests/test_parity.py tests/test_formats_gpu.py tests/test_deliver_parity.py tests/test_retain_parity.py "tests/test_properties_gpu.py::test_config3_windows_invariants_and_sampled_oracle" "tests/test_properties_gpu.py::test_streamed_to_host_pass_equals_host_result" -m gpu -q --timeout 100 > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log ); tail -3 $O/pytest_subset.log | cut -c1-200
( timeout 120 python bench.py --config 3 --steps 4 --warmup 1 --no-pmc --no-secondary --no-d2h --cpu-sample 0 > $O/bench_cfg3.json 2> $O/bench_cfg3.err )
python -c "
import json; d=json.load(open('$O/bench_cfg3.json')); print('cfg3', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], [(c['format'][:6], c['value']) for c in d.get('compact_formats',[])])"
