# GPU session r3p: does a larger output window help the headline?  (default 2^28 hits = 3 GiB of tuples per window, 553 windows per pass)
set -u
O=gpurun_out/r3p
mkdir -p $O
for lg in 29 30; do
  ( timeout 200 python bench.py --config 3 --steps 5 --warmup 2 --window-hits $((1<<lg)) --no-pmc --no-secondary --no-formats --no-d2h --cpu-sample 0 > $O/bench_w$lg.json 2> $O/bench_w$lg.err )
  python -c "
import json; d=json.load(open('$O/bench_w$lg.json')); print('window 2^$lg:', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['config']['windows_per_step'], d['roofline']['avg_launch_ms'])"
done 2>&1 | tee $O/window_size.txt
