# GPU session r2d: full suite on the final kernels, default bench (driver-style), rocprofv3 kernel stats of the same command, latency probes
set -u
O=gpurun_out/r2d
mkdir -p $O
R=$(pwd)
( timeout 1200 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log )
tail -4 $O/pytest_gpu.log
( timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/bench_default.err )
tail -2 $O/bench_default.err
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o r2d -- python $R/bench.py --steps 5 --warmup 1 --no-pmc --no-secondary --cpu-sample 0 --no-d2h > $R/$O/bench_under_rocprof.json 2> $R/$O/bench_under_rocprof.err )
python profiles/summarize_kernel_trace.py $O/prof > $O/kernel_stats.txt 2>&1
head -14 $O/kernel_stats.txt
( timeout 400 python tools/latency.py 3 1.0 > $O/latency_cfg3_full.txt 2>&1 ); tail -3 $O/latency_cfg3_full.txt
( timeout 200 python tools/latency.py 2 1.0 > $O/latency_cfg2.txt 2>&1 ); tail -14 $O/latency_cfg2.txt
