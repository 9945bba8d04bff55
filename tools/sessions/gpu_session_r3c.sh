# GPU session r3c: (1) where does the delivery stage's time go (rocprofv3 kernel trace of the --deliver 0.1 bench),
# (2) config-3 parity sample again (r3b: torch was still reading a window while the library expanded the next one into the
# same buffer — a race in the bench's checker, fixed with a device synchronize per window), (3) host Router mirror tests
set -u
O=gpurun_out/r3c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_deliver -o deliver -- python $GRAFT_REPO_ROOT/bench.py --config 3 --steps 2 --warmup 1 --no-pmc --no-secondary --no-d2h --deliver 0.1 > $GRAFT_REPO_ROOT/$O/bench_deliver_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_deliver_prof.err )
cd $GRAFT_REPO_ROOT
python profiles/summarize_kernel_trace.py $O/prof_deliver > $O/deliver_kernel_stats.txt 2>&1 || find $O/prof_deliver -name "*stats*" | head
head -30 $O/deliver_kernel_stats.txt
find $O/prof_deliver -type f -size +2M -delete
( timeout 600 python -m pytest tests/test_host_router.py tests/test_formats_gpu.py -m gpu -q -x > $O/pytest_host_router.log 2>&1 ); tail -5 $O/pytest_host_router.log
( timeout 900 python bench.py --config 3 --steps 5 --warmup 2 --no-pmc --no-secondary --no-d2h --no-formats > $O/bench_cfg3.json 2> $O/bench_cfg3.err ); tail -3 $O/bench_cfg3.err | cut -c1-1200
