# GPU session r3l: rocprofv3 kernel trace of the delivery stage on the final tree (per-kernel times behind DESIGN §11)
set -u
O=gpurun_out/r3l
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( timeout 500 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o deliver -- python $GRAFT_REPO_ROOT/bench.py --config 3 --steps 2 --warmup 1 --no-pmc --no-secondary --no-d2h --no-parity --deliver 0.1 > $GRAFT_REPO_ROOT/$O/bench_deliver_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_deliver_under_rocprof.err )
cd $GRAFT_REPO_ROOT
python profiles/summarize_kernel_trace.py $O/prof > $O/deliver_kernel_stats.txt 2>&1; head -14 $O/deliver_kernel_stats.txt
find $O/prof -type f -size +1M -delete
