# GPU session r5e: LDS atomic microbenchmark + kernel trace of the delivery pass with either topic pass
set -u
O=$PWD/gpurun_out/r5e
mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/lds_atomic_bench > $O/lds_atomic_bench.txt 2>&1; echo "lds bench rc=$?"; cat $O/lds_atomic_bench.txt
R=$PWD
cd /tmp
for v in 0 1; do
  RGR_DEDUP_BATCH=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_batch$v -o t -- python $R/bench.py --time-format deliver --steps 1 --warmup 0 > /dev/null 2> $O/trace_batch$v.err
  echo "trace $v rc=$?"
  f=$(find $O/trace_batch$v -name "*kernel_stats.csv" | head -1)
  head -8 "$f" | cut -c1-260 | tee $O/kernel_stats_dedup_batch$v.txt
  rm -rf $O/trace_batch$v
done
