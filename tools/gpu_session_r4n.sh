# GPU session r4n (6 GPU-minutes left in the round): the lane-held compact expansion (RGR_COMPACT_LP, expand_compact.inc) against the
# tile-per-block kernel — ids24 and packed timed on ONE table build, the fastest value of each digested over a full pass against the
# product kernel — then its GPU parity test.  The kernel source was validated on the host first (tests/test_hipsim_expand.py, TSAN).
set -u
O=gpurun_out/r4n
mkdir -p $O
timeout 200 python bench.py --time-format ids24,packed --steps 4 --warmup 2 --ab-env RGR_COMPACT_LP=0,4,2 > $O/ab_lp.jsonl 2> $O/ab_lp.err
echo "ab rc=$?"; cut -c1-420 $O/ab_lp.jsonl; tail -3 $O/ab_lp.err | cut -c1-300
( timeout 150 python -m pytest tests/test_formats_gpu.py -q -x --timeout 120 -k "lane_held or equal_tuples" > $O/pytest_lp.log 2>&1; echo "pytest rc=$?" >> $O/pytest_lp.log ); tail -5 $O/pytest_lp.log | cut -c1-300
