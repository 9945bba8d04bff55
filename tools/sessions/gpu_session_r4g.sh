# GPU session r4g: compact expansion with several tiles per block, software-pipelined (records, then all loads, then all stores)
set -u
O=gpurun_out/r4g
mkdir -p $O
for flags in "-DRGR_COMPACT_TILES=4" "-DRGR_COMPACT_TILES=2" "-DRGR_COMPACT_TILES=8"; do
  RGR_EXTRA_FLAGS="$flags" python -c "from rmqtt_amd import build; build.build_gpu(force=True)" 2>> $O/sweep.err
  if [ "$flags" = "-DRGR_COMPACT_TILES=4" ]; then
    ( RGR_EXTRA_FLAGS="$flags" timeout 600 python -m pytest tests/test_formats_gpu.py tests/test_properties_gpu.py -m gpu -q -x --timeout 300 > $O/pytest_tiles4.log 2>&1; echo "pytest rc=$?" >> $O/pytest_tiles4.log ); tail -3 $O/pytest_tiles4.log | cut -c1-200
  fi
  for f in packed ids24; do echo -n "$flags " >> $O/sweep.jsonl; RGR_EXTRA_FLAGS="$flags" timeout 300 python bench.py --time-format $f --steps 5 --warmup 2 >> $O/sweep.jsonl 2>> $O/sweep.err; done
done
python -c "from rmqtt_amd import build; build.build_gpu(force=True)" 2>> $O/sweep.err
cut -c1-340 $O/sweep.jsonl
