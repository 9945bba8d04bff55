# GPU session r4j: tile records of the next window on a side stream (A/B by environment switch) + the whole GPU suite behind it
set -u
O=gpurun_out/r4j
mkdir -p $O
for f in ids24 packed tuple; do
  echo -n "tile-prefetch " >> $O/ab.jsonl; timeout 300 python bench.py --time-format $f --steps 5 --warmup 2 >> $O/ab.jsonl 2>> $O/ab.err
  echo -n "inline-tiles  " >> $O/ab.jsonl; RGR_NO_TILE_PREFETCH=1 timeout 300 python bench.py --time-format $f --steps 5 --warmup 2 >> $O/ab.jsonl 2>> $O/ab.err
done
cut -c1-330 $O/ab.jsonl
( timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -4 $O/pytest_gpu.log | cut -c1-300
