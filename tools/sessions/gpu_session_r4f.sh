# GPU session r4f: packed / ids24 reading the 4-byte packed side array vs the 8-byte entries (A/B by environment switch), formats tests
set -u
O=gpurun_out/r4f
mkdir -p $O
( timeout 600 python -m pytest tests/test_formats_gpu.py tests/test_retain_parity.py tests/test_parity.py tests/test_retain_tiers.py -m gpu -q -x --timeout 300 > $O/pytest_gpu_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_subset.log ); tail -3 $O/pytest_gpu_subset.log | cut -c1-300
for f in packed ids24; do
  echo -n "packed-reads " >> $O/ab.jsonl; timeout 300 python bench.py --time-format $f --steps 5 --warmup 2 >> $O/ab.jsonl 2>> $O/ab.err
  echo -n "entry-reads  " >> $O/ab.jsonl; RGR_NO_PACKED_READS=1 timeout 300 python bench.py --time-format $f --steps 5 --warmup 2 >> $O/ab.jsonl 2>> $O/ab.err
done
echo -n "packed-reads nospans " >> $O/ab.jsonl; RGR_SPAN_SAMPLE=0 timeout 300 python bench.py --time-format ids24 --steps 5 --warmup 2 >> $O/ab.jsonl 2>> $O/ab.err
cut -c1-340 $O/ab.jsonl
du -sh $O
