# GPU session r6c: (1) delivery parity worlds incl. the 8-byte hit format; (2) A/B on one table, full size: 12-byte delivery tuples vs 8-byte hits,
# exempt runs off / on / on without the index probes (a DIAGNOSTIC: wrong flags, tells what the probes cost)
set -u
O=$PWD/gpurun_out/r6c
mkdir -p $O
( time timeout 1200 python3 -m pytest tests/test_deliver_parity.py -m gpu -x -q > $O/pytest_deliver.log 2>&1 ) 2> $O/pytest_deliver_time.txt; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_deliver.log | tail -3
( time timeout 1500 python3 bench.py --time-format deliver,deliver8 --steps 3 --warmup 1 --ab-env RGR_DELIVER_EXEMPT=0,RGR_DELIVER_EXEMPT=1,RGR_DELIVER_EXEMPT=1+RGR_DIAG_EX_NOPROBE=1 > $O/ab_deliver8_exempt.jsonl 2> $O/ab_deliver8_exempt.err ) 2> $O/ab_time.txt; echo "ab rc=$?"
cat $O/ab_deliver8_exempt.jsonl | cut -c1-700
