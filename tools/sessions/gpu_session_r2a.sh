set -u
mkdir -p gpurun_out/r2a
O=gpurun_out/r2a
R=$(pwd)
( timeout 900 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log )
tail -3 $O/pytest_gpu.log
# FETCH_SIZE calibration
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/calib -o pmc -- $R/tools/membench calib > $R/$O/calib/calib.txt 2> $R/$O/calib.err ) ; mkdir -p $O/calib
python tools/pmc_calibrate.py $O/calib --write $O/pmc_calibration.json > $O/calibration.txt 2>&1; cat $O/calibration.txt
[ -f $O/pmc_calibration.json ] && cp $O/pmc_calibration.json profiles/pmc_calibration.json
# headline bench with everything
( timeout 1200 python bench.py --steps 5 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/bench_default.err )
tail -5 $O/bench_default.err
# labs
( timeout 200 tools/expand_lab > $O/expand_lab.txt 2>&1 ); tail -8 $O/expand_lab.txt
( timeout 300 python tools/walk_order_lab.py 2 1.0 5 > $O/walk_order_lab.txt 2>&1 ); tail -4 $O/walk_order_lab.txt
( timeout 100 tools/membench > $O/membench.txt 2>&1 )
